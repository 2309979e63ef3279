"""A/B of encoder builds/switches on ONE box: the same seeded batch is encoded by child processes that
differ only in environment variables (the engine reads its switches once per process); every child
saves its embeddings and prints its timings, the parent compares the outputs bit for bit.

  python scripts/ab_encode.py [model] [n_seq] VAR=a,b [VAR2=c,d ...]     e.g.  VR_GEMM_PP=0,1

Children run one after the other (one process on the GPU at a time), interleaved `rounds` times so that
clock drift shows up as spread, not as a difference."""
import itertools
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(model, n_seq, out_path, reps):
    import torch

    from oracle import bert as obert  # shape table only (bench infrastructure)
    from voitta_rag_amd import Engine
    from voitta_rag_amd import encoder as enc

    shape, pooling = obert.SHAPES[model]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    H, I = shape.hidden, shape.intermediate
    state = {}
    for n in enc.tensor_names(shape.layers):
        if n.endswith("word_embeddings.weight"): shp = (shape.vocab, H)
        elif n.endswith("position_embeddings.weight"): shp = (shape.max_pos, H)
        elif n.endswith("token_type_embeddings.weight"): shp = (shape.type_vocab, H)
        elif n.endswith("intermediate.dense.weight"): shp = (I, H)
        elif n.endswith("intermediate.dense.bias"): shp = (I,)
        elif n.endswith("output.dense.weight") and "attention" not in n: shp = (H, I)
        elif n.endswith(".weight") and "LayerNorm" not in n: shp = (H, H)
        else: shp = (H,)
        t = torch.randn(shp, device=dev, generator=g) * 0.02
        if "LayerNorm.weight" in n: t = t + 1.0
        state[n] = t
    e = Engine(H)
    enc.load_encoder(e, enc.BertDesc(shape.layers, H, shape.heads, I, pooling=pooling, precision="f16"), state)
    rng = np.random.default_rng(1)
    lens = rng.integers(int(os.environ.get('AB_MINLEN', 96)), int(os.environ.get('AB_MAXLEN', 140)) + 1, size=n_seq)
    ids = torch.from_numpy(rng.integers(0, shape.vocab, size=int(lens.sum())).astype(np.int32)).to(dev)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
    out = torch.empty((n_seq, H), device=dev)
    for _ in range(2):
        enc.encode(e, ids, off, out)
    torch.cuda.synchronize()
    e.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        enc.encode(e, ids, off, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    gemm_ms, gemm_n, gemm_flop = e.profile_read(Engine.PROF_GEMM)
    attn_ms, attn_n, attn_flop = e.profile_read(Engine.PROF_ATTENTION)
    e.profile(False)
    np.save(out_path, out.cpu().numpy())
    print(json.dumps({"ms_per_batch": round(dt * 1e3, 3), "tokens": int(lens.sum()),
                      "gemm_TF": round(gemm_flop / (gemm_ms * 1e-3) / 1e12, 1), "gemm_ms_per_batch": round(gemm_ms / reps, 3),
                      "attn_TF": round(attn_flop / (attn_ms * 1e-3) / 1e12, 1), "attn_ms_per_batch": round(attn_ms / reps, 3)}))


def main():
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5]))
        return
    args = [a for a in sys.argv[1:] if "=" not in a]
    model = args[0] if args else "bge-base-en-v1.5"
    n_seq = int(args[1]) if len(args) > 1 else 2048
    rounds = int(os.environ.get("AB_ROUNDS", "2"))
    reps = int(os.environ.get("AB_REPS", "5"))
    axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:] if "=" in a]
    combos = [dict(zip([k for k, _ in axes], vals)) for vals in itertools.product(*[v for _, v in axes])] or [{}]
    tmp = tempfile.mkdtemp()
    outs = {}
    for r in range(rounds):
        for ci, combo in enumerate(combos):
            env = dict(os.environ, **combo)
            path = os.path.join(tmp, f"o{ci}.npy")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", model, str(n_seq), path, str(reps)],
                               env=env, capture_output=True, text=True, cwd=ROOT)
            if p.returncode != 0:
                print(f"{combo}: FAILED rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-3000:]}", flush=True)
                sys.exit(1)
            print(f"round {r} {combo}: {p.stdout.strip().splitlines()[-1]}", flush=True)
            outs[ci] = np.load(path)
    base = outs[0]
    for ci in range(1, len(combos)):
        same = np.array_equal(base.view(np.uint32), outs[ci].view(np.uint32))
        cos = (base * outs[ci]).sum(1) / np.linalg.norm(base, axis=1) / np.linalg.norm(outs[ci], axis=1)
        print(f"{combos[ci]} vs {combos[0]}: bit-identical={same}  max|1-cos|={np.max(np.abs(1 - cos)):.3e}  "
              f"max|diff|={np.max(np.abs(base - outs[ci])):.3e}", flush=True)


if __name__ == "__main__":
    main()
