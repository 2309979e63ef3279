#!/usr/bin/env python3
"""End-to-end tour of the drop-in services on an MI355X, with a synthetic checkpoint (no network):
the same calls IndexingService._index_file_standard and the MCP `search` tool make in the reference
(src/voitta/services/indexing.py:526-563, mcp_server.py:469-485).

    python examples/quickstart.py            # needs the built library: python -c "import __graft_entry__ as g; g.build()"

With a real model: EMBEDDING_MODEL=/path/to/bge-base-en-v1.5 EMBEDDING_DIMENSION=768 python examples/quickstart.py --real
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORDS = ("vector database index retrieval query embedding sparse dense hybrid fusion ranking chunk document folder "
         "search engine kernel memory bandwidth wavefront matrix tile running jumped happily relational").split()


def synthetic_checkpoint(directory: str, hidden: int = 128) -> None:
    """A 2-layer BERT of width `hidden` with seeded weights and a word-level vocabulary."""
    from safetensors.numpy import save_file

    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("abcdefghijklmnopqrstuvwxyz.,!?") + \
            ["##" + c for c in "abcdefghijklmnopqrstuvwxyz"] + WORDS + ["##ing", "##ed", "##s"]
    cfg = {"model_type": "bert", "hidden_size": hidden, "num_hidden_layers": 2, "num_attention_heads": 4,
           "intermediate_size": 2 * hidden, "vocab_size": len(vocab), "max_position_embeddings": 128,
           "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu"}
    os.makedirs(os.path.join(directory, "1_Pooling"), exist_ok=True)
    json.dump(cfg, open(os.path.join(directory, "config.json"), "w"))
    json.dump([{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
               {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
               {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}],
              open(os.path.join(directory, "modules.json"), "w"))
    json.dump({"word_embedding_dimension": hidden, "pooling_mode_cls_token": True, "pooling_mode_mean_tokens": False},
              open(os.path.join(directory, "1_Pooling", "config.json"), "w"))
    open(os.path.join(directory, "vocab.txt"), "w", encoding="utf-8").write("\n".join(vocab) + "\n")
    rng = np.random.default_rng(0)
    H, I, V = hidden, 2 * hidden, len(vocab)
    t = {"bert.embeddings.word_embeddings.weight": (V, H), "bert.embeddings.position_embeddings.weight": (128, H),
         "bert.embeddings.token_type_embeddings.weight": (2, H), "bert.embeddings.LayerNorm.weight": (H,),
         "bert.embeddings.LayerNorm.bias": (H,)}
    for l in range(2):
        p = f"bert.encoder.layer.{l}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            t[p + n + ".weight"], t[p + n + ".bias"] = (H, H), (H,)
        t[p + "attention.output.LayerNorm.weight"], t[p + "attention.output.LayerNorm.bias"] = (H,), (H,)
        t[p + "intermediate.dense.weight"], t[p + "intermediate.dense.bias"] = (I, H), (I,)
        t[p + "output.dense.weight"], t[p + "output.dense.bias"] = (H, I), (H,)
        t[p + "output.LayerNorm.weight"], t[p + "output.LayerNorm.bias"] = (H,), (H,)
    state = {k: ((rng.standard_normal(s) * 0.05 + (1.0 if k.endswith("LayerNorm.weight") else 0.0)).astype(np.float32))
             for k, s in t.items()}
    save_file(state, os.path.join(directory, "model.safetensors"))


def main() -> None:
    work = tempfile.mkdtemp(prefix="voitta_quickstart_")
    if "--real" not in sys.argv:
        synthetic_checkpoint(os.path.join(work, "model"))
        os.environ["EMBEDDING_MODEL"] = os.path.join(work, "model")
        os.environ["EMBEDDING_DIMENSION"] = "128"
    os.environ["VOITTA_INDEX_DIR"] = os.path.join(work, "index")

    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import ChunkMetadata, get_vector_store

    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    rng = np.random.default_rng(1)
    files = {f"docs/file{i}.md": [" ".join(rng.choice(WORDS, size=40)) + "." for _ in range(32)] for i in range(8)}
    for path, chunks in files.items():                                   # indexing.py:526-563
        metas = [ChunkMetadata(file_path=path, folder_path="docs", index_folder="docs", file_name=os.path.basename(path),
                               chunk_index=i, total_chunks=len(chunks), start_char=0, end_char=len(c),
                               indexed_at="2026-01-01T00:00:00") for i, c in enumerate(chunks)]
        vs.store_chunks(list(zip(chunks, emb.embed_texts(chunks), metas)), sparse_vectors=sp.embed_texts(chunks))
    print("stored", vs.get_collection_info()["points_count"], "chunks of", len(files), "files")

    query = files["docs/file3.md"][5]
    hits = vs.search(emb.embed_query(query), limit=5, sparse_query=sp.embed_query(query), sparse_weight=0.1)   # mcp_server.py:469-485
    for h in hits:
        print(f"  {h.score:.4f}  {h.metadata.file_path}#{h.metadata.chunk_index}  {h.text[:50]}...")
    assert hits[0].metadata.file_path == "docs/file3.md" and hits[0].metadata.chunk_index == 5

    assert vs.delete_by_file("docs/file0.md") == 32 and vs.compact() == 32        # re-index / watcher delete, then reclaim

    # the same from RAW documents: ChunkingService + both tokenisers + the fused encode/tf/append call, pipelined
    # across file boundaries (the host stages of batch i+1 run while the GPU works on batch i)
    from voitta_rag_amd.indexer import BulkIndexer, ParsedFile
    raw = [ParsedFile(content="\n\n".join(" ".join(rng.choice(WORDS, size=int(rng.integers(20, 90)))) + "." for _ in range(12)),
                      file_path=f"notes/raw{i}.txt", folder_path="notes", index_folder="notes", file_name=f"raw{i}.txt",
                      source_modified_at=1_750_000_000 + i) for i in range(16)]
    counts = BulkIndexer(batch_chunks=256).index_files(raw)
    print("bulk-indexed", sum(counts.values()), "chunks cut from", len(raw), "raw documents;",
          vs.count_by_file("notes/raw3.txt"), "of them belong to notes/raw3.txt")
    from voitta_rag_amd.chunking import get_chunking_service
    probe = get_chunking_service().chunk_text(raw[3].content)[1].text        # a stored chunk's own text finds itself
    hit = vs.search(emb.embed_query(probe), limit=1, include_folders=["notes"], sparse_query=sp.embed_query(probe))[0]
    assert (hit.metadata.file_path, hit.metadata.chunk_index) == ("notes/raw3.txt", 1), hit.metadata
    print("saved to", vs.save())                                                   # -> $VOITTA_INDEX_DIR
    print("a fresh process with VOITTA_INDEX_DIR set finds", vs.get_collection_info()["points_count"], "chunks again")


if __name__ == "__main__":
    main()
