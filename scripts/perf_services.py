"""Host-side cost per chunk of the drop-in service API (embed_texts -> sparse embed_texts -> store_chunks,
the three calls of IndexingService._index_file_standard) with a tiny synthetic model, so that GPU time is
negligible and what remains is tokenisation + the Python lists the reference's API prescribes."""
import os, sys, tempfile, time, pathlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_services_gpu as T  # the synthetic checkpoint helper

tmp = pathlib.Path(tempfile.mkdtemp())
path, shape, w, vocab = T._checkpoint(tmp, "mini", "mean")
os.environ["EMBEDDING_MODEL"] = path
os.environ["EMBEDDING_DIMENSION"] = str(shape.hidden)
from voitta_rag_amd.embedding import get_embedding_service
from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
from voitta_rag_amd.vector_store import ChunkMetadata, get_vector_store
emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
rng = np.random.default_rng(0)
texts = [" ".join(rng.choice(T.WORDS, size=70)) for _ in range(4096)]
metas = [ChunkMetadata(file_path=f"f{i // 64}.md", folder_path="d", index_folder="d", file_name="f.md", chunk_index=i % 64,
                       total_chunks=64, start_char=0, end_char=9, indexed_at="2026-01-01T00:00:00") for i in range(len(texts))]
emb.embed_texts(texts[:64]); sp.embed_texts(texts[:64])
t0 = time.perf_counter(); e = emb.embed_texts(texts); t1 = time.perf_counter()
s = sp.embed_texts(texts); t2 = time.perf_counter()
vs.store_chunks(list(zip(texts, e, metas)), sparse_vectors=s); t3 = time.perf_counter()
n = len(texts)
print(f"embed_texts {n / (t1 - t0):.0f}/s  sparse embed_texts {n / (t2 - t1):.0f}/s  store_chunks {n / (t3 - t2):.0f}/s  "
      f"all three {n / (t3 - t0):.0f} chunks/s (H={shape.hidden}, host side)")
