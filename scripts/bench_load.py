"""Load-time check: FAISSStore.load() of a flat index file (read + host->device), rows/s and GB/s."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flat as oracle
from rag_inference_pipeline_amd import index_io
from rag_inference_pipeline_amd.components.faiss_store import FAISSStore
from rag_inference_pipeline_amd.config import PipelineSettings

n, d = int(os.environ.get("ROWS", 2_000_000)), 768
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "faiss_index.bin")
X = oracle.synth_rows(1234, 0, n, d)
t = time.time(); index_io.write_flat_index(path, X, 0); print(f"write {n}x{d}: {time.time()-t:.1f}s")
for mmap in (False, True):
    s = PipelineSettings(FAISS_INDEX_PATH=path, faiss_dim=d, FAISS_USE_MMAP=str(mmap).lower())
    st = FAISSStore(s)
    t = time.time(); st.load(); el = time.time() - t
    print(f"load mmap={mmap}: {el:.2f}s  {n*d*4/el/1e9:.2f} GB/s  size={st.index_size}")
    D, I = st.search(oracle.synth_rows(4321, 0, 4, d), 5)
    Do, Io = oracle.search(X, oracle.synth_rows(4321, 0, 4, d), 5)
    assert (I == Io).all()
    st.unload()
os.remove(path)
