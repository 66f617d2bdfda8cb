"""Interleaved A/B timing of scan-kernel launch geometries (one process, one index)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts._sidelib import build
os.environ["RAG_AMD_LIB"] = build("TUNING", ["RAGK_TUNING"])  # 12/16-wave and ring-12/16 variants + env overrides
from rag_inference_pipeline_amd.flat_index import FlatIndex
from oracle import flat as oracle

N = int(os.environ.get("TUNE_ROWS", 10_000_000)); d = int(os.environ.get("TUNE_DIM", 768)); k = int(os.environ.get("TUNE_K", 10))
idx = FlatIndex(d); idx.add_synthetic(N, 1234)
Q = oracle.synth_rows(4321, 0, 32, d)
configs = [(8, 8), (8, 12), (8, 16), (12, 8), (16, 8), (12, 12), (8, 4)]
ref = None
res = {c: [] for c in configs}
for rnd in range(4):
    for (w, r) in configs:
        os.environ["RAG_AMD_SCAN_WAVES"] = str(w); os.environ["RAG_AMD_SCAN_RING"] = str(r)
        D, I = idx.search(Q, k)
        if ref is None: ref = (D.copy(), I.copy())
        assert (I == ref[1]).all() and (D == ref[0]).all(), (w, r)
        idx.profile_enable(True); idx.profile(reset=True)
        for _ in range(5): idx.search(Q, k)
        ms, n = idx.profile(reset=True); idx.profile_enable(False)
        res[(w, r)].append(ms / n)
for c, v in res.items():
    v = np.array(v); gb = 4.0 * N * d / (np.median(v) * 1e-3) / 1e9
    print(f"waves={c[0]:2d} ring={c[1]}: median {np.median(v):.3f} ms  min {v.min():.3f}  -> {gb:.0f} GB/s ({gb/8000:.1%})", flush=True)
