"""A/B of the fp16 screening pass at 8 and 12 waves per workgroup (tuning build; 12 waves: +1 % at 10M rows,
+2.4 % at 1.25M — not adopted, it would double the P = 1 kernel instantiations for that)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts._sidelib import build
os.environ["RAG_AMD_LIB"] = build("TUNING", ["RAGK_TUNING"])
from rag_inference_pipeline_amd.flat_index import FlatIndex, SCREEN_FP16
from oracle import flat as oracle
for N in (10_000_000, 1_250_000):
    idx = FlatIndex(768); idx.add_synthetic(N, 1234); idx.set_screening(SCREEN_FP16)
    Q = oracle.synth_rows(4321, 0, 32, 768)
    ref = None
    res = {8: [], 12: []}
    for rnd in range(4):
        for w in (8, 12):
            os.environ["RAG_AMD_SCREEN_WAVES"] = str(w)
            D, I = idx.search(Q, 10)
            if ref is None: ref = (D.copy(), I.copy())
            assert (I == ref[1]).all() and (D == ref[0]).all()
            idx.profile_enable(True); idx.profile(reset=True)
            for _ in range(8): idx.search(Q, 10)
            ms, n = idx.profile(reset=True); idx.profile_enable(False)
            res[w].append(ms / n)
    for w, v in res.items():
        print(f"N={N} screen pass waves={w}: median {np.median(v):.4f} ms  -> {2.0*N*768/np.median(v)/1e6:.0f} GB/s", flush=True)
    idx.close()
