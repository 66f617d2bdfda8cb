"""Ablations of the scan kernel: matrix work removed (load stream only) / corpus cache-resident
(matrix pipe only), timed against the product build, one subprocess per library."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts._sidelib import build
os.environ["RAG_REPO_ROOT"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, os.environ["RAG_REPO_ROOT"])
import numpy as np
from rag_inference_pipeline_amd.flat_index import FlatIndex
from oracle import flat as oracle
N, d = 10_000_000, 768
idx = FlatIndex(d); idx.add_synthetic(N, 1234)
Q = oracle.synth_rows(4321, 0, 32, d)
for w, r in [(8, 8), (16, 8), (8, 4)]:
    os.environ["RAG_AMD_SCAN_WAVES"] = str(w); os.environ["RAG_AMD_SCAN_RING"] = str(r)
    idx.search(Q, 10)
    idx.profile_enable(True); idx.profile(reset=True)
    for _ in range(8): idx.search(Q, 10)
    ms, n = idx.profile(reset=True); idx.profile_enable(False)
    print(f"  {os.environ.get('VARIANT')}: waves={w} ring={r}: {ms/n:.3f} ms -> {4.0*N*d/(ms/n*1e-3)/1e9:.0f} GB/s-equivalent", flush=True)
'''
for variant, lib in [("product", build("TUNING", ["RAGK_TUNING"])),
                     ("NO_MFMA", build("NO_MFMA", ["RAGK_TUNING", "RAGK_ABLATE_NO_MFMA"])),
                     ("L2_WINDOW", build("L2_WINDOW", ["RAGK_TUNING", "RAGK_ABLATE_L2_WINDOW"]))]:
    env = dict(os.environ, VARIANT=variant)
    if lib: env["RAG_AMD_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
