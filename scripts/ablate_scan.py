import os, sys, subprocess
sys.path.insert(0, ".")
code = r'''
import os, sys
sys.path.insert(0, ".")
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
for variant, lib in [("product", ""), ("NO_MFMA", "rag_inference_pipeline_amd/csrc/exp/librag_amd_NO_MFMA.so"),
                     ("L2_WINDOW", "rag_inference_pipeline_amd/csrc/exp/librag_amd_L2_WINDOW.so")]:
    env = dict(os.environ, VARIANT=variant)
    if lib: env["RAG_AMD_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
