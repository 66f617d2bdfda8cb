import os
"""Scratch timing of the flat scan (kernel time from HIP events + wall)."""
import sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_inference_pipeline_amd.flat_index import FlatIndex
from oracle import flat as oracle

def run(N, d, nq, k, steps=20):
    idx = FlatIndex(d)
    t = time.time(); idx.add_synthetic(N, 1234); t_fill = time.time() - t
    Q = oracle.synth_rows(4321, 0, nq, d)
    idx.search(Q, k)
    idx.profile_enable(True)
    t = time.time()
    for _ in range(steps):
        D, I = idx.search(Q, k)
    wall = (time.time() - t) / steps
    ms, n = idx.profile(reset=True)
    kms = ms / n
    gbs = 4.0 * N * d / (kms * 1e-3) / 1e9
    print(f"N={N} d={d} nq={nq} k={k}: fill {t_fill:.2f}s  scan kernel {kms:.3f} ms  {gbs:.0f} GB/s ({gbs/8000:.1%} of 8TB/s)  wall/batch {wall*1e3:.3f} ms  qps {nq/wall:.0f}", flush=True)
    idx.close()

if __name__ == "__main__":
    run(100_000, 768, 32, 10)
    run(1_000_000, 384, 32, 10)
    run(1_000_000, 768, 32, 10)
    run(10_000_000, 768, 32, 10, steps=10)
    run(1_000_000, 384, 32, 100)
