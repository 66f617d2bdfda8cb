import os
"""Two-stage search vs the one-pass fp32 scan on the same synthetic corpus: wall time per batch,
stage-1 kernel time (HIP events), certificate counters, and a result comparison."""
import sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_inference_pipeline_amd.flat_index import FlatIndex, SCREEN_FP16, SCREEN_OFF
from oracle import flat as oracle


def timed(idx, Q, k, steps):
    idx.search(Q, k)
    idx.profile(reset=True)
    idx.profile_enable(True)
    t = time.time()
    for _ in range(steps):
        D, I = idx.search(Q, k)
    wall = (time.time() - t) / steps
    ms, n = idx.profile(reset=True)
    idx.profile_enable(False)
    return wall, ms / max(n, 1), D, I


def run(N, d, nq, k, steps=20):
    idx = FlatIndex(d)
    idx.add_synthetic(N, 1234)
    Q = oracle.synth_rows(4321, 0, nq, d)
    w0, k0, D0, I0 = timed(idx, Q, k, steps)
    t = time.time(); idx.set_screening(SCREEN_FP16); t_conv = time.time() - t
    w1, k1, D1, I1 = timed(idx, Q, k, steps)
    st = idx.screen_stats()
    same = bool(np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32)))
    d64 = (d + 63) // 64 * 64
    print(f"N={N} d={d} nq={nq} k={k}: fp32 {w0*1e3:.3f} ms/batch (kernel {k0:.3f}, {4.0*N*d/k0/1e6:.0f} GB/s) | "
          f"two-stage {w1*1e3:.3f} ms/batch (stage-1 kernel {k1:.3f}, {2.0*N*d64/k1/1e6:.0f} GB/s of fp16 rows) "
          f"speedup {w0/w1:.2f}x identical={same} convert {t_conv:.2f}s stats={st}", flush=True)
    idx.close()


if __name__ == "__main__":
    big = len(sys.argv) > 1 and sys.argv[1] == "big"
    run(1_000_000, 384, 32, 10)
    run(1_250_000, 768, 32, 10)
    run(1_000_000, 768, 32, 100)
    if big:
        run(10_000_000, 768, 32, 10, steps=10)
        run(10_000_000, 768, 32, 100, steps=5)
        run(10_000_000, 768, 1, 10, steps=10)
