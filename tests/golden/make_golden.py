"""Regenerates tests/golden/flat_search.json with the CPU oracle (oracle/flat_oracle.c).

The reference itself cannot produce vectors for this path: its arithmetic is faiss-cpu 1.13.1,
which is not installed (and not vendored), and its own tests hold no search fixtures
(SURVEY.md §8c).  So these are *oracle* outputs on inputs from the deterministic synthetic
generator (rago_synth_rows: integer hash -> exact fp32), cross-checked at generation time against
an independent float64 numpy ranking.  They pin (a) the oracle against regressions and (b) the HIP
path against a committed answer.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import flat as oracle  # noqa: E402

CASES = [
    # name, corpus seed, N, d, query seed, nq, k, metric
    ("config_A_10k_384_b1", 1234, 10_000, 384, 4321, 1, 10, 0),
    ("small_B_4096_768_b32_k10", 1234, 4096, 768, 4321, 32, 10, 0),
    ("small_B_4096_768_b32_k100", 1234, 4096, 768, 4321, 32, 100, 0),
    ("small_C_20000_384_b32_k100", 77, 20_000, 384, 78, 32, 100, 0),
    ("l2_5000_384_b7", 5, 5000, 384, 6, 7, 10, 1),
    ("odd_dim_777_100_b3", 9, 777, 100, 10, 3, 10, 0),
    ("k_gt_n_5_64", 11, 5, 64, 12, 2, 8, 0),
]


def main() -> None:
    out = {}
    for name, cs, N, d, qs, nq, k, metric in CASES:
        X = oracle.synth_rows(cs, 0, N, d)
        Q = oracle.synth_rows(qs, 0, nq, d)
        D, I = oracle.search(X, Q, k, metric)
        # independent float64 check of the ranking (skipping fp32-noise near-ties)
        X64, Q64 = X.astype(np.float64), Q.astype(np.float64)
        S = Q64 @ X64.T if metric == 0 else -((Q64[:, None, :] - X64[None]) ** 2).sum(-1)
        order = np.argsort(-S, axis=1, kind="stable")[:, :k]
        for b in range(nq):
            kk = min(k, N)
            sorted_s = S[b, order[b, :kk]]
            gaps = np.abs(np.diff(sorted_s))
            if gaps.size == 0 or gaps.min() > 1e-6:
                assert (order[b, :kk] == I[b, :kk]).all(), (name, b)
        out[name] = {
            "corpus_seed": cs, "n": N, "d": d, "query_seed": qs, "nq": nq, "k": k, "metric": metric,
            "ids": I.tolist(),
            "scores_bits": D.view(np.uint32).tolist(),  # exact fp32 bit patterns
        }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flat_search.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
