"""Batch-size sweep of the search at 10M x 768, k = 10 (the gateway-side callers send chunks of a
batch — 8 of 32 — through the same kernels): host queries in, results on host (rag_index_search), one-pass
and two-stage.  The scan streams the corpus once whatever the batch, so latency is flat up to 32 queries and
queries/s scale with the batch; beyond 32 the batch takes one pass per 32 queries."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import flat as oracle
from rag_inference_pipeline_amd.flat_index import FlatIndex, SCREEN_FP16

N, d, k = int(os.environ.get("SWEEP_ROWS", 10_000_000)), 768, 10
idx = FlatIndex(d); idx.add_synthetic(N, 1234)
out = {"rows": N, "dim": d, "k": k, "path": "rag_index_search (host queries -> results on host)", "modes": {}}
for mode in ("one_pass", "two_stage"):
    if mode == "two_stage":
        idx.set_screening(SCREEN_FP16)
    rows = []
    for B in (1, 4, 8, 16, 32, 64):
        Q = oracle.synth_rows(4321, 0, B, d)
        for _ in range(3):
            idx.search(Q, k)
        per = []
        for _ in range(30):
            t = time.perf_counter(); idx.search(Q, k); per.append(time.perf_counter() - t)
        p50 = float(np.median(per))
        rows.append({"batch": B, "p50_ms": round(p50 * 1e3, 3), "queries_per_s": round(B / p50)})
    out["modes"][mode] = rows
print(json.dumps(out), flush=True)
