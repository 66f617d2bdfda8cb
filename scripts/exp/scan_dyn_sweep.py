"""Dynamic deal of the scan's last rounds: step time at shard size against (rounds dealt by ticket, single-tile
tickets per workgroup).  The knobs are read when an index is created (RAG_AMD_SCAN_DYN_ROUNDS / _SINGLES)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import flat as oracle
from rag_inference_pipeline_amd.flat_index import FlatIndex, SCREEN_FP16, SEARCH_DEFER_FALLBACK

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
d, B, k = 768, 32, 10
Q = torch.from_numpy(oracle.synth_rows(4321, 0, B, d)).cuda()
out_s = torch.empty((B, k), dtype=torch.float32, device="cuda")
out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
sptr = torch.cuda.current_stream().cuda_stream
combos = [tuple(int(v) for v in c.split(',')) for c in os.environ.get('DYN_COMBOS', '0,0 1,0 2,0 3,0 4,0 6,0 10,0 1000,0').split()]
ref = None
res = []
for rounds, singles in combos:
    os.environ["RAG_AMD_SCAN_DYN_ROUNDS"] = str(rounds)
    os.environ["RAG_AMD_SCAN_DYN_SINGLES"] = str(singles)
    idx = FlatIndex(d); idx.add_synthetic(rows, 1234)
    row = {"dyn_rounds": rounds, "singles_per_wg": singles}
    for mode in ("one_pass", "two_stage"):
        if mode == "two_stage":
            idx.set_screening(SCREEN_FP16)
        fn = lambda: idx.search_device_ex(Q.data_ptr(), B, k, out_s.data_ptr(), out_i.data_ptr(), SEARCH_DEFER_FALLBACK,
                                          flag.data_ptr(), sptr)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            idx.profile_enable(True); idx.profile(reset=True)
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / steps * 1e3
            ms, n = idx.profile(reset=True); idx.profile_enable(False)
            best = min(best, el)
        row[mode] = {"step_ms": round(best, 4), "scan_kernel_ms": round(ms / n, 4)}
        got = (out_i.cpu().numpy().copy(), out_s.cpu().numpy().copy())
        if ref is None:
            ref = got
        row[mode]["identical"] = bool(np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32)))
    idx.close()
    res.append(row)
    print(json.dumps(row), flush=True)
