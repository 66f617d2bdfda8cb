"""Where a scan launch's fixed cost goes: per-workgroup s_memrealtime stamps (experiment build -DRAGK_STAMPS)
at entry / prologue done / round 0 done / main loop done / exit, relative to the first workgroup's entry."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scripts._sidelib import build
os.environ["RAG_AMD_LIB"] = build("STAMPS", ["RAGK_STAMPS"])
import numpy as np
from oracle import flat as oracle
from rag_inference_pipeline_amd import _native
from rag_inference_pipeline_amd.flat_index import FlatIndex, SCREEN_FP16

d = 768
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
Q = oracle.synth_rows(4321, 0, 32, d)
lib = _native.lib()
lib.rag_debug_scan_stamps.restype = C.c_int
lib.rag_debug_scan_stamps.argtypes = [C.c_void_p, C.c_int32]
lib.rag_debug_resolve_stamps.restype = C.c_int
lib.rag_debug_resolve_stamps.argtypes = [C.c_void_p, C.c_int32]
idx = FlatIndex(d); idx.add_synthetic(rows, 1234)
for mode in ("one-pass", "two-stage"):
    if mode == "two-stage":
        idx.set_screening(SCREEN_FP16)
    for _ in range(5):
        idx.search(Q, 10)
    agg = []
    for rep in range(5):
        idx.search(Q, 10)
        st = np.zeros((256, 8), dtype=np.uint64)
        assert lib.rag_debug_scan_stamps(st.ctypes.data, 256) == 0
        t = (st[:, :5].astype(np.int64) - int(st[:, 0].min())) / 100.0   # us
        agg.append(t)
    t = np.median(np.stack(agg), axis=0)
    names = ["entry", "prologue done", "round 0 done", "loop done", "exit"]
    print(mode, f"rows={rows}")
    for i, nme in enumerate(names):
        print(f"  {nme:14s} min {t[:, i].min():8.2f}  median {np.median(t[:, i]):8.2f}  max {t[:, i].max():8.2f} us")
    print(f"  prologue (median) {np.median(t[:,1]-t[:,0]):.2f} us; epilogue (median) {np.median(t[:,4]-t[:,3]):.2f} us; "
          f"loop-done spread {t[:,3].max()-t[:,3].min():.2f} us (p10 {np.percentile(t[:,3],10):.1f} p90 {np.percentile(t[:,3],90):.1f})", flush=True)
    # who finishes late: by XCD (workgroups are dealt round-robin, b % 8) and by tail-round membership
    done = t[:, 3]
    print("  loop-done by b % 8:", " ".join(f"{done[x::8].mean():.1f}" for x in range(8)))
    n_tiles = (rows + 31) // 32
    left = n_tiles - (n_tiles // 2048) * 2048
    if 0 < left < 256:
        print(f"  loop-done, workgroups with a leftover tile (b < {left}): {done[:left].mean():.1f}; without: {done[left:].mean():.1f}")
    if mode == "two-stage":
        rs = np.zeros((32, 8), dtype=np.uint64)
        assert lib.rag_debug_resolve_stamps(rs.ctypes.data, 32) == 0
        rt = (rs[:, :6].astype(np.int64) - int(st[:, 0].min())) / 100.0
        names_r = ["entry", "staged", "k-th key", "band", "scored", "exit"]
        print("  resolve kernel (us after the scan's first entry; median over 32 workgroups): " +
              ", ".join(f"{nm} {np.median(rt[:, i]):.1f}" for i, nm in enumerate(names_r)))
        print("  resolve phases (median us): " + ", ".join(f"{names_r[i+1]} {np.median(rt[:, i+1]-rt[:, i]):.2f}" for i in range(5)))
    q = np.argsort(done)
    print("  ten latest workgroups:", q[-10:].tolist(), " ten earliest:", q[:10].tolist(), flush=True)
