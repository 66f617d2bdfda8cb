"""Per-kernel HBM bytes and MFMA utilisation of a RAG_GEMM_F16 cross-encoder pass from the rocprofv3 output of
scripts/prof_rerank_f16_pmc.sh: python scripts/derive_f16_pass.py gpurun_out/<tag>/f16 [passes=4] > profiles/…json
FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md §HBM); both counters are in KiB."""
import csv, json, sys, collections
root = sys.argv[1]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4

def short(n):
    n = n.replace("void ragb::", "").replace("ragb::", "")
    return n.split("(")[0][:70]

def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc

fetch, write, mfma = counters(f"{root}/pmc_fetch/rr_counter_collection.csv"), counters(f"{root}/pmc_write/rr_counter_collection.csv"), counters(f"{root}/pmc_mfma/rr_counter_collection.csv")
stats = {short(r["Name"]): r for r in csv.DictReader(open(f"{root}/kstats/rr_kernel_stats.csv"))}
out, tot = [], collections.defaultdict(float)
for k, r in stats.items():
    ms = float(r["TotalDurationNs"]) / 1e6 / passes
    if ms < 0.02: continue
    fb = fetch[k]["FETCH_SIZE"] * 2 * 1024 / passes
    wb = write[k]["WRITE_SIZE"] * 1024 / passes
    busy, gui = mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"], mfma[k]["GRBM_GUI_ACTIVE"]
    # SQ_VALU_MFMA_BUSY_CYCLES sums over the SIMDs (4 per CU x 256 CUs), GRBM_GUI_ACTIVE over the 8 XCDs
    util = busy / 1024 / (gui / 8) if gui else None
    out.append({"kernel": k, "calls_per_pass": int(r["Calls"]) / passes, "ms_per_pass": round(ms, 3), "hbm_read_GB": round(fb / 1e9, 3),
                "hbm_write_GB": round(wb / 1e9, 3), "hbm_TBps": round((fb + wb) / (ms * 1e-3) / 1e12, 2),
                "mfma_busy_frac": None if util is None or busy == 0 else round(util, 3)})
    tot["ms"] += ms; tot["r"] += fb; tot["w"] += wb
print(json.dumps({"passes_averaged": passes, "per_pass": {"ms_sum_of_kernels": round(tot["ms"], 3), "hbm_read_GB": round(tot["r"] / 1e9, 2),
      "hbm_write_GB": round(tot["w"] / 1e9, 2), "hbm_floor_ms_at_6.3TBps": round((tot["r"] + tot["w"]) / 6.3e12 * 1e3, 2)}, "kernels": out}, indent=1))
