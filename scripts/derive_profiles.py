"""Derived summaries for profiles/ from the raw rocprofv3 outputs of scripts/collect_bench_profiles.sh and
scripts/prof_rerank_pmc.sh (gpurun_out/<tag>/...):  HBM traffic of the scan launches (FETCH_SIZE x 2, the gfx950
correction of MI355X_MICROARCH.md), the scan kernel's per-dispatch average separated from switched-off launches, and the
MFMA utilisation of the cross-encoder's GEMM kernel.   usage: python scripts/derive_profiles.py <tag> [out_dir]"""
import collections, csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", tag)
out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(root), "..", "profiles")


def rows(path):
    with open(path) as fh:
        return list(csv.DictReader(fh))


# ---- scan HBM traffic
fetch = collections.defaultdict(list)
for r in rows(os.path.join(root, "pmc", "bench_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE" and "scan_topk_kernel" in r["Kernel_Name"]:
        fetch["P1" if r["Kernel_Name"].rstrip(" (ragk::ScanParams)").endswith("1>") else "P0"].append(float(r["Counter_Value"]))
line = json.load(open(os.path.join(root, "bench_under_pmc.json")))
cfgw = line["config"]


def traffic(vals):
    big = [v for v in vals if v > 1e6]          # real launches (KiB); switched-off ones read a few KiB
    return {"fetch_size_kib_mean": sum(big) / len(big), "launches": len(big),
            "traffic_bytes_per_launch": sum(big) / len(big) * 1024 * 2}


t0, t1 = traffic(fetch["P0"]), traffic(fetch["P1"])
doc = {"workload": {"rows": cfgw["rows"], "dim": cfgw["dim"], "batch": cfgw["batch"], "k": cfgw["k"], "n_gpus": line["n_gpus"]},
       "kernel": "scan_topk_kernel<8,1,8,false,0> (one-pass fp32 scan)", **t0,
       "correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane streaming reads; MI355X_MICROARCH.md HBM section)",
       "algorithmic_bytes_per_launch": 4.0 * cfgw["rows"] * cfgw["dim"],
       "two_stage": {"kernel": "scan_topk_kernel<8,1,8,false,1> (fp16 screening pass)", **t1,
                     "algorithmic_bytes_per_launch": 2.0 * cfgw["rows"] * ((cfgw["dim"] + 63) // 64 * 64)},
       "source": f"rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --latency-steps 0 "
                 f"--no-cpu-baseline --no-encoder-leg (scripts/collect_bench_profiles.sh {tag}; rows of profiles/{tag}_scan_pmc_fetch_size.csv)"}
json.dump(doc, open(os.path.join(out_dir, f"{tag}_scan_hbm_traffic.json"), "w"), indent=1)
with open(os.path.join(out_dir, f"{tag}_scan_pmc_fetch_size.csv"), "w") as fh:
    fh.write("Kernel_Name,Counter_Name,Counter_Value_KiB\n")
    for r in rows(os.path.join(root, "pmc", "bench_counter_collection.csv")):
        if r["Counter_Name"] == "FETCH_SIZE" and "scan_topk_kernel" in r["Kernel_Name"]:
            fh.write(f"\"{r['Kernel_Name']}\",FETCH_SIZE,{r['Counter_Value']}\n")

# ---- scan kernel average, real launches only
durs = collections.defaultdict(list)
for r in rows(os.path.join(root, "kstats", "bench_kernel_trace.csv")):
    if "scan_topk_kernel<8, 1, 8, false, 0>" in r["Kernel_Name"]:
        durs["P0"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
real = [d for d in durs["P0"] if d > 1.0]
prof_line = json.load(open(os.path.join(root, "bench_under_rocprof.json")))
json.dump({"kernel": "scan_topk_kernel<8,1,8,false,0>", "real_launches": len(real), "avg_ms_rocprofv3": sum(real) / len(real),
           "other_launches_of_the_same_kernel": len(durs["P0"]) - len(real),
           "avg_ms_hip_events_same_run": prof_line["roofline"]["avg_kernel_ms"],
           "note": f"profiles/{tag}_bench_n1_kernel_stats.csv has one row per kernel: the one-pass scan's row also holds the short launches "
                   "of the same instantiation (sample passes, warm-up searches on a tiny corpus); this file separates the full-corpus "
                   "launches out of the kernel trace of the same run"},
          open(os.path.join(out_dir, f"{tag}_bench_n1_scan_kernel_avg.json"), "w"), indent=1)

# ---- MFMA utilisation of the cross-encoder GEMM
pm = os.path.join(root, "pmc_rerank", "rr_counter_collection.csv")
if os.path.exists(pm):
    acc = collections.defaultdict(float)
    n, tms, name = 0, 0.0, ""
    seen = set()
    for r in rows(pm):
        if "gemm_nt_wl_kernel" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0]
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n += 1
            tms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    n_simd = 256 * 4
    busy_per_simd = acc["SQ_VALU_MFMA_BUSY_CYCLES"] / n_simd
    gpu_cycles = acc["GRBM_GUI_ACTIVE"] / 8.0
    json.dump({"workload": "cross-encoder ms-marco-MiniLM-L-6-v2 architecture (6x384), 3200 (query, doc) pairs, 178405 packed tokens, default "
                           "GEMM mode (two fp16 planes per operand, three products)",
               "kernel": name + " (128x128 tiles, both operands by LDS-DMA, v_mfma_f32_32x32x16_f16 x 3 per fragment pair)",
               "launches": n, "SQ_VALU_MFMA_BUSY_CYCLES_sum": acc["SQ_VALU_MFMA_BUSY_CYCLES"],
               "GRBM_GUI_ACTIVE_sum_over_8_xcds": acc["GRBM_GUI_ACTIVE"], "kernel_time_total_ms": tms,
               "derived": {"mfma_busy_cycles_per_simd": busy_per_simd, "gpu_cycles": gpu_cycles,
                           "mfma_utilisation": busy_per_simd / gpu_cycles, "effective_clock_ghz": gpu_cycles / (tms * 1e-3) / 1e9,
                           "fp16_products_per_fp32_product": 3, "fp32_equivalent_peak_tflops_at_2.4ghz": 2500.0 / 3},
               "source": f"scripts/prof_rerank_pmc.sh {tag}: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace "
                         "-- python3 scripts/prof_rerank.py (counters in their own pass)",
               "note": "round 4: ten of the pass's twelve LayerNorm passes are folded into these GEMMs' epilogues (tile statistics out, "
                       "per-row LayerNorm in); the K loop is round 3's — bound by the L2 -> LDS operand stream (scripts/exp/gemm_wl_bench.hip ablations)"},
              open(os.path.join(out_dir, f"{tag}_rerank_gemm_mfma_util.json"), "w"), indent=1)
# ---- MFMA utilisation of the query encoder's small-batch GEMMs
pe = os.path.join(root, "pmc_encoder", "enc_counter_collection.csv")
if os.path.exists(pe):
    acc = collections.defaultdict(float)
    n, tms = 0, 0.0
    seen = set()
    for r in rows(pe):
        if "gemm_nt_ws_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n += 1
            tms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    busy_per_simd = acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4)
    gpu_cycles = acc["GRBM_GUI_ACTIVE"] / 8.0
    json.dump({"workload": "query encoder, bge-base-en-v1.5 architecture (12x768), 32 queries = 447 packed tokens, default GEMM mode",
               "kernel": "gemm_nt_ws_kernel<1|2, 4> (64-row tiles, sixteen waves, split-K for the N = H GEMMs)",
               "launches": n, "SQ_VALU_MFMA_BUSY_CYCLES_sum": acc["SQ_VALU_MFMA_BUSY_CYCLES"],
               "GRBM_GUI_ACTIVE_sum_over_8_xcds": acc["GRBM_GUI_ACTIVE"], "kernel_time_total_ms": tms,
               "derived": {"mfma_busy_cycles_per_simd": busy_per_simd, "gpu_cycles": gpu_cycles,
                           "mfma_utilisation": busy_per_simd / gpu_cycles if gpu_cycles else None},
               "source": f"scripts/prof_encoder_pmc.sh {tag}: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- "
                         "python3 scripts/prof_encoder.py (counters in their own pass)",
               "note": "a 447-token batch is launch- and operand-latency-bound (84 dependent launches of 5-26 us); the matrix pipe idles"},
              open(os.path.join(out_dir, f"{tag}_encoder_mfma_util.json"), "w"), indent=1)
print("ok")
