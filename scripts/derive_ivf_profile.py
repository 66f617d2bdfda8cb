"""profiles/rNN_ivf_nprobe.json from one rocprofv3 --kernel-trace run of scripts/bench_ivf.py:

    python scripts/derive_ivf_profile.py <bench_ivf stdout json> <..._kernel_trace.csv> [more bench_ivf json files ...]

The trace is cut into searches (one ivf_plan_kernel launch per pass of <= 32 queries); the list-scan launches are
matched to the batch sizes in the order bench_ivf.py runs them (warmup + steps per batch size), so each batch size
gets the average duration of ITS ivf_batch_scan_kernel launches and — against the union bytes bench_ivf.py counted —
the kernel's own GB/s next to the whole search's.
"""
import csv
import json
import sys


def last_json(path):
    lines = [ln for ln in open(path).read().strip().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


def main() -> None:
    run = last_json(sys.argv[1])
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # us
    names = ("query_sqnorm_kernel", "0, false>(ragk::ScanParams)", "ivf_coarse_select_kernel", "ivf_plan_kernel",
             "1, true>(ragk::ScanParams)", "screen_resolve_kernel", "ivf_batch_scan_kernel", "tournament_merge_kernel")
    # the list scan: the exact kernel, or — two-stage search — the flat scan kernel's screening pass in MAP mode (its exact
    # fallback launches are no-ops of a few microseconds)
    two = any(b.get("two_stage") for b in run["batches"].values())
    is_scan = (lambda nm: "scan_topk_kernel" in nm and "1, true>(" in nm) if two else (lambda nm: "ivf_batch_scan_kernel" in nm)
    scans = [dur(r) for r in rows if is_scan(r["Kernel_Name"])]
    per_kernel = {n: [dur(r) for r in rows if n in r["Kernel_Name"]] for n in names}
    batches = list(run["batches"].items())
    total_passes = sum(-(-int(nq) // 32) for nq, _ in batches)
    per = len(scans) // total_passes   # warmup + steps launches per pass
    out = {"source": "rocprofv3 --kernel-trace of scripts/bench_ivf.py (see profiles/README.md)", "workload": run["workload"],
           "list_rows": run["list_rows"], "hbm_peak_gbps": 8000.0, "batches": {}}
    at = 0
    for nq, b in batches:
        passes = -(-int(nq) // 32)
        mine = scans[at:at + per * passes]
        at += per * passes
        mine = mine[len(mine) // 4:]   # drop the warmup launches (and then some)
        scan_us = sum(mine) / len(mine) * passes
        out["batches"][nq] = {
            "whole_search_ms": b["ms_per_batch"], "whole_search_gbps_vs_union": b["search_gbps_vs_union"],
            "two_stage": bool(b.get("two_stage")),
            "list_scan_kernel_us": round(scan_us, 1), "union_bytes": b["union_bytes"],
            "scan_bytes": b.get("scan_bytes", b["union_bytes"]),
            "list_scan_gbps": round(b.get("scan_bytes", b["union_bytes"]) / scan_us / 1e3, 1),
            "list_scan_frac_of_hbm_peak": round(b.get("scan_bytes", b["union_bytes"]) / scan_us / 1e3 / 8000.0, 4),
            "corpus_fraction_read": b["corpus_fraction_read"], "pair_bytes": b["pair_bytes"],
        }
    label = {"0, false>(ragk::ScanParams)": "scan_topk_kernel<P=0> (coarse quantizer, sums parked)",
             "1, true>(ragk::ScanParams)": "scan_topk_kernel<P=1, MAP> (fp16 screening pass over the probed lists)",
             "ivf_batch_scan_kernel": "ivf_batch_scan_kernel (exact list scan; a no-op fallback launch in the two-stage search)"}
    out["kernels_avg_us_all_launches"] = {label.get(n, n): round(sum(v) / len(v), 1) for n, v in per_kernel.items() if v}
    for extra in sys.argv[3:]:
        e = last_json(extra)
        out.setdefault("other_runs", []).append({k: e[k] for k in ("workload", "list_rows", "batches") if k in e}
                                                | {k: e[k] for k in ("exhaustive_flat_ms_per_batch", "exhaustive_flat_two_stage_ms_per_batch") if k in e})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
