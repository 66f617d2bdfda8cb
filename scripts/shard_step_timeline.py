"""Kernel-by-kernel timeline of steady-state batches out of a rocprofv3 --kernel-trace CSV of scripts/bench_shard_step.py:
for each occurrence pattern (anchored on the screening / fp32 scan launch) prints kernel durations and the gap to the
previous kernel's end.  usage: shard_step_timeline.py <kernel_trace.csv>"""
import csv, re, sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:78]))
rows.sort()
scans = [i for i, r in enumerate(rows) if "scan_topk_kernel" in r[2] and (r[1] - r[0]) > 100_000]
# group steady-state batches by the sequence of kernel names between consecutive long scans
seqs = {}
for a, b in zip(scans, scans[1:]):
    key = tuple(r[2] for r in rows[a:b])
    seqs.setdefault(key, []).append((a, b))
for key, occ in sorted(seqs.items(), key=lambda kv: -len(kv[1])):
    if len(occ) < 5:
        continue
    a, b = occ[len(occ) // 2]
    print(f"# pattern seen {len(occ)} times; one batch from the middle:")
    for j in range(a, b):
        s, e, n = rows[j]
        gap = (s - rows[j - 1][1]) / 1e3 if j > a else 0.0
        print(f"  {(e - s) / 1e3:9.2f} us  gap {gap:7.2f}  {n}")
    print(f"  batch period (scan start -> next scan start): {(rows[b][0] - rows[a][0]) / 1e3:.1f} us "
          f"(median over the pattern {sorted((rows[y][0] - rows[x][0]) / 1e3 for x, y in occ)[len(occ) // 2]:.1f})\n")
