"""Print the kernel timeline of one steady-state batch from a rocprofv3 (rocpd sqlite) trace."""
import re, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "screen_finalize"
rows = list(db.cursor().execute("select name, start, end, grid_x from kernels order by start"))
seq = [(re.sub(r"\(.*", "", n)[:70], s, e, g) for n, s, e, g in rows]
idxs = [i for i, r in enumerate(seq) if anchor in r[0]]
for pick in (len(idxs) // 3, len(idxs) - 3):
    i = idxs[pick]
    for j in range(max(i - 5, 1), min(i + 5, len(seq))):
        n, s, e, g = seq[j]
        print(f"{n:72s} dur {(e-s)/1e3:8.1f} us  gap_before {(s-seq[j-1][2])/1e3:7.1f} us grid {g}")
    print()
for i, (n, s, e, g) in enumerate(seq):
    if "tournament_merge" in n and (e - s) > 6e3:
        print("a long merge:", (e - s) / 1e3, "us after", seq[i - 1][0][:50])
        break
