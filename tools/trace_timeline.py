"""Diagnostic: per-kernel timeline of the LAST complete period in a rocprofv3 kernel-trace CSV: start offset,
duration and the gap to the previous kernel's end.  usage: trace_timeline.py trace.csv anchor_kernel_substring"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
# period = from one anchor occurrence to the next; take a late one
per = [(idx[i], idx[i + 1]) for i in range(len(idx) - 1)]
a, b = per[-3]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
tot_k = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:90]}")
    prev_end = max(prev_end, e); tot_k += e - s
print(f"period {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernel time {tot_k / 1e3:.1f} us, nodes {b - a}")
