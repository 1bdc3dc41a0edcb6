#!/usr/bin/env python3
"""Print the kernel timeline (start, duration, gap) of the last N `mi::` dispatches of a
rocprofv3 --kernel-trace CSV, plus per-kernel averages over non-trivial (>1 us) dispatches."""
import csv
import sys
from collections import defaultdict

path, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = [r for r in csv.DictReader(open(path)) if "mi::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = defaultdict(list)
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(d)
print("kernel, calls, calls>1us, avg_ns(>1us), avg_ns(all)")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    big = [d for d in v if d > 1000]
    print(f"{k:45s} {len(v):6d} {len(big):6d} {sum(big) / max(1, len(big)):10.0f} {sum(v) / len(v):10.0f}")
t = rows[-last:]
t0 = int(t[0]["Start_Timestamp"])
prev = None
for r in t:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = s - prev if prev is not None else 0
    print(f"{s / 1e3:9.1f} us dur {(e - s) / 1e3:7.2f} gap {gap / 1e3:7.2f}  {r['Kernel_Name'].split('(')[0].replace('void mi::', '')}")
    prev = e

# one whole solve in the middle of the trace (k_solve_begin ... k_solve_end) and the host turnaround to the next one
is_begin = lambda r: "k_solve_begin" in r["Kernel_Name"] or "k_entry_zero" in r["Kernel_Name"]   # noqa: E731
begins = [i for i, r in enumerate(rows) if is_begin(r)]
# prefer a solve of the folded loop (the headline): an entry kernel directly followed by k_gemv_pcg, or by the skipped S*x0
folded = [i for i in begins if i + 3 < len(rows) and any("k_gemv_pcg" in rows[i + k]["Kernel_Name"] for k in (1, 2, 3))]
if len(begins) > 4:
    pool = folded if len(folded) > 4 else begins
    b = pool[len(pool) // 2]
    nb = next((i for i in begins if i > b), len(rows))
    print("\none solve (k_solve_begin .. next k_solve_begin):")
    t0, prev = int(rows[b]["Start_Timestamp"]), None
    for r in rows[b:nb + 1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        gap = s - prev if prev is not None else 0
        print(f"{s / 1e3:9.1f} us dur {(e - s) / 1e3:7.2f} gap {gap / 1e3:7.2f}  {r['Kernel_Name'].split('(')[0].replace('void mi::', '')}")
        prev = e
    turn = []
    for i in begins[1:]:
        if "k_solve_end" in rows[i - 1]["Kernel_Name"] and i in pool:
            turn.append(int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]))
    if turn:
        turn.sort()
        print(f"host turnaround k_solve_end -> next k_solve_begin: median {turn[len(turn) // 2] / 1e3:.1f} us, min {turn[0] / 1e3:.1f} us over {len(turn)} solves")
