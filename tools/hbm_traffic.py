#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM-side traffic of the
dominant kernels and write profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

Units and corrections follow /opt/skills/guides (MI355X_MICROARCH.md §HBM, cdna_hip_programming.md §7):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read (16 B/lane), so the read side is doubled; WRITE_SIZE is exact. Infinity-Cache
hits are counted (the counters sit on the L2's memory side), so this is L2-miss traffic.

    python tools/hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB -> bytes; "
                     "FETCH_SIZE x2 (gfx950 wide-read correction); median over dispatches that do work"}
    for k in sorted(f):
        if "mi::" not in k:
            continue
        # launches that return at once (stop flag, x0 == 0 shortcut) fetch ~nothing: median over the dispatches that
        # move at least half of what the largest one moves
        def working_median(vals):
            vals = sorted(vals)
            if not vals:
                return 0.0
            work = [v for v in vals if v >= 0.5 * vals[-1]]
            return work[len(work) // 2]
        fm = working_median(f[k])
        wm = working_median(w.get(k, [0.0]))
        fv = f[k]
        res[k] = {"dispatches": len(fv), "fetch_KiB_raw": fm, "write_KiB_raw": wm,
                  "bytes_per_launch": int((2.0 * fm + wm) * 1024)}
    # dominant kernel of the timed region = the folded PCG GEMV launches (both phases); else the plain S-apply GEMV
    fold = [v["bytes_per_launch"] for k, v in res.items() if isinstance(v, dict) and "k_gemv_pcg" in k]
    plain = [v["bytes_per_launch"] for k, v in res.items() if isinstance(v, dict) and "k_gemv_batched" in k and "false" in k]
    if fold:
        res["dominant_kernel_bytes_per_launch"] = int(sum(fold) / len(fold))
    elif plain:
        res["dominant_kernel_bytes_per_launch"] = plain[0]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
