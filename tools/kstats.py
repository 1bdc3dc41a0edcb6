#!/usr/bin/env python3
"""Print name / calls / total us / avg us of the mi:: kernels of a rocprofv3 *_kernel_stats.csv."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mi::" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].split("(")[0].replace("void ", "")
    print(f"{name[:60]:60s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs'])/1e3:10.1f} us  avg {float(r['AverageNs'])/1e3:8.2f} us")
