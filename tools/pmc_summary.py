#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one row per dispatch and counter) into per-kernel averages."""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        m = re.search(r"engine2_kernel<ddrl::(\w+)>|ddrl::(\w+)\(", k)
        name = (m.group(1) or m.group(2)) if m else k[:40]
        # acting-size launches (grid much smaller) are kept apart
        big = int(row.get("Grid_Size", 0) or 0) >= 256 * 2000
        key = name + ("" if big else ":small")
        c = agg[key][row["Counter_Name"]]
        c[0] += float(row["Counter_Value"])
        c[1] += 1
names = sorted({c for k in agg for c in agg[k]})
print("kernel," + ",".join(names))
for k in sorted(agg):
    print(k + "," + ",".join("%.4g" % (agg[k][c][0] / max(1, agg[k][c][1])) if c in agg[k] else "" for c in names))
