#!/usr/bin/env python3
"""tools/pmc_kcount.py DIR -- per k_count dispatch, the counters of a rocprofv3 --pmc run (counter_collection CSV)."""
import csv, glob, os, sys
from collections import OrderedDict
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
rows = OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_count<" not in r["Kernel_Name"]:
        continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in rows.values() for k in v})
print("dispatch " + " ".join(f"{n:>22s}" for n in names))
for d, v in rows.items():
    print(f"{d:>8s} " + " ".join(f"{v.get(n, 0):22.0f}" for n in names))
