#!/usr/bin/env python3
"""Per-counter totals over the full-size dispatches of one kernel in a rocprofv3 counter_collection.csv.
usage: pmc_rows.py <csv> <kernel-name substring>"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2]
rows = [r for r in rows if want in r.get("Kernel_Name", "")]
if not rows:
    sys.exit("no dispatch of %s" % want)
grid = max(int(r["Grid_Size"]) for r in rows)
rows = [r for r in rows if int(r["Grid_Size"]) == grid]          # (the fix-up pass launches a small grid)
by = defaultdict(list)
for r in rows:
    by[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(by.items()):
    print("%-28s dispatches %3d  mean per dispatch %.6g" % (k, len(v), sum(v) / len(v)))
