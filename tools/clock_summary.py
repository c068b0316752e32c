"""Effective shader clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE counter_collection.csv: the counter is summed over
the 8 XCDs, so GHz = GRBM_GUI_ACTIVE / 8 / (End_Timestamp - Start_Timestamp) (MI355X_MICROARCH.md, DVFS give-back; reads high
on dispatches well under 0.3 ms).  Usage: clock_summary.py counter_collection.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
        continue
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if dur <= 0:
        continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    a = acc[name]
    a[0] += 1; a[1] += dur; a[2] += float(r["Counter_Value"])
print(f"{'kernel':44s} {'calls':>6s} {'avg_us':>9s} {'GHz':>6s}")
for name, (n, dur, cyc) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:44]:44s} {n:6d} {dur / n / 1e3:9.1f} {cyc / 8 / dur:6.3f}")
