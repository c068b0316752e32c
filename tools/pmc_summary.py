#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv (SQ wave/issue counters).
usage: python3 tools/pmc_summary.py <counter_collection.csv>
Ratios are per wave-cycle: wait_any / wait_inst / valu / lds / vmem; mfma = SQ_VALU_MFMA_BUSY_CYCLES /
(4 * SQ_BUSY_CU_CYCLES) (one MFMA pipe per SIMD)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        calls[k] += 1
print(f"{'kernel':34s} {'calls':>5s} {'wait_any':>8s} {'wait_inst':>9s} {'valu':>6s} {'lds':>6s} {'vmem':>6s} {'mfma':>6s} {'waves/simd':>10s}")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    busy = c.get("SQ_BUSY_CU_CYCLES", 0) or 1
    f = lambda n: c.get(n, 0) / wc
    print(f"{k[:34]:34s} {calls[k]:5d} {f('SQ_WAIT_ANY'):8.2f} {f('SQ_WAIT_INST_ANY'):9.2f} {f('SQ_ACTIVE_INST_VALU'):6.2f} "
          f"{f('SQ_ACTIVE_INST_LDS'):6.2f} {f('SQ_ACTIVE_INST_VMEM'):6.2f} {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * busy):6.2f} {wc / (4 * busy):10.2f}")
