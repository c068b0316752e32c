#!/bin/bash
# rocprofv3 per-kernel average launch times of the bench workload (2 single-stream rollouts) for every ab_build/lib*.so, side by side
export TMPDIR=/tmp
for f in ab_build/lib*.so; do
  v=$(basename $f .so); O=$(pwd)/gpurun_out/ks_$v; rm -rf $O; mkdir -p $O
  NNJ_LIB_PATH=$(pwd)/$f rocprofv3 --kernel-trace --stats -d $O/s --output-format csv -- python3 $(pwd)/tools/prof_run.py 256 50 1024 2 > $O/log.txt 2>&1
  cp $(find $O/s -name "*kernel_stats.csv" | head -1) gpurun_out/ks_$v.csv; rm -rf $O
done
python3 - <<'PY'
import csv, glob, re
tabs = {}
for f in sorted(glob.glob("gpurun_out/ks_lib*.csv")):
    v = re.search(r"ks_lib(\w+)\.csv", f).group(1)
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Name"])
        tabs.setdefault(n, {})[v] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
vs = sorted({v for t in tabs.values() for v in t})
print("kernel".ljust(46), " ".join(f"{v:>10s}" for v in vs), "  (total ms over 2 rollouts)")
for n, t in sorted(tabs.items(), key=lambda kv: -max(x[1] for x in kv[1].values())):
    if max(x[1] for x in t.values()) < 1.0: continue
    print(n[:46].ljust(46), " ".join(f"{t.get(v, (0, 0))[1]:10.2f}" for v in vs))
PY
