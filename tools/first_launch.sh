#!/bin/bash
# duration of the FIRST launch of a kernel in each of two rollouts (layer 0 of the encoder: real data whatever a timing-only
# knock-out does to the later layers), per ab_build/lib*.so: tools/first_launch.sh k_row_s
K=${1:-k_row_s}
export TMPDIR=/tmp
for f in ab_build/lib*.so; do
  v=$(basename $f .so); O=$(pwd)/gpurun_out/fl_$v; rm -rf $O; mkdir -p $O
  NNJ_LIB_PATH=$(pwd)/$f rocprofv3 --kernel-trace -d $O --output-format csv -- python3 $(pwd)/tools/prof_run.py 256 50 1024 2 > $O/log.txt 2>&1
  T=$(find $O -name "*kernel_trace.csv" | head -1)
  python3 - "$T" "$K" "$v" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n = len(d) // 2
print(sys.argv[3].ljust(8), sys.argv[2], "launches", len(d), "first of rollout 1 / 2 (us): %.1f %.1f" % (d[0], d[n]), " all of rollout 2:", " ".join("%.0f" % x for x in d[n:]))
PY
  rm -rf $O
done
