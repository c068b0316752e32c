#!/bin/bash
# per-kernel times of the fp64 encoder at 200 x 4096 (rocprofv3 kernel trace): tools/enc64_prof.sh TAG
TAG=${1:-enc64}
export TMPDIR=/tmp
O=$(pwd)/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $(pwd)/tools/enc64_prof.py 200 4096 3 > $O/run.txt 2> $O/err.txt
S=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp $S $O/kernel_stats.csv
rm -rf $O/stats
cat $O/run.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
