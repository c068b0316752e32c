#!/bin/bash
# rocprofv3 kernel stats of 5 single-alignment rollouts (50 x 1024, plain launches: the graph is off under the profiler's
# per-call buffers) -> gpurun_out/b1_kernel_stats.csv + a per-kernel table
export TMPDIR=/tmp
O=$(pwd)/gpurun_out/b1prof
mkdir -p $O
NNJ_GRAPH=0 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $(pwd)/tools/prof_run.py 1 50 1024 5 > $O/run.txt 2> $O/err.txt
S=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp $S $(pwd)/gpurun_out/b1_kernel_stats.csv
rm -rf $O/stats
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/b1_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"B = 1, 50 x 1024: {tot/5e6:.3f} ms of kernels per rollout")
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    print(f"{r['Name'][:58]:58s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:7.1f} ms/rollout {float(r['TotalDurationNs'])/5e6:6.3f}")
PY
