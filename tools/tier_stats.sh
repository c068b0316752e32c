#!/bin/bash
# per-kernel average launch times of the bench workload under an environment setting: tools/tier_stats.sh TAG "ENV=.."
# (rocprofv3 kernel trace of 2 single-stream rollouts; prints the scorer kernels' rows)
TAG=$1; shift
export TMPDIR=/tmp
O=$(pwd)/gpurun_out/tier_$TAG
mkdir -p $O
for e in "$@"; do export $e; done
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $(pwd)/bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-single-msa --no-verify --no-compat --no-profile > $O/bench.json 2> $O/err.txt
S=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp $S $O/kernel_stats.csv
rm -rf $O/stats
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])):
    if any(k in r['Name'] for k in ('k_inc_score','k_step_alpha','k_pair_')):
        print(f"$TAG {r['Name'][:64]:64s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
