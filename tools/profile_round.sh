#!/bin/bash
# Round profile pack (run on the GPU box from the repo root): rocprofv3 kernel stats of the bench workload, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes), SQ issue / wait counters.  Output under gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r02}
R=$(pwd)
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --streams 1 --steps 3 --no-cpu-baseline --no-single-msa --no-verify --no-compat > $O/bench_b256_under_rocprof.json 2> $O/stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/tools/prof_run.py 256 50 1024 1 > $O/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/tools/prof_run.py 256 50 1024 1 > $O/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES -d $O/sq --output-format csv -- python3 $R/tools/prof_run.py 256 50 1024 1 > $O/sq.log 2>&1
echo "sq done"
F=$(find $O/fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/write -name "*counter_collection.csv" | head -1)
Q=$(find $O/sq -name "*counter_collection.csv" | head -1)
S=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 $R/tools/traffic_summary.py $F $W $O/traffic_b256.json $O/traffic.json > $O/traffic_summary.txt
python3 $R/tools/pmc_summary.py $Q > $O/sq_counters_b256.txt
cp $S $O/bench_b256_kernel_stats.csv
# the raw per-dispatch csvs are large: keep the summaries only
rm -rf $O/fetch $O/write $O/sq $O/stats
ls -la $O
