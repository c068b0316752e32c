#!/bin/bash
# per-kind kernel ms of every ab_build/lib*.so through tools/ko_time.py (timing-only builds: no numeric check), ROUNDS times
R=${1:-2}
for i in $(seq $R); do for f in ab_build/lib*.so; do NNJ_LIB_PATH=$(pwd)/$f timeout -k 10 120 python tools/ko_time.py 256 2 2>/dev/null | tail -1; done; done
