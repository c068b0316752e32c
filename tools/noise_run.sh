#!/bin/bash
# noise of every ab_build/lib*.so on the wide fixtures (tools/noise_ab.py), one line per (variant, fixture)
for f in ab_build/lib*.so; do
v=$(basename $f .so); v=${v#lib}
NNJ_LIB_PATH=$(pwd)/$f python tools/noise_ab.py $v "$@" 2>/dev/null
done
