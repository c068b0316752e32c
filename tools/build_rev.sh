#!/bin/bash
# builds the library of a git revision into ab_build/lib<name>.so for same-box A/B runs: tools/build_rev.sh NAME REV
# (revisions with nnj_step0_tu.hip: that translation unit with its own flags, as neuralnj_amd/build.py does)
set -e
name=$1; rev=$2
R=$(cd "$(dirname "$0")/.." && pwd)
d=$(mktemp -d /tmp/nnj_rev.XXXXXX)
git -C $R archive $rev neuralnj_amd/csrc include | tar -x -C $d
mkdir -p $R/ab_build
cd $d/neuralnj_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
if [ -f nnj_step0_tu.hip ]; then
  /opt/rocm/bin/hipcc $F -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp -c -o tu2.o nnj_step0_tu.hip 2>/dev/null &
  /opt/rocm/bin/hipcc $F -c -o api.o nnj_api.hip 2>/dev/null
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab_build/lib$name.so api.o tu2.o
else
  /opt/rocm/bin/hipcc $F -shared -o $R/ab_build/lib$name.so nnj_api.hip 2>/dev/null
fi
rm -rf $d
echo built lib$name.so from $rev
