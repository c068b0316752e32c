#!/bin/bash
# builds the library of a git revision into ab_build/lib<name>.so for same-box A/B runs: tools/build_rev.sh NAME REV
set -e
name=$1; rev=$2
R=$(cd "$(dirname "$0")/.." && pwd)
d=$(mktemp -d /tmp/nnj_rev.XXXXXX)
git -C $R archive $rev neuralnj_amd/csrc include | tar -x -C $d
mkdir -p $R/ab_build
cd $d/neuralnj_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -o $R/ab_build/lib$name.so nnj_api.hip 2>/dev/null
rm -rf $d
echo built lib$name.so from $rev
