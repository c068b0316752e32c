#!/bin/bash
# Timing-only variants of the library for same-box A/B runs (tools/abn.sh): copies csrc/ to ab_build/src_<name>/,
# applies the sed expressions given as "file::expr" arguments, builds ab_build/lib<name>.so.  Extra hipcc flags
# after "--".  Knock-out variants compute WRONG results on purpose; they never leave ab_build/ (git-ignored).
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
d=$R/ab_build/src_$name/neuralnj_amd/csrc
rm -rf $R/ab_build/src_$name; mkdir -p $d $R/ab_build/src_$name/include
cp $R/neuralnj_amd/csrc/* $d/; cp $R/include/nnj.h $R/ab_build/src_$name/include/
flags=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; flags="$*"; break; fi
  f=${1%%::*}; e=${1#*::}; sed -i "$e" $d/$f; shift
done
cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $flags -o $R/ab_build/lib$name.so nnj_api.hip 2>/dev/null
rm -rf $R/ab_build/src_$name
echo built lib$name.so
