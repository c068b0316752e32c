#!/bin/bash
# headline-margin scan (tools/e64_scan.py, 8 sampled trees per rank seed) of every ab_build/lib*.so for the given rank seeds
for f in ab_build/lib*.so; do
  echo "== $f"
  NNJ_LIB_PATH=$(pwd)/$f E64_K=8 python tools/e64_scan.py "$@" 2>/dev/null | grep -o "^[0-9] {.ok.: [A-Za-z]*, .score_err_rel_vs_fp32_oracle.: [0-9.e-]*, .score_err_rel_vs_fp64.: [0-9.e-]*"
done
