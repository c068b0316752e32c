#!/bin/bash
# same-box comparison of every ab_build/lib*.so, B = 256 (per-kind kernel ms, single stream) AND B = 1 (ms per tree of the
# graph-replayed single-alignment rollout), ROUNDS times in alternation: tools/ab_libs.sh [ROUNDS]
R=${1:-2}
for i in $(seq $R); do
for f in ab_build/lib*.so; do
v=$(basename $f .so); v=${v#lib}
NNJ_LIB_PATH=$(pwd)/$f python bench.py --streams 1 --steps ${STEPS:-2} --no-cpu-baseline --no-verify --no-compat --no-single-msa 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$v'.ljust(6), round(d['value'],1), ' '.join(f'{n[2:]}={k.get(n,0):.1f}' for n in ('k_pair_alpha','k_pair_score','k_pair_alpha_incr','k_pair_score_incr','k_tok1','k_ffn','k_qkv6','k_row_s','k_row_pv')))"
echo -n "$v B=1 "; NNJ_LIB_PATH=$(pwd)/$f python tools/b1_ab.py 2>/dev/null | tail -1
done; done
