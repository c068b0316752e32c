#!/bin/bash
# same-box A/B of two builds of the library (ab_build/libA.so, libB.so): alternating bench runs, per-kind kernel ms
for i in 1 2; do for v in A B; do
NNJ_LIB_PATH=$(pwd)/ab_build/lib$v.so python bench.py --streams 1 --steps 3 --no-cpu-baseline --no-verify --no-compat --no-single-msa 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$v', round(d['value'],1), ' '.join(f'{n[2:]}={k.get(n,0):.1f}' for n in ('k_pair_alpha','k_pair_score','k_pair_alpha_incr','k_pair_score_incr','k_tok1','k_ffn','k_qkv6','k_row_s','k_row_pv','k_agg_alpha','k_agg_finish')))"
done; done
