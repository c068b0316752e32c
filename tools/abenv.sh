#!/bin/bash
# same-box comparison of ONE library under several environment settings: tools/abenv.sh "NNJ_X=0" "NNJ_X=1" ...
# (one bench run each: single stream, kernel events; per-kind kernel ms per rollout)
for e in "$@"; do
env $e python bench.py --streams 1 --steps ${STEPS:-2} --no-cpu-baseline --no-verify --no-compat --no-single-msa 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$e'.ljust(24), round(d['value'],1), ' '.join(f'{n[2:]}={k.get(n,0):.1f}' for n in ('k_pair_alpha','k_pair_score','k_pair_alpha_incr','k_pair_score_incr','k_step_small','k_alpha_softmax','k_assemble_argmax','k_tok1','k_ffn','k_qkv6','k_row_s','k_row_pv')))"
done
