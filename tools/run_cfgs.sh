for w in 512 1024 2048 4096; do echo "wgs $w"; NNJ_INCR_WGS=$w python bench.py --steps 3 --streams 1 --no-cpu-baseline --no-compat --no-verify --no-single-msa 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value'],1), k['k_pair_alpha_incr'], k['k_pair_score_incr'])"; done
for s in 2 3 4; do echo "streams $s"; python bench.py --steps 4 --streams $s --no-cpu-baseline --no-compat --no-verify --no-single-msa --no-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; done
