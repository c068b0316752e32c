import sys, os, json, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import bench
from neuralnj_amd import synth, utils, weights
from neuralnj_amd._lib import Nnj
cfgs = utils.shipped_config()
packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
g = Nnj(cfgs, "cuda:0"); g.load_weights(packed)
out = {}
ranks = [int(a) for a in sys.argv[1:]] or [0, 3]      # the data seeds of bench.py's ranks (1000 + rank)
K = int(os.environ.get('E64_K', '24'))                # trees per rank (bench.py itself verifies 8)
for rank in ranks:
    codes = torch.from_numpy(synth.synth_codes(256, 50, 1024, seed=1000 + rank, gap_frac=0.2)).cuda()
    m = g.rollout_argmax(codes, None)["merges"].cpu()
    v = bench.verify_sample(g, cfgs, packed, codes, m, 50, 1024, k=K, threads=16)
    out[rank] = {k: v[k] for k in ("ok", "score_err_rel_vs_fp32_oracle", "score_err_rel_vs_fp64", "fp32_oracle_err_rel_vs_fp64",
                                   "elem_rel_err_vs_fp64_entries_over_1pct_of_scale", "elem_rel_err_vs_fp64_top5_of_each_table",
                                   "merge_lists_identical_to_fp64_oracle", "trees", "worst_table")}
    print(rank, out[rank], flush=True)
