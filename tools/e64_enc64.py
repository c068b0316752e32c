"""Diagnostic (headline margin): the 8 verified trees of rank seeds 1004 / 1000 with the fp64 encoder forced at
50 rows (NNJ_ENC64=2) against the shipped f16x3 encoder: how much of the distance from fp64 is the encoder, how much the fp32
recursion of the NJ loop.   python tools/e64_enc64.py [rank ...]"""
import os
import subprocess
import sys

code = r'''
import sys, os, json, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import bench
from neuralnj_amd import synth, utils, weights
from neuralnj_amd._lib import Nnj
cfgs = utils.shipped_config()
packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
g = Nnj(cfgs, "cuda:0"); g.load_weights(packed)
rank = int(sys.argv[1])
codes = torch.from_numpy(synth.synth_codes(256, 50, 1024, seed=1000 + rank, gap_frac=0.2)).cuda()
idx = sorted({int(i) for i in np.linspace(0, 255, 8)})
sub = codes[idx].contiguous()
m = g.rollout_argmax(sub, None)["merges"].cpu()
v = bench.verify_sample(g, cfgs, packed, sub, m, 50, 1024, k=8, threads=16)
w = v["worst_table"]
print(os.environ.get("NNJ_ENC64", "1"), rank, "hip_vs_fp64 %.2e  fp32_oracle_vs_fp64 %.2e  worst step %d (rows %d)  step0 %.2e  after step 8 %.2e" % (
    v["score_err_rel_vs_fp64"], v["fp32_oracle_err_rel_vs_fp64"], w["step"], w["rows_alive"], w["err_rel_by_step_first8"][0], w["err_rel_max_after_step8"]), flush=True)
'''
for rank in [int(a) for a in sys.argv[1:]] or [4, 0]:
    for mode in ("1", "2"):
        env = dict(os.environ, NNJ_ENC64=mode)
        subprocess.run([sys.executable, "-c", code, str(rank)], env=env, check=False)
