#!/usr/bin/env python3
"""One Argmax rollout for profiling under rocprofv3 (no timing, no oracle).
usage: python3 tools/prof_run.py [B] [T] [L] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfgs = utils.shipped_config()
g = Nnj(cfgs, "cuda:0")
g.load_weights(weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp")))
g.set_concurrency(1)      # whole-batch launches on one stream: per-launch byte counts belong to one launch of B alignments
codes = torch.from_numpy(synth.synth_codes(B, T, L, seed=1, gap_frac=0.2)).cuda()
for _ in range(reps):
    r = g.rollout_argmax(codes, None)
torch.cuda.synchronize()
print("done", r["merges"][0, :3].tolist())
