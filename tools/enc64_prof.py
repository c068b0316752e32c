#!/usr/bin/env python3
"""The fp64 encoder of BASELINE configs[4] (one 200 x 4096 alignment) for rocprofv3:  rocprofv3 --kernel-trace --stats -- python3 tools/enc64_prof.py [T] [L] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfgs = utils.shipped_config()
g = Nnj(cfgs, "cuda:0")
g.load_weights(weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp")))
codes = torch.from_numpy(synth.synth_codes_tree(1, T, L, seed=4242)).cuda()
g.encode(codes, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    x = g.encode(codes, None)
torch.cuda.synchronize()
print(f"encode {T} x {L}: {1e3 * (time.perf_counter() - t0) / reps:.1f} ms", float(x[0, 0, 0, 0]))
