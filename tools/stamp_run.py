"""Diagnostic: where does a site iteration of k_inc_score_w spend its cycles?  Needs a library built with -DNNJ_STAMP
(tools/ko_build.sh stamp -- -DNNJ_STAMP) at NNJ_LIB_PATH.  Runs one 256 x 50 x 1024 rollout and prints the accumulated
s_memtime deltas per phase (cdna_hip_programming.md section 7)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402

cfgs = utils.shipped_config()
g = Nnj(cfgs, "cuda:0")
g.load_weights(weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp")))
g.set_concurrency(1)
codes = torch.from_numpy(synth.synth_codes(256, 50, 1024, seed=1, gap_frac=0.2)).cuda()
out = (C.c_ulonglong * 8)()
g.lib.nnj_debug_read_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
g.rollout_argmax(codes, None)
g.lib.nnj_debug_read_stamps(out)                 # (warm-up rollout: cleared)
g.rollout_argmax(codes, None)
g.lib.nnj_debug_read_stamps(out)
v = list(out)
n = max(v[0], 1)
print("site iterations (all waves, the w<2> and w<3> launches of one rollout):", v[0])
print(f"cycles per iteration: loads until landed {v[1] / n:.0f}; phase A incl. loads {v[2] / n:.0f}; "
      + "; ".join(f"phase B tile {t} {v[3 + t] / n:.0f}" for t in range(4)))
tot = v[2] + sum(v[3:7])
print(f"total per iteration {tot / n:.0f} cycles; shares: loads {v[1] / tot:.2f}, phase A proper {(v[2] - v[1]) / tot:.2f}, "
      f"phase B {sum(v[3:7]) / tot:.2f}")
