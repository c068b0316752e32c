"""B = 1 latency under the environment's switches: ms per tree of the hipGraph-replayed single-alignment rollout
(the bench's single_msa figure), 200 replays."""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402

cfgs = utils.shipped_config()
g = Nnj(cfgs, "cuda:0")
g.load_weights(weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp")))
codes = torch.from_numpy(synth.synth_codes(1, 50, 1024, seed=1, gap_frac=0.2)).cuda()
out = g.rollout_argmax(codes, None)
for _ in range(5):
    out = g.rollout_argmax(codes, None, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    out = g.rollout_argmax(codes, None, out=out)
torch.cuda.synchronize()
print(" ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("NNJ_")), f"ms_per_tree {(time.perf_counter() - t) / 200 * 1e3:.3f}",
      "merges_sum", int(out["merges"].sum()))
