"""One complete Search round at BASELINE configs[4] (200 taxa x 4096 sites, 8 sampled rollouts, GTR+I+G estimated,
three branch-length sweeps) -- the workload tools/search_profile.sh traces with rocprofv3.  Prints the round's wall time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd.environment import PhyInferEnv  # noqa: E402
from neuralnj_amd.model import PhyloATTN  # noqa: E402
from neuralnj_amd.rollout import search_rollouts  # noqa: E402

T, L, R = 200, 4096, 8
cfgs = utils.shipped_config()
agent = PhyloATTN(cfgs)
agent.load_state_dict({k: torch.from_numpy(v) for k, v in weights.seeded_state(cfgs, 0, "sharp").items()}, strict=True)
agent = agent.to("cuda:0")
codes = synth.synth_codes_tree(1, T, L, seed=4242)
batch = {"codes": torch.from_numpy(codes), "seqs": [[""] * T], "seq_keys": [[f"taxon{i + 1}" for i in range(T)]],
         "seq_weights": torch.ones((1, L), dtype=torch.float32)}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    det = {}
    _, ll, trees = search_rollouts(batch, agent, PhyInferEnv(cfgs, "cuda:0"), R, seed=5, temperature=1.0, model="auto",
                                   sweeps=3, details=det)
    torch.cuda.synchronize()
    print(f"round {rep}: {time.perf_counter() - t0:.3f} s, {len(trees)} distinct trees, best log-likelihood {ll:.3f}", flush=True)
