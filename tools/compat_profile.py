#!/usr/bin/env python3
"""cProfile of the API-compatible per-step path (rollout.reinforce_rollout_argmax) at the bench shape: where the host
time of the reference's call sequence goes.  usage: python3 tools/compat_profile.py [B]"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd.environment import PhyInferEnv  # noqa: E402
from neuralnj_amd.model import PhyloATTN  # noqa: E402
from neuralnj_amd.rollout import reinforce_rollout_argmax  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, L = 50, 1024
cfgs = utils.shipped_config()
agent = PhyloATTN(cfgs)
sd = weights.seeded_state(cfgs, 0, "sharp")
agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
agent = agent.to("cuda:0")
c = synth.synth_codes(B, T, L, seed=1, gap_frac=0.2)
batch = {"data": torch.from_numpy(synth.codes_to_onehot(c)), "seqs": [[""] * T for _ in range(B)],
         "seq_keys": [[f"taxon{i + 1}" for i in range(T)] for _ in range(B)],
         "seq_weights": torch.ones((B, L), dtype=torch.float32)}
reinforce_rollout_argmax(batch, agent, PhyInferEnv(cfgs, "cuda:0"))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
reinforce_rollout_argmax(batch, agent, PhyInferEnv(cfgs, "cuda:0"))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
