#!/usr/bin/env python3
"""cProfile of the host side of Finetune episodes at the bench shape (the B = 1 episode is bound by the host: ~3000
kernel launches and ~1500 autograd nodes).  usage: python3 tools/finetune_host_profile.py [episodes] > out.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd.environment import PhyInferEnv  # noqa: E402
from neuralnj_amd.model import PhyloATTN  # noqa: E402
from neuralnj_amd.rollout import reinforce_loss  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 3
T, L = 50, 1024
dev = torch.device("cuda:0")
cfgs = utils.shipped_config()
agent = PhyloATTN(cfgs)
agent.load_state_dict({k: torch.from_numpy(v) for k, v in weights.seeded_state(cfgs, 0, "plain").items()}, strict=True)
agent = agent.to(dev).eval()
opt = torch.optim.Adam(agent.parameters(), lr=1e-5)
codes = synth.synth_codes_tree(1, T, L, seed=3)
batch = {"data": torch.from_numpy(synth.codes_to_onehot(codes)), "seqs": [synth.codes_to_seqs(codes[0])],
         "seq_keys": [[f"taxon{i + 1}" for i in range(T)]], "seq_weights": torch.ones((1, L), dtype=torch.float32)}
rng = np.random.default_rng(0)
merges = np.zeros((1, T - 1, 2), np.int32)
for s, n in enumerate(range(T, 1, -1)):
    merges[0, s] = sorted(rng.choice(n, size=2, replace=False))


def episode():
    opt.zero_grad()
    t0 = time.perf_counter()
    loss, _ = reinforce_loss(batch, agent, PhyInferEnv(cfgs, dev), merges, np.array([1.0], np.float32), 0.5)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


episode()
ts = np.array([episode() for _ in range(E)])
print("host seconds per episode (forward issue, backward issue, optimizer + drain):", ts.mean(0).round(4).tolist())
pr = cProfile.Profile()
pr.enable()
for _ in range(E):
    episode()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue())
