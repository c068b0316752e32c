#!/usr/bin/env python3
"""One Finetune episode with gradients at the bench shape: sampled merges from nnj_rollout_sample,
reinforce_loss (differentiable forward), backward, Adam step.  Prints seconds per episode and peak device memory.
usage: python3 tools/finetune_step.py [T] [L] [episodes]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd.environment import PhyInferEnv  # noqa: E402
from neuralnj_amd.model import PhyloATTN  # noqa: E402
from neuralnj_amd.rollout import reinforce_loss  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
E = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
cfgs = utils.shipped_config()
agent = PhyloATTN(cfgs)
sd = weights.seeded_state(cfgs, 0, "plain")
agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
agent = agent.to(dev).eval()
opt = torch.optim.Adam(agent.parameters(), lr=1e-5)
codes = synth.synth_codes_tree(1, T, L, seed=3)
batch = {"data": torch.from_numpy(synth.codes_to_onehot(codes)), "seqs": [synth.codes_to_seqs(codes[0])],
         "seq_keys": [[f"taxon{i + 1}" for i in range(T)]], "seq_weights": torch.ones((1, L), dtype=torch.float32)}
times, losses = [], []
for ep in range(E + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        u = torch.from_numpy(np.random.default_rng(ep).random((1, T - 1)).astype(np.float32))
        r = agent._context().rollout_sample(torch.from_numpy(codes), None, u, temperature=1.0, replicas=1)
        merges = r["merges"].cpu().numpy()
    opt.zero_grad()
    loss, _ = reinforce_loss(batch, agent, PhyInferEnv(cfgs, dev), merges, np.array([1.0], np.float32), 0.5)
    loss.backward()
    torch.nn.utils.clip_grad_value_(agent.parameters(), clip_value=1.0)
    opt.step()
    torch.cuda.synchronize()
    if ep:
        times.append(time.perf_counter() - t0)
    losses.append(float(loss.detach()))
print(json.dumps({"workload": f"Finetune episode, B=1, {T}x{L}: sample + differentiable rollout + backward + Adam",
                  "s_per_episode": float(np.median(times)), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2**30,
                  "losses": losses}))
