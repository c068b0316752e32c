#!/usr/bin/env python3
"""rl_finetuning as the CLI runs it (greedy baseline, then per epoch: E sampled replicas, likelihood rewards, one batched
differentiable replay, backward, clipped Adam step) at the bench shape.  usage: python3 tools/finetune_epochs.py [T] [L] [E] [epochs]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd.environment import PhyInferEnv  # noqa: E402
from neuralnj_amd.model import PhyloATTN  # noqa: E402
from neuralnj_amd.rollout import rl_finetuning  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
E = int(sys.argv[3]) if len(sys.argv) > 3 else 8
EP = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
cfgs = utils.shipped_config()
cfgs.num_episodes, cfgs.num_epoch = E, EP
agent = PhyloATTN(cfgs)
agent.load_state_dict({k: torch.from_numpy(v) for k, v in weights.seeded_state(cfgs, 0, "plain").items()}, strict=True)
agent = agent.to(dev)
opt = torch.optim.Adam(agent.parameters(), lr=1e-5)
codes = synth.synth_codes_tree(1, T, L, seed=3)
batch = {"data": torch.from_numpy(synth.codes_to_onehot(codes)), "seqs": [synth.codes_to_seqs(codes[0])],
         "seq_keys": [[f"taxon{i + 1}" for i in range(T)]], "seq_weights": torch.ones((1, L), dtype=torch.float32)}
times = []
for rep in range(2):                                  # first repetition warms the allocator and the graphs up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = rl_finetuning(cfgs, batch, agent, opt, PhyInferEnv(cfgs, dev), stop_step=10 ** 6, seed=rep, device=dev)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
print(json.dumps({"workload": f"rl_finetuning, {T} x {L}: {EP} epochs of {E} episodes as one batch of replicas (greedy "
                              "baseline, sampling, likelihood rewards, gradients, Adam)",
                  "seconds": times[-1], "s_per_episode": times[-1] / (E * EP), "first_repetition_seconds": times[0],
                  "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "losses": out["losses"],
                  "best_score": out["the_best_score"]}))
