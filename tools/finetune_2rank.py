#!/usr/bin/env python3
"""Rehearsal of the data-parallel Finetune step: the episodes of an epoch split over the ranks, ONE
all-reduce of the flat gradient bucket per optimizer step, every rank takes the same step.  On a one-GPU box both ranks
share cuda:0 and the collective runs over gloo (NNJ_BACKEND=gloo, the default here); on a multi-GPU node use
NNJ_BACKEND=nccl (RCCL), one GPU per rank.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \\
      tools/finetune_2rank.py [T] [L]
Rank 0 prints one JSON line: the weights of all ranks are identical after the epochs (checksums gathered), and differ
from the initial ones."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd.environment import PhyInferEnv  # noqa: E402
from neuralnj_amd.model import PhyloATTN  # noqa: E402
from neuralnj_amd.rollout import rl_finetuning  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
backend = os.environ.get("NNJ_BACKEND", "gloo")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=dev)
else:
    dist.init_process_group(backend)
cfgs = utils.shipped_config()
cfgs.num_episodes, cfgs.num_epoch = 4, 3
agent = PhyloATTN(cfgs)
agent.load_state_dict({k: torch.from_numpy(v) for k, v in weights.seeded_state(cfgs, 0, "plain").items()}, strict=True)
agent = agent.to(dev)
w0 = torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).double().cpu()
opt = torch.optim.Adam(agent.parameters(), lr=1e-4)
codes = synth.synth_codes_tree(1, T, L, seed=5)
batch = {"data": torch.from_numpy(synth.codes_to_onehot(codes)), "seqs": [synth.codes_to_seqs(codes[0])],
         "seq_keys": [[f"taxon{i + 1}" for i in range(T)]], "seq_weights": torch.ones((1, L), dtype=torch.float32)}
t0 = time.perf_counter()
out = rl_finetuning(cfgs, batch, agent, opt, PhyInferEnv(cfgs, dev), stop_step=10 ** 6, seed=1, device=dev, dist=dist)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
w1 = torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).double().cpu()
mine = torch.tensor([float(w1.sum()), float((w1 * w1).sum()), float(w1.abs().max()), float((w1 - w0).abs().max())],
                    dtype=torch.float64)
sums = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(sums, mine)
if rank == 0:
    same = all(torch.equal(sums[0][:3], s[:3]) for s in sums)
    print(json.dumps({"workload": f"data-parallel Finetune rehearsal: {world} ranks, backend {backend}, {T} x {L}, "
                                  f"{cfgs.num_epoch} epochs of {cfgs.num_episodes} episodes split over the ranks",
                      "weights_identical_on_all_ranks": bool(same), "largest_weight_change": float(sums[0][3]),
                      "losses_rank0": out["losses"], "episodes_counted": out["step_cur"], "seconds": dt}))
    if not same or not float(sums[0][3]) > 0:
        sys.exit(3)
dist.barrier()
dist.destroy_process_group()
