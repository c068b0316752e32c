#!/usr/bin/env python3
"""Which contractions a Finetune episode spends its GEMM time in: wraps train_ops.gemm with event timing (one
synchronisation per call: diagnostic only) and prints the shapes by total time.  usage: python3 tools/finetune_gemm_shapes.py [T] [L]"""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, train_ops, utils, weights  # noqa: E402
from neuralnj_amd.environment import PhyInferEnv  # noqa: E402
from neuralnj_amd.model import PhyloATTN  # noqa: E402
from neuralnj_amd.rollout import reinforce_loss  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
cfgs = utils.shipped_config()
agent = PhyloATTN(cfgs)
agent.load_state_dict({k: torch.from_numpy(v) for k, v in weights.seeded_state(cfgs, 0, "plain").items()}, strict=True)
agent = agent.to(dev).eval()
codes = synth.synth_codes_tree(1, T, L, seed=3)
batch = {"data": torch.from_numpy(synth.codes_to_onehot(codes)), "seqs": [synth.codes_to_seqs(codes[0])],
         "seq_keys": [[f"taxon{i + 1}" for i in range(T)]], "seq_weights": torch.ones((1, L), dtype=torch.float32)}
rng = np.random.default_rng(0)
merges = np.array([[sorted(rng.choice(n, size=2, replace=False)) for n in range(T, 1, -1)]], dtype=np.int32)
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
orig = train_ops.gemm


def timed(A, B, Cout, M, N, K, sA, sB, sC, nb=(1, 1), **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(A, B, Cout, M, N, K, sA, sB, sC, nb=nb, **kw)
    e1.record()
    e1.synchronize()
    key = (M, N, K, nb[0], "Ak" if sA[1] == 1 else "Am", "Bk" if sB[0] == 1 else "Bn")
    a = acc[key]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += 2.0 * M * N * K * nb[0]


for it in range(2):
    if it == 1:
        train_ops.gemm = timed
    loss, _ = reinforce_loss(batch, agent, PhyInferEnv(cfgs, dev), merges, np.array([1.0], np.float32), 0.5)
    loss.backward()
    torch.cuda.synchronize()
tot = sum(a[1] for a in acc.values())
print(f"GEMM time {tot:.1f} ms over {sum(a[0] for a in acc.values())} calls")
for key, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{str(key):48s} calls {a[0]:5d}  ms {a[1]:8.2f}  TFLOP/s {a[2] / a[1] / 1e9:7.2f}")
