"""Gradients of one REINFORCE loss on a golden gradient fixture, computed by the HIP Finetune path and written to
gpurun_out/ours_<name>.npy (flat, state_dict order; the score tables to ours_tables_<name>.npy) -- for comparing, off the box, against the reference's float32
gradients and the float64 oracle's.  usage: python tools/finetune_grad_dump.py b1_t50_l1024_s0"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd.environment import PhyInferEnv  # noqa: E402
from neuralnj_amd.model import PhyloATTN  # noqa: E402
from neuralnj_amd.rollout import reinforce_loss  # noqa: E402

name = sys.argv[1]
z = np.load(os.path.join(ROOT, "tests", "golden", f"grad_{name}.npz"), allow_pickle=True)
cfgs = utils.shipped_config()
cfgs.model.num_enc_layers = int(z["layers"])
agent = PhyloATTN(cfgs)
sd = weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"]))
agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
dev = torch.device("cuda:0")
agent = agent.to(dev).eval()
codes, mask = z["codes"], z["mask"]
B, T, L = codes.shape
batch = {"data": torch.from_numpy(synth.codes_to_onehot(codes)), "seqs": [synth.codes_to_seqs(codes[b]) for b in range(B)],
         "seq_keys": [[f"taxon{i + 1}" for i in range(T)] for _ in range(B)],
         "seq_weights": torch.from_numpy((~mask).astype(np.float32))}
env = PhyInferEnv(cfgs, dev)
loss, tables = reinforce_loss(batch, agent, env, z["merges"], z["tree_scores"], float(z["baseline"]), float(z["temperature"]),
                         float(z["strength"]))
agent.zero_grad()
loss.backward()
g = np.concatenate([p.grad.detach().cpu().numpy().reshape(-1) for p in agent.state_dict(keep_vars=True).values()])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", f"ours_{name}.npy"), g)
np.save(os.path.join(ROOT, "gpurun_out", f"ours_tables_{name}.npy"),
        torch.cat([t.detach().reshape(B, -1) for t in tables], dim=1).cpu().numpy())
print(f"{name}: loss {float(loss.detach()):.6f} (reference {float(z['loss']):.6f}), {g.size} gradient values written")
