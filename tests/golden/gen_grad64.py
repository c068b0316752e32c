"""float64 gradients of a golden gradient fixture, from the CPU gradient oracle (oracle/grad_oracle.py; needs no reference):
tests/golden/grad64_<name>.npy, flat in state_dict order, stored as float32.

Why: at the bench shape (1 x 50 x 1024, "sharp" weights) the score tables reach |500| and the chosen merges have
probabilities near 1, so the policy gradient adv * (onehot - p) turns a 2e-5 relative float32 error of a logit into a 0.5 %
error of the gradient.  There the reference's OWN float32 gradients are 4.3e-3 (of a tensor's scale) away from the float64
result; a float32 implementation can only be asked to be as close to the truth as the reference is, so
tests/test_gpu_finetune.py checks the large case against this file as well.

usage: python tests/golden/gen_grad64.py b1_t50_l1024_s0        (about two minutes on 8 cores)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import grad_oracle  # noqa: E402
from neuralnj_amd import synth, utils, weights  # noqa: E402


def main(name):
    z = np.load(os.path.join(HERE, f"grad_{name}.npz"), allow_pickle=True)
    cfgs = utils.shipped_config()
    cfgs.model.num_enc_layers = int(z["layers"])
    st = weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"]))
    sd = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in st.items()}
    loss, _ = grad_oracle.reinforce_loss(sd, synth.codes_to_onehot(z["codes"]), z["mask"], z["merges"], z["tree_scores"],
                                         float(z["baseline"]), float(z["temperature"]), float(z["strength"]),
                                         int(z["layers"]), torch.float64)
    loss.backward()
    g = np.concatenate([p.grad.numpy().reshape(-1) for p in sd.values()])
    np.save(os.path.join(HERE, f"grad64_{name}.npy"), g.astype(np.float32))
    ref = z["grads"]
    print(f"{name}: float64 loss {float(loss.detach()):.6f} (reference float32 {float(z['loss']):.6f}); "
          f"reference gradient / float64 gradient, least squares: {float(ref @ g / (g @ g)):.5f}")


if __name__ == "__main__":
    main(sys.argv[1])
