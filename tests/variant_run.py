"""Child process of test_gpu_parity.py::test_switchable_kernels_agree: one traced Argmax rollout of a small seeded batch
under whatever NNJ_* switches the environment carries (the library reads them once per process), tables and merge
lists written to the .npz named on the command line."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402


def main(out_path):
    cfgs = utils.shipped_config()
    g = Nnj(cfgs, "cuda:0")
    g.load_weights(weights.pack(cfgs, weights.seeded_state(cfgs, 3, "sharp")))
    codes = synth.synth_codes(3, 50, 72, seed=21, gap_frac=0.2)
    mask = np.zeros((3, 72), dtype=np.uint8)
    mask[1, 60:] = 1                                       # one padded alignment (odd number of live sites in a chunk)
    out = g.rollout_argmax(torch.from_numpy(codes).cuda(), torch.from_numpy(mask).cuda(), want_trace=True)
    torch.cuda.synchronize()
    g.check_numeric()
    np.savez(out_path, merges=out["merges"].cpu().numpy(), logits=out["logits"].cpu().numpy())


if __name__ == "__main__":
    main(sys.argv[1])
