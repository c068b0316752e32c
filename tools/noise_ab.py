"""Diagnostic: how far are the HIP score tables and the HIP encoder output of a fixture from the fp64
oracle's, for the library at NNJ_LIB_PATH (variant builds: tools/ko_build.sh NAME -- -DNNJ_...)?  The fp64 results are
cached under gpurun_out/noise_cache/ so that several variants measured in one call pay for them once.
    python tools/noise_ab.py TAG fixture [fixture ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
from helpers import load_golden, onehot_f32, split_trace  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

tag, names = sys.argv[1], sys.argv[2:]
os.makedirs("gpurun_out/noise_cache", exist_ok=True)
rows = []
for name in names:
    z, cfgs, packed = load_golden(name)
    B, T, L = z["codes"].shape
    cache = f"gpurun_out/noise_cache/{name}.npz"
    if not os.path.exists(cache):
        o64 = Oracle(cfgs, packed, "f64")
        o64.set_threads(16)
        r64 = o64.rollout_argmax(onehot_f32(z["codes"]), z["mask"], forced_merges=z["merges"], want_state=True)
        np.savez(cache, logits=r64["logits"], state=r64["state"])
    c = np.load(cache)
    t64, e64 = c["logits"], c["state"]
    g = Nnj(cfgs, "cuda:0")
    g.load_weights(packed)
    r = g.rollout_argmax(torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"]), forced_merges=z["merges"],
                         want_trace=True, want_state=True)
    hip, enc = r["logits"].cpu().numpy(), r["state"].cpu().numpy()
    g.close()
    scale = float(np.abs(t64).max())
    per_step = [float(np.abs(a - b).max()) / scale for a, b in zip(split_trace(hip, T), split_trace(t64, T))]
    row = dict(tag=tag, fixture=name, hip_vs_fp64=max(per_step), step0=per_step[0],
               reference_vs_fp64=float(np.abs(z["logits"] - t64).max()) / scale,
               encoder_vs_fp64=float(np.abs(enc - e64).max() / np.abs(e64).max()))
    rows.append(row)
    print(json.dumps(row), flush=True)
