"""Diagnostic: the fp64 encoder of the > 64-row path (csrc/nnj_encoder64.hpp) against the fp64 oracle's encoder on a few
shapes (masked tails, odd site counts, patches, a narrow model), with the layer-0 taps.  Prints max |hip - f64| / max |f64|:
the embeddings are rounded to fp32 once, so ~6e-8 is the floor.
    python tools/enc64_check.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
from helpers import onehot_f32  # noqa: E402
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

r = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
worst = 0.0
for (B, T, L, K, dim, layers) in [(2, 70, 64, 1, 64, 6), (1, 100, 96, 1, 64, 6), (1, 130, 41, 1, 64, 2), (1, 256, 32, 1, 64, 1),
                                  (1, 80, 48, 2, 64, 2), (2, 72, 40, 4, 32, 3), (1, 100, 256, 1, 64, 6)]:
    cfgs = utils.shipped_config()
    cfgs.model.num_enc_layers = layers
    cfgs.model.patch_size = K
    cfgs.model.embed_dim = dim
    cfgs.model.num_enc_heads = dim // 8
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 41, "sharp"))
    g = Nnj(cfgs, "cuda:0")
    g.load_weights(packed)
    codes = synth.synth_codes_tree(B, T, L, 500 + T)
    mask = np.zeros((B, L), bool)
    tail = 2 * K
    mask[:, L - tail:] = True
    codes[:, :, L - tail:] = 5
    o64, o32 = Oracle(cfgs, packed, "f64"), Oracle(cfgs, packed)
    o64.set_threads(8)
    t0 = time.time()
    e64, t64 = o64.encode(onehot_f32(codes), mask, taps=True)
    e32 = o32.encode(onehot_f32(codes), mask)
    tc, tm = torch.from_numpy(codes), torch.from_numpy(mask)
    out = {}
    for stop, nm in ((1, "row0"), (2, "col0")):
        g.debug_encoder_stop(stop)
        out[nm] = r(g.encode(tc, tm).cpu().numpy(), t64[stop])
    g.debug_encoder_stop(0)
    h = g.encode(tc, tm).cpu().numpy()
    e = r(h, e64)
    worst = max(worst, e)
    # the float one-hot input form of nnj_encode
    hf = g.encode_onehot(torch.from_numpy(onehot_f32(codes)), tm).cpu().numpy() if hasattr(g, "encode_onehot") else None
    print(f"B={B} T={T} L={L} patch={K} dim={dim} layers={layers}: hip vs f64 {e:.2e} (row0 {out['row0']:.2e}, col0 {out['col0']:.2e})"
          f"  fp32 oracle vs f64 {r(e32, e64):.2e}" + (f"  onehot input {r(hf, e64):.2e}" if hf is not None else "")
          + f"  [{time.time() - t0:.0f} s]", flush=True)
    g.close()
print("worst", worst)
sys.exit(0 if worst < 5e-7 else 1)
