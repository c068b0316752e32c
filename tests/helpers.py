"""Shared helpers for the parity tests: golden fixture access, seeded inputs."""
from __future__ import annotations

import glob
import os

import numpy as np

from neuralnj_amd import synth, utils, weights

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(max_taxa=None, min_taxa=None):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        nm = os.path.basename(p)[:-4]
        z = np.load(p)
        T = z["codes"].shape[1]
        if max_taxa is not None and T > max_taxa:
            continue
        if min_taxa is not None and T < min_taxa:
            continue
        out.append(nm)
    return out


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfgs = utils.shipped_config()
    cfgs.model.num_enc_layers = int(z["layers"])
    st = weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"]))
    packed = weights.pack(cfgs, st)
    assert weights.digest(packed) == str(z["weights_sha256"]), "seeded weights drifted from the fixture"
    return z, cfgs, packed


def onehot_f32(codes):
    return synth.codes_to_onehot(codes).astype(np.float32)


def split_trace(flat, T):
    """[B, sum P(n)] -> list over steps of [B, P(n)], n = T..2."""
    out, off = [], 0
    for n in range(T, 1, -1):
        p = n * (n - 1) // 2
        out.append(flat[:, off:off + p])
        off += p
    assert off == flat.shape[1]
    return out


def assert_logits_close(got, ref, rel=1e-4, what="logits"):
    """Pair scores within `rel` of the table's scale (BASELINE.json: 1e-4 relative)."""
    scale = max(float(np.abs(ref).max()), 1.0)
    err = float(np.abs(got - ref).max())
    assert err <= rel * scale, f"{what}: max abs err {err:.3e} > {rel:g} * {scale:.3e}"
    return err / scale


def decisive_steps(top2_gap, logits_scale, rel=1e-4):
    """Steps whose top-2 gap is far above the allowed score error: the argmax there
    must agree exactly; elsewhere a flip is a legitimate fp32 near-tie."""
    return top2_gap > 4.0 * rel * logits_scale
