"""Randomised parity sweep (a tool, not a test): many seeded (weights, shape, alignment) cases, HIP rollout against
the fp32 oracle teacher-forced along the HIP merges.  Prints the distribution of the score error relative to the
table scale (the tests assert 1e-4) and whether any decisive step disagrees.

    python tests/parity_sweep.py [cases] [out.json]

Lives under tests/ because it calls the oracle (test infrastructure only).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(2024)
    cfgs = utils.shipped_config()
    rows = []
    for k in range(cases):
        wseed = int(rng.integers(100, 10_000))
        style = "sharp" if k % 4 else "plain"
        packed = weights.pack(cfgs, weights.seeded_state(cfgs, wseed, style))
        T = int(rng.integers(3, 65))
        L = int(rng.integers(4, 160)) * 4
        B = int(rng.integers(1, 4))
        seed = int(rng.integers(0, 1 << 30))
        codes = synth.synth_codes_tree(B, T, L, seed)
        mask = np.zeros((B, L), bool)
        if k % 3 == 0 and L > 16:
            codes[:, :, L - 8:] = 5
            mask[:, L - 8:] = True
        g = Nnj(cfgs, "cuda:0")
        g.load_weights(packed)
        r = g.rollout_argmax(torch.from_numpy(codes), torch.from_numpy(mask), want_trace=True)
        merges = r["merges"].cpu().numpy()
        logits = r["logits"].cpu().numpy()
        g.check_numeric()
        g.close()
        o = Oracle(cfgs, packed)
        ref = o.rollout_argmax(synth.codes_to_onehot(codes).astype(np.float32), mask, forced_merges=merges)
        scale = max(float(np.abs(ref["logits"]).max()), 1.0)
        err = float(np.abs(logits - ref["logits"]).max()) / scale
        decisive = ref["top2_gap"] > 4e-4 * scale
        flips = int((ref["merges"][decisive] != merges[decisive]).any(axis=-1).sum())
        rows.append(dict(case=k, style=style, wseed=wseed, B=B, T=T, L=L, scale=scale, err=err, decisive_flips=flips))
        print(f"{k:3d} {style:5s} B={B} T={T:2d} L={L:3d} scale {scale:9.3g} err {err:.2e} flips {flips}", flush=True)
    errs = np.array([r["err"] for r in rows])
    summary = dict(cases=cases, tolerance=1e-4, max=float(errs.max()), p90=float(np.quantile(errs, 0.9)),
                   median=float(np.median(errs)), over_tolerance=int((errs > 1e-4).sum()),
                   decisive_flips=int(sum(r["decisive_flips"] for r in rows)))
    print("summary:", summary)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(dict(summary=summary, cases=rows), f, indent=1)


if __name__ == "__main__":
    main()
