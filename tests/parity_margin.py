"""Report how far the HIP path is from the reference's golden vectors (not a test: the tests assert
1e-4, this prints the measured margin so a change of arithmetic can be judged).

    python tests/parity_margin.py [out.json]

For every fixture of tests/golden: max |score - golden| / table scale (teacher-forced along the reference's
merges) and max |encoder output - golden| / max |golden|; then the same score error measured against the
fp64 build of the oracle ("truth"), next to the golden table's own distance from that truth -- the fp32
noise of the reference, which bounds what any fp32 implementation can be asked to match.
(Lives under tests/ because it calls the oracle: test infrastructure only.)
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

from helpers import golden_names, load_golden, onehot_f32  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402


def main():
    rows = {}
    for name in golden_names():
        z, cfgs, packed = load_golden(name)
        g = Nnj(cfgs, "cuda:0")
        g.load_weights(packed)
        codes, mask = torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"])
        r = g.rollout_argmax(codes, mask, forced_merges=z["merges"], want_trace=True, want_state=True)
        logits = r["logits"].cpu().numpy()
        scale = max(float(np.abs(z["logits"]).max()), 1.0)
        e_log = float(np.abs(logits - z["logits"]).max()) / scale
        st = r["state"].cpu().numpy()
        if "enc" in z.files:
            e_enc = float(np.abs(st - z["enc"]).max() / np.abs(z["enc"]).max())
        else:
            e_enc = float(np.abs(st[:, ::7, ::61, :] - z["enc_slice"]).max() / np.abs(z["enc_slice"]).max())
        o64 = Oracle(cfgs, packed, "f64")
        t64 = o64.rollout_argmax(onehot_f32(z["codes"]), z["mask"], forced_merges=z["merges"])["logits"]
        e_hip64 = float(np.abs(logits - t64).max()) / scale
        e_ref64 = float(np.abs(z["logits"] - t64).max()) / scale
        rows[name] = {"scores_rel": e_log, "encoder_rel": e_enc, "table_scale": scale,
                      "scores_rel_vs_f64": e_hip64, "reference_rel_vs_f64": e_ref64}
        print(f"{name:36s} scores {e_log:.2e}  encoder {e_enc:.2e}  | vs f64: hip {e_hip64:.2e}  reference "
              f"{e_ref64:.2e}  (scale {scale:.3g})", flush=True)
        g.close()
    worst = {"scores_rel": max(v["scores_rel"] for v in rows.values()),
             "encoder_rel": max(v["encoder_rel"] for v in rows.values()),
             "scores_rel_vs_f64": max(v["scores_rel_vs_f64"] for v in rows.values()),
             "reference_rel_vs_f64": max(v["reference_rel_vs_f64"] for v in rows.values())}
    print("worst:", worst)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump({"tolerance": 1e-4, "worst": worst, "fixtures": rows}, f, indent=1)


if __name__ == "__main__":
    main()
