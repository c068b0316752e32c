"""Where does the fp32-level noise of the wide (> 64 rows) path come from?  Diagnostic, not a test.
    python tests/wide_margin.py
For a few wide cases: encoder output and step-0 table of HIP, of the fp32 oracle and (goldens) of the reference against
the fp64 oracle; and the HIP scorer fed with the fp64 encoder output (separates scorer noise from encoder noise)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, onehot_f32, split_trace  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402


def case(name, cfgs, packed, codes, mask, gold=None):
    g = Nnj(cfgs, "cuda:0")
    g.load_weights(packed)
    B, T, L = codes.shape
    oh = onehot_f32(codes)
    o32, o64 = Oracle(cfgs, packed), Oracle(cfgs, packed, "f64")
    e64 = o64.encode(oh, mask)
    e32 = o32.encode(oh, mask)
    eh = g.encode(torch.from_numpy(codes), torch.from_numpy(mask)).cpu().numpy()
    es = np.abs(e64).max()
    t64 = o64.pair_scores_full(e64, mask)
    t32 = o32.pair_scores_full(e32, mask)
    th = g.pair_scores_full(torch.from_numpy(eh), torch.from_numpy(mask)).cpu().numpy()
    th_on64 = g.pair_scores_full(torch.from_numpy(e64), torch.from_numpy(mask)).cpu().numpy()
    t32_on64 = o32.pair_scores_full(e64, mask)
    sc = np.abs(t64).max()
    r = lambda a, b, s: float(np.abs(a - b).max() / s)  # noqa: E731
    print(f"{name}: enc  hip {r(eh, e64, es):.2e}  o32 {r(e32, e64, es):.2e}   step0  hip {r(th, t64, sc):.2e}  o32 {r(t32, t64, sc):.2e}"
          f"   scorer-only (fp64 enc in)  hip {r(th_on64, t64, sc):.2e}  o32 {r(t32_on64, t64, sc):.2e}"
          + (f"   ref-vs-f64 {r(gold, t64, sc):.2e}" if gold is not None else ""), flush=True)
    g.close()


def main():
    for nm in ("synth_b1_t100_l256_s11", "data_G_l_256_n_100_0_0p01_101", "synth_b1_t50_l1024_s1"):
        z, cfgs, packed = load_golden(nm)
        T = z["codes"].shape[1]
        case(nm, cfgs, packed, z["codes"], z["mask"], split_trace(z["logits"], T)[0])
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 21, "sharp"))
    codes = synth.synth_codes_tree(1, 256, 20, 18)
    case("seeded 1x256x20", cfgs, packed, codes, np.zeros((1, 20), bool))
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
    one = synth.synth_codes_tree(1, 200, 4096, seed=4242)[:, :, :160]
    case("200x160 window", cfgs, packed, one, np.zeros((1, 160), bool))
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "plain"))
    case("200x160 window, plain weights", cfgs, packed, one, np.zeros((1, 160), bool))


if __name__ == "__main__":
    main()
