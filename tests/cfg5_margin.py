"""Where does the fp32-level noise of BASELINE configs[4] (200 x 4096) come from?  Diagnostic, not a test:
encoder output of HIP and of the fp32 oracle against the fp64 oracle after 1, 2, 3, 6 layers (and after the row /
column block of layer 0), then the step-0 table of the HIP scorer fed with its own and with the fp64 encoder output
against the fp64 golden table (tests/golden/cfg5_f64_t200_l4096.npz).  Minutes of CPU for the fp64 encodes.
    python tests/cfg5_margin.py [L]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import onehot_f32  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402


def main():
    T, L = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    codes = synth.synth_codes_tree(1, T, 4096, seed=4242)[:, :, :L]
    mask = np.zeros((1, L), bool)
    oh = onehot_f32(codes)
    tc, tm = torch.from_numpy(codes), torch.from_numpy(mask)
    r = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
    full = utils.shipped_config()
    st_full = weights.seeded_state(full, 0, "sharp")
    for nl in (1, 2, 3, 6):
        cfgs = utils.shipped_config()
        cfgs.model.num_enc_layers = nl
        st = {k: v for k, v in st_full.items() if not k.startswith("seq_emb_layers.") or int(k.split(".")[1]) < nl}
        packed = weights.pack(cfgs, st)
        g = Nnj(cfgs, "cuda:0")
        g.load_weights(packed)
        o32, o64 = Oracle(cfgs, packed), Oracle(cfgs, packed, "f64")
        t0 = time.time()
        e64, t64 = o64.encode(oh, mask, taps=True)
        print(f"  fp64 encode, {nl} layer(s): {time.time() - t0:.0f} s", flush=True)
        e32, t32 = o32.encode(oh, mask, taps=True)
        if nl == 1:
            for stop, nm in ((1, "row"), (2, "col")):
                g.debug_encoder_stop(stop)
                h = g.encode(tc, tm).cpu().numpy()
                print(f"  {T}x{L} layer0 after {nm}: hip {r(h, t64[stop]):.2e}  o32 {r(t32[stop], t64[stop]):.2e}", flush=True)
            g.debug_encoder_stop(0)
        h = g.encode(tc, tm).cpu().numpy()
        print(f"  {T}x{L} after {nl} layer(s): hip {r(h, e64):.2e}  o32 {r(e32, e64):.2e}", flush=True)
        if nl == 6 and L == 4096:
            z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_f64_t200_l4096.npz"))
            t64s = z["table_0"]
            sc = np.abs(t64s).max()
            th = g.pair_scores_full(torch.from_numpy(h), tm).cpu().numpy()[0]
            th64 = g.pair_scores_full(torch.from_numpy(e64.astype(np.float32)), tm).cpu().numpy()[0]
            th32 = g.pair_scores_full(torch.from_numpy(e32), tm).cpu().numpy()[0]
            print(f"  step-0 table vs fp64 golden: HIP enc + HIP scorer {np.abs(th - t64s).max() / sc:.2e}   fp64 enc + HIP scorer "
                  f"{np.abs(th64 - t64s).max() / sc:.2e}   fp32-oracle enc + HIP scorer {np.abs(th32 - t64s).max() / sc:.2e}", flush=True)
        g.close()


if __name__ == "__main__":
    main()
