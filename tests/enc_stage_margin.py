"""Encoder noise by stage (diagnostic): HIP and fp32 oracle vs fp64 oracle after the row-attention block, the column
block and the FFN of layer 0, and after 1, 2, 3, 6 layers."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import onehot_f32  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402


def run(T, L, style="sharp"):
    codes = synth.synth_codes_tree(1, T, L, seed=4242)
    mask = np.zeros((1, L), bool)
    oh = onehot_f32(codes)
    tc, tm = torch.from_numpy(codes), torch.from_numpy(mask)
    r = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
    for nl in (1, 2, 3, 6):
        cfgs = utils.shipped_config()
        cfgs.model.num_enc_layers = nl
        full = utils.shipped_config()
        st = weights.seeded_state(full, 0, style)
        st = {k: v for k, v in st.items() if not k.startswith("seq_emb_layers.") or int(k.split(".")[1]) < nl}
        packed = weights.pack(cfgs, st)
        g = Nnj(cfgs, "cuda:0")
        g.load_weights(packed)
        o32, o64 = Oracle(cfgs, packed), Oracle(cfgs, packed, "f64")
        e64, t64 = o64.encode(oh, mask, taps=True)
        e32, t32 = o32.encode(oh, mask, taps=True)
        if nl == 1:
            for stop, nm in ((1, "row"), (2, "col")):
                g.debug_encoder_stop(stop)
                h = g.encode(tc, tm).cpu().numpy()
                print(f"  {T}x{L} {style} layer0 after {nm}: hip {r(h, t64[stop]):.2e}  o32 {r(t32[stop], t64[stop]):.2e}")
            g.debug_encoder_stop(0)
        h = g.encode(tc, tm).cpu().numpy()
        print(f"  {T}x{L} {style} after {nl} layer(s): hip {r(h, e64):.2e}  o32 {r(e32, e64):.2e}", flush=True)
        g.close()


if __name__ == "__main__":
    if len(sys.argv) >= 3:                                   # python tests/enc_stage_margin.py T L [style]
        run(int(sys.argv[1]), int(sys.argv[2]), *(sys.argv[3:4]))
    else:
        run(200, 160)
        run(50, 1024)
        run(256, 20)
