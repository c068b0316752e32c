"""Golden tables of BASELINE configs[4] (200 taxa x 4096 sites) from the fp64 build of the CPU oracle.

The reference's own formulation does not fit the build container at this shape, so the pin at the
full size is the oracle -- itself pinned to the reference's outputs on every fixture up to 100 taxa
(tests/test_oracle_golden.py) -- evaluated in float64: a free Argmax run of ONE alignment, its merge list, top-2 gaps
and the complete score tables of eight sampled steps.  tests/test_gpu_parity.py teacher-forces the HIP rollout along
the stored merges and asserts the tables within 1e-4 ("next round" 2b).

Takes tens of minutes of CPU; run once:   python tests/golden/gen_cfg5_f64.py [threads]
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import onehot_f32, split_trace  # noqa: E402
from neuralnj_amd import synth, utils, weights  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

T, L, SEED = 200, 4096, 4242
STEPS = (0, 1, 40, 90, 135, 136, 170, 197)        # rows live: 200, 199, 160, 110, 65, 64, 30, 3


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
    codes = synth.synth_codes_tree(1, T, L, seed=SEED)
    o = Oracle(cfgs, packed, "f64")
    o.set_threads(threads)
    t0 = time.time()
    ref = o.rollout_argmax(onehot_f32(codes), None)
    dt = time.time() - t0
    tabs = split_trace(ref["logits"], T)
    out = dict(codes_sha256=hashlib.sha256(codes.tobytes()).hexdigest(), shape=np.array([T, L]), seed=SEED,
               wseed=0, style="sharp", weights_sha256=weights.digest(packed), merges=ref["merges"][0],
               top2_gap=ref["top2_gap"][0], steps=np.array(STEPS), oracle_seconds=dt,
               scale=float(np.abs(ref["logits"]).max()))
    for s in STEPS:
        out[f"table_{s}"] = tabs[s][0]
    np.savez_compressed(os.path.join(HERE, "cfg5_f64_t200_l4096.npz"), **out)
    print(f"done in {dt:.0f} s, scale {out['scale']:.2f}")


if __name__ == "__main__":
    main()
