"""200 rows x L sites: HIP and the fp32 oracle against the fp64 oracle over a whole teacher-forced rollout (diagnostic)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import onehot_f32, split_trace
from oracle_lib import Oracle
from neuralnj_amd import synth, utils, weights
from neuralnj_amd._lib import Nnj
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfgs = utils.shipped_config()
packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
g = Nnj(cfgs, "cuda:0"); g.load_weights(packed)
codes = synth.synth_codes(1, T, L, seed=7, gap_frac=0.2)
r = g.rollout_argmax(torch.from_numpy(codes), None, want_trace=True)
m = r["merges"].cpu().numpy(); hip = r["logits"].cpu().numpy()
oh = onehot_f32(codes)
t0 = time.time(); r32 = Oracle(cfgs, packed).rollout_argmax(oh, None, forced_merges=m); t32 = time.time() - t0
t0 = time.time(); r64 = Oracle(cfgs, packed, "f64").rollout_argmax(oh, None, forced_merges=m); t64 = time.time() - t0
sc = np.abs(r64["logits"]).max()
e = lambda a, b: float(np.abs(a - b).max() / sc)
print(f"{T}x{L}: hip-f64 {e(hip, r64['logits']):.2e}  o32-f64 {e(r32['logits'], r64['logits']):.2e}  hip-o32 {e(hip, r32['logits']):.2e}  (oracle seconds {t32:.0f} / {t64:.0f})")
hs, os_, ts = split_trace(hip, T), split_trace(r32["logits"], T), split_trace(r64["logits"], T)
for s in (0, 1, T // 2, T - 66, T - 60, T - 3):
    print(f"  step {s} (n={T - s}): hip-f64 {e(hs[s], ts[s]):.2e}  o32-f64 {e(os_[s], ts[s]):.2e}")
