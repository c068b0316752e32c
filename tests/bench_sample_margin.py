"""Accuracy of the HIP path on the bench.py workload itself (a tool, not a test): k alignments of the synthetic
256 x 50 x 1024 batch, HIP tables against the fp32 oracle and both against the fp64 oracle (teacher-forced along the
HIP merges).  python tests/bench_sample_margin.py [k]"""
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

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfgs = utils.shipped_config()
packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
g = Nnj(cfgs, "cuda:0")
g.load_weights(packed)
codes = synth.synth_codes(256, 50, 1024, seed=1000, gap_frac=0.2)
idx = sorted({int(i) for i in np.linspace(0, 255, k)})
sub = codes[idx]
r = g.rollout_argmax(torch.from_numpy(sub), None, want_trace=True)
m = r["merges"].cpu().numpy()
hip = r["logits"].cpu().numpy()
oh = synth.codes_to_onehot(sub).astype(np.float32)
mask = np.zeros((len(idx), 1024), bool)
o32 = Oracle(cfgs, packed).rollout_argmax(oh, mask, forced_merges=m)["logits"]
o64 = Oracle(cfgs, packed, "f64").rollout_argmax(oh, mask, forced_merges=m)["logits"]
for b in range(len(idx)):
    scale = max(float(np.abs(o32[b]).max()), 1.0)
    e = lambda a, c: float(np.abs(a[b] - c[b]).max()) / scale
    print(f"tree {idx[b]:3d} scale {scale:7.1f}  hip-vs-o32 {e(hip, o32):.2e}  hip-vs-f64 {e(hip, o64):.2e}  o32-vs-f64 {e(o32, o64):.2e}",
          flush=True)
