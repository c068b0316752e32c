"""Diagnostic: does a small-batch rollout REPLAYED from the handle's hipGraph (third and later calls with identical
arguments) give the results of plain launches?  python tools/graph_check.py [stream]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402

cfgs = utils.shipped_config()
packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
use_stream = "stream" in sys.argv
TRACE = "notrace" not in sys.argv
BIG = "nobig" not in sys.argv
B, T, L = (1 if "b1" in sys.argv else 8), 50, 1024
print("args", sys.argv[1:], flush=True)
g = Nnj(cfgs, "cuda:0"); g.load_weights(packed)
os.environ["NNJ_GRAPH"] = "0"
ref = Nnj(cfgs, "cuda:0"); ref.load_weights(packed)
del os.environ["NNJ_GRAPH"]
big = torch.from_numpy(synth.synth_codes(256, T, L, seed=999, gap_frac=0.2)).cuda()
buf = torch.empty((B, T, L), dtype=torch.uint8, device="cuda:0")
st = torch.cuda.Stream() if use_stream else None
for it in range(6):
    c = torch.from_numpy(synth.synth_codes(B, T, L, seed=1000 + it, gap_frac=0.2)).cuda()
    torch.cuda.synchronize()
    if BIG and it in (2, 4):
        g.rollout_argmax(big, None)["merges"].cpu()        # a large rollout in between (sub-batch streams), as tools/e64_scan.py has
    if "gather" in sys.argv:
        # tools/e64_scan.py's pattern: the input is a fresh gather of a big batch, produced on the caller's stream right
        # before the call, no synchronisation in between
        del buf
        big2 = torch.from_numpy(synth.synth_codes(256, T, L, seed=1000 + it, gap_frac=0.2)).cuda()
        g.rollout_argmax(big2, None)["merges"].cpu()
        idx = sorted({int(i) for i in np.linspace(0, 255, B)})
        buf = big2[idx].contiguous()
        c = buf.clone()
        print("  input at", hex(buf.data_ptr()), flush=True)
    else:
        buf.copy_(c)
        torch.cuda.synchronize()
    if st is not None:
        with torch.cuda.stream(st):
            r = g.rollout_argmax(buf, None, want_trace=TRACE)
            if "devsync" in sys.argv:
                torch.cuda.synchronize()
            m, lg = r["merges"].cpu(), (r["logits"].cpu() if TRACE else torch.zeros(1))
    else:
        r = g.rollout_argmax(buf, None, want_trace=TRACE)
        m, lg = r["merges"].cpu(), (r["logits"].cpu() if TRACE else torch.zeros(1))
    torch.cuda.synchronize()
    del r
    rr = ref.rollout_argmax(c, None, want_trace=True)
    m0, lg0 = rr["merges"].cpu(), (rr["logits"].cpu() if TRACE else torch.zeros(1))
    print(f"call {it}: merges equal {bool(torch.equal(m, m0))}  max |logit diff| {float((lg - lg0).abs().max()):.3e}  scale {float(lg0.abs().max()):.1f}", flush=True)
