"""Diagnostic 2: what does a replayed small-batch rollout return -- the previous call's results, or something else?
Fixed input / output buffers on a side stream; new data every call; encoder state and merges against a graph-free handle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from neuralnj_amd import synth, utils, weights
from neuralnj_amd._lib import Nnj
cfgs = utils.shipped_config()
packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
B, T, L = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 12, 128)
g = Nnj(cfgs, "cuda:0"); g.load_weights(packed)
os.environ["NNJ_GRAPH"] = "0"
ref = Nnj(cfgs, "cuda:0"); ref.load_weights(packed)
del os.environ["NNJ_GRAPH"]
buf = torch.empty((B, T, L), dtype=torch.uint8, device="cuda:0")
st = torch.cuda.Stream()
prev = None
for it in range(5):
    c = torch.from_numpy(synth.synth_codes(B, T, L, seed=50 + it, gap_frac=0.2)).cuda()
    buf.copy_(c); torch.cuda.synchronize()
    with torch.cuda.stream(st):
        r = g.rollout_argmax(buf, None, want_trace=True, want_state=True)
        torch.cuda.synchronize()
        cur = {k: v.cpu() for k, v in r.items()}
    del r
    rr = {k: v.cpu() for k, v in ref.rollout_argmax(c, None, want_trace=True, want_state=True).items()}
    line = f"call {it}:"
    for k in ("state", "logits", "merges"):
        ok = bool(torch.equal(cur[k], rr[k]))
        stale = prev is not None and bool(torch.equal(cur[k], prev[k]))
        line += f"  {k}: {'right' if ok else ('STALE (= previous call)' if stale else 'WRONG (neither)')}"
    print(line, flush=True)
    prev = cur
