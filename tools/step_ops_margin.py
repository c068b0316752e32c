"""Diagnostic: per-operation noise of the HIP entry points on EXACT inputs (the fp64 oracle's state, rounded to fp32)
at chosen steps of a golden fixture: merged row (nnj_aggregate) and new scores (nnj_pair_scores_incr) against the fp64
oracle on the same inputs, next to the fp32 oracle.   python tools/step_ops_margin.py NAME s0 s1 ..."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from helpers import load_golden, split_trace, onehot_f32
from neuralnj_amd._lib import Nnj
from oracle_lib import Oracle
name = sys.argv[1]
steps = [int(v) for v in sys.argv[2:]]
z, cfgs, packed = load_golden(name)
B, T, L = z["codes"].shape
mask = z["mask"]; tm = torch.from_numpy(mask)
g = Nnj(cfgs, "cuda:0"); g.load_weights(packed)
o64, o32 = Oracle(cfgs, packed, "f64"), Oracle(cfgs, packed)
o64.set_threads(16); o32.set_threads(16)
state = o64.encode(onehot_f32(z["codes"]), mask)          # float32 array holding the fp64 result rounded once
logits = o64.pair_scores_full(state, mask)
sc = float(np.abs(z["logits"]).max())
for step, n in enumerate(range(T, 2, -1)):
    ij = z["merges"][:, step]
    if step in steps:
        a64 = o64.aggregate(state, ij); a32 = o32.aggregate(state, ij)
        ah = g.aggregate(torch.from_numpy(state), ij).cpu().numpy()
        s_ = np.abs(a64).max()
        print(f"step {step} rows {n}: merged row  hip {np.abs(ah - a64).max() / s_:.2e}  o32 {np.abs(a32 - a64).max() / s_:.2e}", end="   ")
    nstate = o64.env_step(state, ij)
    nl64, new64 = o64.pair_scores_incr(nstate, mask, ij, logits, want_new=True)
    if step in steps:
        nl32 = o32.pair_scores_incr(nstate, mask, ij, logits)
        nlh = g.pair_scores_incr(torch.from_numpy(nstate), tm, torch.from_numpy(ij.copy()), torch.from_numpy(logits)).cpu().numpy()
        print(f"new table  hip {np.abs(nlh - nl64).max() / sc:.2e}  o32 {np.abs(nl32 - nl64).max() / sc:.2e}", flush=True)
    state, logits = nstate, nl64
