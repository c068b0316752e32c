"""Diagnostic: per-step error of the HIP rollout (teacher-forced along the golden merges) against a reference golden,
for the two-pass and the four-pass NJ step, next to the fp32 oracle's.   python tools/golden_steps.py NAME"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from helpers import load_golden, split_trace, onehot_f32
from neuralnj_amd._lib import Nnj
from oracle_lib import Oracle
name = sys.argv[1]
z, cfgs, packed = load_golden(name)
B, T, L = z["codes"].shape
gold = split_trace(z["logits"], T)
scale = float(np.abs(z["logits"]).max())
res = {}
for tp in ("1", "0"):
    os.environ["NNJ_TWO_PASS"] = tp
    g = Nnj(cfgs, "cuda:0"); g.load_weights(packed)
    r = g.rollout_argmax(torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"]), forced_merges=z["merges"], want_trace=True)
    res[tp] = [float(np.abs(a - b).max()) / scale for a, b in zip(split_trace(r["logits"].cpu().numpy(), T), gold)]
    g.close()
o = Oracle(cfgs, packed); o.set_threads(16)
ro = o.rollout_argmax(onehot_f32(z["codes"]), z["mask"], forced_merges=z["merges"])
res["o32"] = [float(np.abs(a - b).max()) / scale for a, b in zip(split_trace(ro["logits"], T), gold)]
o64 = Oracle(cfgs, packed, "f64"); o64.set_threads(16)
t64 = split_trace(o64.rollout_argmax(onehot_f32(z["codes"]), z["mask"], forced_merges=z["merges"])["logits"], T)
os.environ["NNJ_TWO_PASS"] = "1"
g = Nnj(cfgs, "cuda:0"); g.load_weights(packed)
hip = split_trace(g.rollout_argmax(torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"]), forced_merges=z["merges"], want_trace=True)["logits"].cpu().numpy(), T)
res["hip64"] = [float(np.abs(a - b).max()) / scale for a, b in zip(hip, t64)]
res["gold64"] = [float(np.abs(a - b).max()) / scale for a, b in zip(gold, t64)]
res["o3264"] = [float(np.abs(a - b).max()) / scale for a, b in zip(split_trace(ro["logits"], T), t64)]
print("vs fp64: HIP max %.2e (step 0 %.2e)   reference golden max %.2e (step 0 %.2e)   fp32 oracle max %.2e" % (
    max(res["hip64"]), res["hip64"][0], max(res["gold64"]), res["gold64"][0], max(res["o3264"])))
top = np.argsort(res["hip64"])[::-1][:8]
print("largest HIP-vs-fp64 steps:", [(int(s_), T - int(s_), "%.2e" % res["hip64"][s_], "gold %.2e" % res["gold64"][s_], "o32 %.2e" % res["o3264"][s_]) for s_ in top])
print("scale", scale)
for s in list(range(0, T - 1, max(1, (T - 1) // 24))) + [T - 2]:
    print(f"step {s:3d} rows {T - s:3d}  two-pass {res['1'][s]:.2e}  four-pass {res['0'][s]:.2e}  fp32 oracle {res['o32'][s]:.2e}")
print("max", max(res['1']), max(res['0']), max(res['o32']))
