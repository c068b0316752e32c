"""Long alignments (a tool, not a test): 50 x 4096 and 64 x 2048 against the fp32 oracle, and the time of a 32-alignment
rollout at 4096 sites.  Lives under tests/ because it calls the oracle.  python tests/long_case.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from neuralnj_amd import synth, utils, weights
from neuralnj_amd._lib import Nnj
from oracle_lib import Oracle
cfgs = utils.shipped_config()
packed = weights.pack(cfgs, weights.seeded_state(cfgs, 5, "sharp"))
g = Nnj(cfgs, "cuda:0"); g.load_weights(packed)
for (B, T, L) in ((1, 50, 4096), (1, 64, 2048)):
    codes = synth.synth_codes_tree(B, T, L, 9)
    mask = np.zeros((B, L), bool)
    r = g.rollout_argmax(torch.from_numpy(codes), torch.from_numpy(mask), want_trace=True)
    merges = r["merges"].cpu().numpy(); logits = r["logits"].cpu().numpy(); g.check_numeric()
    o = Oracle(cfgs, packed)
    t0 = time.time()
    ref = o.rollout_argmax(synth.codes_to_onehot(codes).astype(np.float32), mask, forced_merges=merges)
    scale = max(float(np.abs(ref["logits"]).max()), 1.0)
    err = float(np.abs(logits - ref["logits"]).max()) / scale
    dec = ref["top2_gap"] > 4e-4 * scale
    print(B, T, L, "err", err, "flips", int((ref["merges"][dec] != merges[dec]).any(axis=-1).sum()), "oracle s", round(time.time() - t0, 1), flush=True)
codes = torch.from_numpy(synth.synth_codes(32, 50, 4096, seed=3, gap_frac=0.2)).cuda()
for _ in range(2):
    g.rollout_argmax(codes, None)["merges"].cpu()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2):
    g.rollout_argmax(codes, None)["merges"].cpu()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
print("B=32 50x4096:", round(dt * 1e3, 1), "ms per rollout,", round(32 / dt, 1), "trees/s")
g.check_numeric()
