"""BASELINE configs[4] (200 taxa x 4096 sites) on the GPU: timing of the Argmax rollout at B = 1, 2, 4 and -- optionally,
it takes minutes of CPU -- the whole rollout of one alignment against the fp32 / fp64 oracle.
    python tests/config5_run.py [--oracle] [out.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import onehot_f32  # noqa: E402
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402


def main():
    want_oracle = "--oracle" in sys.argv
    outp = [a for a in sys.argv[1:] if not a.startswith("--")]
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
    g = Nnj(cfgs, "cuda:0")
    g.load_weights(packed)
    T, L = 200, 4096
    res = {"shape": [T, L], "weights": "seed 0, sharp"}
    for B in (1, 2, 4):
        codes = torch.from_numpy(synth.synth_codes(B, T, L, seed=7, gap_frac=0.2)).cuda()
        g.rollout_argmax(codes, None)["merges"].cpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            g.rollout_argmax(codes, None)["merges"].cpu()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[f"B{B}"] = {"s_per_rollout": dt, "trees_per_sec": B / dt}
        print(f"B={B}: {dt:.3f} s per rollout, {B / dt:.2f} trees/s", flush=True)
    g.profile_enable(True)
    codes = torch.from_numpy(synth.synth_codes(1, T, L, seed=7, gap_frac=0.2)).cuda()
    r = g.rollout_argmax(codes, None, want_trace=True)
    torch.cuda.synchronize()
    prof = g.profile_read()
    g.profile_enable(False)
    res["kernel_ms_B1"] = {k: round(v[0], 2) for k, v in prof.items() if v[1]}
    print(res["kernel_ms_B1"], flush=True)
    # Search mode at the BASELINE shape (infer_opt=Search): 8 sampled rollouts of ONE alignment, encoded once
    u = torch.rand((8, T - 1), generator=torch.Generator().manual_seed(1))
    g.rollout_sample(codes, None, u, temperature=1.0, replicas=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rs = g.rollout_sample(codes, None, u, temperature=1.0, replicas=8)
    keys = g.topology_hash(rs["merges"]).cpu()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g.check_numeric()
    res["search_8_rollouts"] = {"s": dt, "rollouts_per_sec": 8 / dt, "distinct_topologies": int(torch.unique(keys).numel())}
    print("search: 8 sampled rollouts of one 200 x 4096 alignment (encoded once):", res["search_8_rollouts"], flush=True)
    g.profile_enable(True)
    g.rollout_sample(codes, None, u, temperature=1.0, replicas=8)
    torch.cuda.synchronize()
    prof = g.profile_read()
    g.profile_enable(False)
    res["kernel_ms_search_round"] = {k: round(v[0], 2) for k, v in prof.items() if v[1]}
    print("search round, per kernel kind:", res["kernel_ms_search_round"], flush=True)
    if want_oracle:
        from oracle_lib import Oracle
        m = r["merges"].cpu().numpy()
        oh = onehot_f32(codes.cpu().numpy())
        t0 = time.time()
        ref = Oracle(cfgs, packed).rollout_argmax(oh, None, forced_merges=m)
        res["oracle_fp32_seconds"] = time.time() - t0
        sc = float(np.abs(ref["logits"]).max())
        hip = r["logits"].cpu().numpy()
        res["score_err_rel_vs_fp32_oracle"] = float(np.abs(hip - ref["logits"]).max()) / sc
        decisive = ref["top2_gap"] > 4e-4 * sc
        res["merges_equal_on_decisive_steps"] = bool((ref["merges"][decisive] == m[decisive]).all())
        res["decisive_steps"] = [int(decisive.sum()), int(decisive.size)]
        print("vs fp32 oracle:", res["score_err_rel_vs_fp32_oracle"], res["merges_equal_on_decisive_steps"], flush=True)
        t0 = time.time()
        ref64 = Oracle(cfgs, packed, "f64").rollout_argmax(oh, None, forced_merges=m)
        res["oracle_fp64_seconds"] = time.time() - t0
        res["score_err_rel_vs_fp64"] = float(np.abs(hip - ref64["logits"]).max()) / sc
        res["fp32_oracle_err_rel_vs_fp64"] = float(np.abs(ref["logits"] - ref64["logits"]).max()) / sc
        print("vs fp64:", res["score_err_rel_vs_fp64"], "oracle32 vs fp64:", res["fp32_oracle_err_rel_vs_fp64"], flush=True)
    if outp:
        with open(outp[0], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
