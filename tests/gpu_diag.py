#!/usr/bin/env python3
"""Stage-by-stage numeric diagnostics of libnnj_hip.so against the CPU oracle (GPU box).
Usage: python tests/gpu_diag.py [golden-name ...]   Prints max-abs errors per stage."""
import os
import sys
import time
import traceback

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from helpers import load_golden, onehot_f32, split_trace  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

from neuralnj_amd._lib import Nnj  # noqa: E402


def err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return f"maxabs={np.abs(a - b).max():.3e} (ref max {np.abs(b).max():.3e})"


def stage(name, fn):
    t = time.time()
    try:
        msg = fn()
        print(f"[{name}] {msg}  ({time.time() - t:.2f}s)", flush=True)
    except Exception:
        print(f"[{name}] EXCEPTION\n{traceback.format_exc()}", flush=True)


def run(name):
    print(f"===== {name}", flush=True)
    z, cfgs, packed = load_golden(name)
    codes, mask = z["codes"], z["mask"]
    B, T, L = codes.shape
    o = Oracle(cfgs, packed)
    oh = onehot_f32(codes)
    enc_ref, taps = o.encode(oh, mask, taps=True)
    g = Nnj(cfgs)
    g.load_weights(packed)
    tc, tm = torch.from_numpy(codes), torch.from_numpy(mask)

    def enc_stage(stop, ref):
        def f():
            g.debug_encoder_stop(stop)
            out = g.encode(tc, tm).cpu().numpy()
            g.debug_encoder_stop(0)
            return err(out, ref)
        return f

    if int(z["layers"]) >= 1:
        stage("encode:after_row0", enc_stage(1, taps[1]))
        stage("encode:after_col0", enc_stage(2, taps[2]))
    stage("encode:full", enc_stage(0, enc_ref))

    ref_tables = split_trace(z["logits"], T)
    st = torch.from_numpy(enc_ref)
    stage("pair_scores_full", lambda: err(g.pair_scores_full(st, tm).cpu().numpy(), o.pair_scores_full(enc_ref, mask)))
    ij0 = z["merges"][:, 0]
    if T > 2:
        stage("aggregate", lambda: err(g.aggregate(st, ij0).cpu().numpy(), o.aggregate(enc_ref, ij0)))
        st1_ref = o.env_step(enc_ref, ij0)
        stage("env_step", lambda: err(g.env_step(st, ij0).cpu().numpy(), st1_ref))
        lg0 = o.pair_scores_full(enc_ref, mask)
        stage("pair_scores_incr", lambda: err(
            g.pair_scores_incr(torch.from_numpy(st1_ref), tm, ij0, lg0).cpu().numpy(),
            o.pair_scores_incr(st1_ref, mask, ij0, lg0)))
        stage("index_map", lambda: str(np.array_equal(g.score_index_map(ij0, T - 1).cpu().numpy(),
                                                      o.score_index_map(ij0, T - 1))))
    stage("select_pair", lambda: str(np.array_equal(g.select_pair(torch.from_numpy(ref_tables[0].copy()), T)[0].cpu().numpy(),
                                                    z["merges"][:, 0])))

    def rollout_forced():
        r = g.rollout_argmax(tc, tm, forced_merges=z["merges"], want_trace=True, want_state=True)
        lg = r["logits"].cpu().numpy()
        ok = (r["merges"].cpu().numpy() == z["merges"])
        return (f"logits {err(lg, z['logits'])}; state {err(r['state'].cpu().numpy(), enc_ref)}; "
                f"argmax agrees {ok.all(axis=2).mean():.3f}; min golden gap {z['top2_gap'][:, :-1].min() if T > 2 else 0:.3e}")
    stage("rollout(teacher-forced)", rollout_forced)

    def rollout_free():
        r = g.rollout_argmax(tc, tm)
        return f"merges equal golden: {np.array_equal(r['merges'].cpu().numpy(), z['merges'])}"
    stage("rollout(free)", rollout_free)
    g.close()


if __name__ == "__main__":
    names = sys.argv[1:] or ["tiny_b2_t3_l64_s5", "synth_b2_t8_l128_s0", "padded_b2_t8_l128_s3",
                             "plain_b1_t12_l96_s4", "synth_b1_t20_l256_s0", "synth_b1_t50_l1024_s0"]
    for n in names:
        run(n)
