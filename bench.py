#!/usr/bin/env python3
"""Throughput benchmark of the Argmax hot path (encoder + neural NJ loop) on MI355X.

  python bench.py --gpus N --steps K --warmup W
      N = 1: runs here.  N > 1 from a bare shell (WORLD_SIZE unset): starts the N ranks itself -- a CHILD
      `python -m torch.distributed.run --nproc-per-node N bench.py ...`, launched before this process touches the GPU --
      and relays rank 0's line and the exit code; more ranks than GPUs, or a line whose n_gpus is not N, is an error.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

A step = one device-resident Argmax rollout (nnj_rollout_argmax) of one batch of
synthetic 50-taxa x 1024-site MSAs per GPU, site codes already in HBM, merge lists
copied back to the host at the end of the step.  Batches of independent MSAs are sharded
over the ranks with no data-path collective (weak scaling: 256 MSAs per GPU).
Rank 0 prints ONE JSON line (contract in the task statement); extra objects:
  roofline     : dominant kernel (by HIP-event time inside the timed region)
  cpu_baseline : the CPU oracle ("port") timed on this host's cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def _visible_gpus():
    """Device count, asked in a CHILD process so that this one never initialises the GPU."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                       capture_output=True, text=True, timeout=600)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise SystemExit(f"bench.py: could not count the GPUs of this node: {r.stderr[-500:]}")


def _self_launch():
    """`python bench.py --gpus N` from a bare shell (no launcher, WORLD_SIZE unset), N > 1: this process -- which has
    made NO GPU call and imported nothing that makes one -- starts `python -m torch.distributed.run --nproc-per-node N
    bench.py ...` as a CHILD (subprocess, never exec), relays rank 0's JSON line and the exit code.  Asking for more
    ranks than the node has GPUs is an error, never a silent one-GPU run (NNJ_BENCH_SHARE_GPU=1, the one-GPU
    rehearsal, puts every rank on cuda:0 and is exempt)."""
    n, argv = 1, sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n < 1:
        raise SystemExit(f"bench.py: --gpus {n}")
    if "WORLD_SIZE" in os.environ or n == 1:
        return
    import socket
    import subprocess
    if os.environ.get("NNJ_BENCH_SHARE_GPU") != "1":
        have = _visible_gpus()
        if n > have:
            raise SystemExit(f"bench.py: --gpus {n} but this node shows {have} GPU(s): refusing to measure fewer GPUs than asked")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    for x in r.stdout.splitlines():
        if not x.startswith("{"):
            print(x, file=sys.stderr)
    if r.returncode != 0:
        for x in lines:
            print(x, flush=True)
        raise SystemExit(r.returncode)
    if not lines:
        raise SystemExit("bench.py: the launched ranks printed no result line")
    got = json.loads(lines[-1]).get("n_gpus")
    if got != n:
        raise SystemExit(f"bench.py: asked for --gpus {n}, the run reports n_gpus = {got}")
    print(lines[-1], flush=True)
    raise SystemExit(0)


if __name__ == "__main__":
    _self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 (v_mfma_f32_32x32x16_f16, 32 cycles)
# fp32 GEMMs run as "f16x3": every fp32 operand is split into two fp16 pieces and a product is three fp16 MFMA
# products with fp32 accumulation, so the matrix-pipe ceiling for ALGORITHMIC fp32 flops is the
# fp16 peak / 3.
PEAK_F32_VIA_F16X3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
PEAK_HBM_GBS = 8000.0
# the per-step kernels of the NJ loop (profile kinds of libnnj_hip.so)
NJ_STEP_KINDS = ("k_pair_alpha_incr", "k_alpha_softmax", "k_pair_score_incr", "k_assemble_argmax", "k_agg_alpha",
                 "k_agg_finish", "k_step_small")


def kernel_models(B, T, L, layers):
    """ALGORITHMIC FLOPs and HBM bytes PER LAUNCH of each kernel kind: SURVEY.md section 8(d)'s
    per-tree formulas split by kernel x the B trees one launch processes.
    For the per-step NJ kernels (48 launches of different n per rollout) the figure is the
    mean over n = T-1..2."""
    C, D, F = L, 64, 256
    N = T * C
    E = ((T * 8 + 15) // 16) * 16
    P = T * (T - 1) // 2
    row = C * D * 4
    m = {}
    X6, F32 = PEAK_F32_VIA_F16X3_TFLOPS, PEAK_F32_MFMA_TFLOPS
    heads = 8
    planes = heads * C * E * 4.0              # bytes of one of Q6/K6/V6 per tree (2 fp16 planes)
    sbytes = heads * C * C * 4.0              # the score images of one tree
    m["k_qkv6"] = dict(flops=B * N * 6.0 * D * D, bytes=B * (N * D * 4 + 3.0 * planes), peak=X6)
    m["k_row_s"] = dict(flops=B * 2.0 * C * C * T * D, bytes=B * (2.0 * planes + sbytes), peak=X6)
    m["k_row_pv"] = dict(flops=B * 2.0 * C * C * T * D, bytes=B * (planes + sbytes + heads * C * E * 4.0), peak=X6)
    m["k_tok1"] = dict(flops=B * (N * 10.0 * D * D + 4.0 * T * T * C * D), bytes=B * 3.0 * N * D * 4, peak=X6)
    m["k_ffn"] = dict(flops=B * N * 4.0 * D * F, bytes=B * 6.0 * N * D * 4, peak=X6)
    m["k_embed"] = dict(flops=B * N * 8.0 * D, bytes=B * (N + N * D * 4.0), peak=F32)
    # reference per (pair, site): W_h 2D^2, W_q 2D^2, alpha 2nD | x_g 2nD, W_g 2D^2, s_out 2D^2+2D
    m["k_pair_alpha"] = dict(flops=B * P * C * (4.0 * D * D + 2.0 * T * D), bytes=B * 3.0 * T * row, peak=X6)
    m["k_pair_score"] = dict(flops=B * P * C * (2.0 * T * D + 4.0 * D * D + 2.0 * D), bytes=B * 2.0 * T * row, peak=X6)
    ns = list(range(T - 1, 1, -1))
    m["k_pair_alpha_incr"] = dict(flops=B * sum(n * C * (4.0 * D * D + 2.0 * n * D) for n in ns if n > 2) / max(1, len([n for n in ns if n > 2])),
                                  bytes=B * sum((n + 1.0) * row for n in ns) / len(ns), peak=X6)
    m["k_pair_score_incr"] = dict(flops=B * sum(n * C * (2.0 * n * D + 4.0 * D * D + 2.0 * D) for n in ns) / len(ns),
                                  bytes=B * sum((n + 1.0) * row for n in ns) / len(ns), peak=X6)
    return m


def cpu_baseline(cfgs, packed, T, L, budget_s=8.0):
    """Times the CPU oracle (oracle/nnj_oracle.c, kind "port") on all host cores."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle_lib import Oracle
    o = Oracle(cfgs, packed)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    avail = cores
    cores = o.set_threads(min(cores, 16))    # the GPU box's CPU share per GPU
    codes = synth.synth_codes(1, T, L, seed=4242, gap_frac=0.2)
    oh = synth.codes_to_onehot(codes).astype(np.float32)
    mask = np.zeros((1, L), dtype=bool)
    n, t0 = 0, time.time()
    while True:
        o.rollout_argmax(oh, mask)
        n += 1
        el = time.time() - t0
        if el >= budget_s or n >= 64:
            break
    out = dict(value=n / el, unit="trees/sec", cores=cores, kind="port",
               cores_note=f"{cores} OpenMP threads of the {avail} cores this process may run on ({os.cpu_count()} in the node): "
                          "the GPU box's CPU share per GPU",
               sample=f"{n} single-MSA Argmax rollouts of {T} taxa x {L} sites, fp32 OpenMP oracle, {el:.1f} s")
    # CPU model, and a one-thread figure beside the all-cores one (one rollout)
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu_model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        out["cpu_model"] = "unknown"
    o.set_threads(1)
    t1 = time.time()
    o.rollout_argmax(oh, mask)
    out["one_thread"] = dict(value=1.0 / (time.time() - t1), unit="trees/sec", sample="1 rollout, 1 thread")
    return out


def verify_sample(g, cfgs, packed, codes, merges, T, L, k=8, threads=16, tol=1e-4):
    """SURVEY 8(d) / VERDICT r1 item 1(d): a fixed sample of >= 8 trees of THIS rank's timed batch against the CPU
    oracle, outside the timed region.  The sampled alignments are rolled out again alone with the score trace.
      * RF: the HIP free run against the FREE runs of the fp32 oracle and of its fp64 build (Robinson-Foulds
        distance of the trees; helpers.free_run_verdict arbitrates a differing merge list: first divergent step a
        near-tie by fp64 and the HIP pick the runner-up -- anything else fails);
      * scores: HIP tables against the fp32 and fp64 oracles teacher-forced along HIP's merges.
    `ok` is False -- and bench.py exits non-zero -- when a merge list fails the gate, or when the HIP tables are farther
    than `tol` (relative to the largest score of the sampled tables) from the fp64 evaluation.  The distance from the
    fp32 oracle -- itself 4e-5 .. 1.5e-4 from fp64 on this workload, a checker as noisy as the tolerance -- is reported,
    not gated.  Element-relative errors (|d| / |ref| per entry) are reported for the entries of at least 1 % of the
    scale and for the top-5 entries of every table (the ones that decide merges)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import free_run_verdict, newick_from_merges, split_trace
    from oracle_lib import Oracle
    B = codes.shape[0]
    idx = sorted({int(i) for i in np.linspace(0, B - 1, min(k, B))})
    sub = codes[idx].contiguous()
    r = g.rollout_argmax(sub, None, want_trace=True)
    m = r["merges"].cpu().numpy()
    same = bool(np.array_equal(m, merges.numpy()[idx]))
    oh = synth.codes_to_onehot(sub.cpu().numpy()).astype(np.float32)
    nomask = np.zeros((len(idx), L), bool)
    o32, o64 = Oracle(cfgs, packed), Oracle(cfgs, packed, "f64")
    o32.set_threads(threads)
    free32 = o32.rollout_argmax(oh, nomask)
    free64 = o64.rollout_argmax(oh, nomask)
    same32 = np.array([np.array_equal(free32["merges"][i], m[i]) for i in range(len(idx))])
    same64 = np.array([np.array_equal(free64["merges"][i], m[i]) for i in range(len(idx))])
    # teacher-forced along HIP's merges (a free run that took the same merges IS that run)
    ref, ref64 = dict(free32), dict(free64)
    if not same32.all():
        ref = o32.rollout_argmax(oh, nomask, forced_merges=m)
    if not same64.all():
        ref64 = o64.rollout_argmax(oh, nomask, forced_merges=m)
    hip = r["logits"].cpu().numpy()
    scale = max(float(np.abs(ref["logits"]).max()), 1.0)
    rel = lambda a, b: float(np.abs(a - b).max()) / scale  # noqa: E731
    keys = [f"taxon{i + 1}" for i in range(T)]
    hip_t = split_trace(hip, T)
    rf_rows, gate_ok = [], True
    for name, free, same_v, orc in (("fp32", free32, same32, o64), ("fp64", free64, same64, o64)):
        ft = split_trace(free["logits"], T)
        for i in range(len(idx)):
            def truth(i=i, free=free):
                t = orc.rollout_argmax(oh[i:i + 1], nomask[i:i + 1], forced_merges=free["merges"][i:i + 1])
                return [x[0] for x in split_trace(t["logits"], T)]
            try:
                row = free_run_verdict(m[i], [x[i] for x in hip_t], free["merges"][i], [x[i] for x in ft],
                                       newick_from_merges(free["merges"][i], keys), keys, truth)
                row["gate"] = "pass"
            except AssertionError as e:
                row = dict(rf=None, gate="FAIL", why=str(e)[:300])
                gate_ok = False
            rf_rows.append(dict(tree=idx[i], oracle=name, **row))
    e32, e64, o3264 = rel(hip, ref["logits"]), rel(hip, ref64["logits"]), rel(ref["logits"], ref64["logits"])
    # where the largest distance from the fp64 evaluation sits: (sampled tree, step, rows alive, pair of the table), and the
    # distance per step (the maximum over the sampled trees) -- VERDICT r4 item 2: "which table, which step"
    per_step = [float(np.abs(a - b_).max()) / scale for a, b_ in zip(hip_t, split_trace(ref64["logits"], T))]
    ws_ = int(np.argmax(per_step))
    wd = np.abs(hip_t[ws_] - split_trace(ref64["logits"], T)[ws_])
    wt, wp = np.unravel_index(int(np.argmax(wd)), wd.shape)
    worst = dict(tree=idx[int(wt)], step=ws_, rows_alive=T - ws_, pair_index=int(wp), err_rel=per_step[ws_],
                 fp32_oracle_err_rel_same_table=float(np.abs(split_trace(ref["logits"], T)[ws_] -
                                                             split_trace(ref64["logits"], T)[ws_]).max()) / scale,
                 err_rel_by_step_first8=[round(x, 7) for x in per_step[:8]],
                 err_rel_max_after_step8=round(max(per_step[8:]), 7) if len(per_step) > 8 else None)
    scores_ok = e64 <= tol
    # element-relative error: entries of at least 1 % of the scale; and the top-5 entries of every table
    r64 = ref64["logits"].astype(np.float64)
    big = np.abs(r64) >= 0.01 * scale
    elem = float((np.abs(hip - r64)[big] / np.abs(r64)[big]).max()) if big.any() else 0.0
    top_elem = 0.0
    for th_, t64_ in zip(hip_t, split_trace(ref64["logits"], T)):
        k5 = min(5, t64_.shape[1])
        idx5 = np.argsort(-t64_, axis=1)[:, :k5]
        a, b_ = np.take_along_axis(th_, idx5, 1), np.take_along_axis(t64_, idx5, 1)
        top_elem = max(top_elem, float((np.abs(a - b_) / np.maximum(np.abs(b_), 1e-30)).max()))
    decisive = ref["top2_gap"] > 4e-4 * scale
    return dict(trees=len(idx), ok=bool(gate_ok and scores_ok and same), same_merges_as_timed_run=same,
                rf_vs_fp32_oracle=[r_["rf"] for r_ in rf_rows if r_["oracle"] == "fp32"],
                rf_vs_fp64_oracle=[r_["rf"] for r_ in rf_rows if r_["oracle"] == "fp64"],
                merge_lists_identical_to_fp32_oracle=int(same32.sum()), merge_lists_identical_to_fp64_oracle=int(same64.sum()),
                arbitrated=[r_ for r_ in rf_rows if r_.get("gate") != "pass" or not r_.get("identical_merges", True)],
                decisive_steps=int(decisive.sum()), steps=int(decisive.size),
                merges_equal_on_decisive_steps=bool((ref["merges"][decisive] == m[decisive]).all()),
                score_tolerance=tol, score_err_rel_vs_fp32_oracle=e32, score_err_rel_vs_fp64=e64,
                fp32_oracle_err_rel_vs_fp64=o3264, worst_table=worst, scores_ok=bool(scores_ok), rf_gate_ok=bool(gate_ok),
                score_err_definition="max |hip - ref| over all entries of the sampled tables / max |ref| (scale-relative)",
                elem_rel_err_vs_fp64_entries_over_1pct_of_scale=elem, elem_rel_err_vs_fp64_top5_of_each_table=top_elem,
                elem_rel_note="the scores are signed logits, not distances: an entry near zero makes |d| / |ref| arbitrarily "
                              "large (the top-5 figure is dominated by such entries), which is why the tolerance is "
                              "taken relative to the scale of the table",
                note="the run FAILS (exit 3) when a merge list leaves the oracle's without being an fp64-certified "
                     "near-tie, or when the scale-relative score error against the fp64 evaluation exceeds the "
                     "tolerance; the distance from the fp32 oracle is reported only")


def step_api_path(g, codes, mask, T, merges_ref):
    """Throughput of a maintainer's binding of the fused step: nnj_encode, nnj_pair_scores_full, nnj_select_pair, then
    T-2 x nnj_step (one C-ABI call per loop iteration, dense state tensor in and out as the reference's env.step
    returns it, no host round trip).  Same kernels as the rollout plus one dense gather of the live rows per step."""
    def run():
        state = g.encode(codes, mask)
        logits = g.pair_scores_full(state, mask)
        ij, _ = g.select_pair(logits, T)
        ms = [ij]
        for n in range(T - 1, 1, -1):
            r = g.step(state, mask, ij, logits)
            state, logits, ij = r["state"], r["logits"], r["ij"]
            ms.append(ij)
        return torch.stack(ms, 1).cpu()
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    B = codes.shape[0]
    return dict(workload=f"Batch={B}: nnj_encode + nnj_pair_scores_full + {T - 2} x nnj_step (fused step, dense state in/out)",
                trees_per_sec=B / dt, ms_per_rollout=1e3 * dt,
                merges_equal_fused_rollout=bool(torch.equal(m, merges_ref)))


def compat_path(cfgs, packed, codes, T, L, dev):
    """Throughput of the API-compatible step-by-step path (neuralnj_amd.rollout.reinforce_rollout_argmax: the
    reference's own call sequence through model.PhyloATTN / environment.PhyInferEnv, one host round trip per
    step) on the same batch: what "finetune_rl_search.py drops in unchanged" delivers, next to the fused
    nnj_rollout_argmax figure that is `value`."""
    from neuralnj_amd.environment import PhyInferEnv
    from neuralnj_amd.model import PhyloATTN
    from neuralnj_amd.rollout import reinforce_rollout_argmax
    agent = PhyloATTN(cfgs)
    sd = weights.seeded_state(cfgs, 0, "sharp")
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    agent = agent.to(dev)
    B = codes.shape[0]
    keys = [[f"taxon{i + 1}" for i in range(T)] for _ in range(B)]
    c = codes.cpu().numpy()
    batch = {"data": torch.from_numpy(synth.codes_to_onehot(c)), "seqs": [[""] * T for _ in range(B)],
             "seq_keys": keys, "seq_weights": torch.ones((B, L), dtype=torch.float32)}
    env = PhyInferEnv(cfgs, dev)
    reinforce_rollout_argmax(batch, agent, env)            # warm-up (workspace, weights)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    env = PhyInferEnv(cfgs, dev)
    _, _, m = reinforce_rollout_argmax(batch, agent, env)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    out = dict(workload=f"Batch={B}, {T}x{L}, reinforce_rollout_argmax (model/environment API, per-step host loop)",
               trees_per_sec=B / dt, ms_per_rollout=1e3 * dt)
    # ... and the loop exactly as the reference writes it: host index map per step, one .item() per
    # alignment and step (12,544 synchronising reads per rollout of 256), log-softmax of every table
    from neuralnj_amd.rollout import reinforce_rollout_reference_pattern
    env = PhyInferEnv(cfgs, dev)
    reinforce_rollout_reference_pattern(batch, agent, env)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    env = PhyInferEnv(cfgs, dev)
    _, _, m_ref = reinforce_rollout_reference_pattern(batch, agent, env)
    torch.cuda.synchronize(dev)
    dt2 = time.perf_counter() - t0
    out["reference_loop_unchanged"] = dict(
        workload="the same rollout driven as finetune_rl_search.py:100-189 drives it, host side unchanged: host-built index "
                 "map uploaded per step, .item() per alignment, log-softmax of every table",
        trees_per_sec=B / dt2, ms_per_rollout=1e3 * dt2, merges_equal=bool(np.array_equal(m_ref, m)))
    return out, m


def finetune_episode(cfgs, T, L, dev):
    """Seconds per episode of the Finetune mode at B = 1 on the bench shape: one sampled rollout on the
    fused kernels, the episode replayed with gradients (neuralnj_amd.rollout.reinforce_loss: forward and backward
    kernels of libnnj_train_hip.so), backward, clipped Adam step.  A side figure; never fails the bench."""
    try:
        from neuralnj_amd.environment import PhyInferEnv
        from neuralnj_amd.model import PhyloATTN
        from neuralnj_amd.rollout import reinforce_loss
        agent = PhyloATTN(cfgs)
        sd = weights.seeded_state(cfgs, 0, "plain")
        agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        agent = agent.to(dev).eval()
        opt = torch.optim.Adam(agent.parameters(), lr=1e-5)
        c = synth.synth_codes_tree(1, T, L, seed=3)
        batch = {"data": torch.from_numpy(synth.codes_to_onehot(c)), "seqs": [synth.codes_to_seqs(c[0])],
                 "seq_keys": [[f"taxon{i + 1}" for i in range(T)]], "seq_weights": torch.ones((1, L), dtype=torch.float32)}
        times = []
        torch.cuda.synchronize(dev)
        torch.cuda.reset_peak_memory_stats(dev)
        held = torch.cuda.memory_allocated(dev)              # workspaces of the earlier legs that are still alive
        for ep in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            with torch.no_grad():
                u = torch.from_numpy(np.random.default_rng(ep).random((1, T - 1)).astype(np.float32))
                m = agent._context().rollout_sample(torch.from_numpy(c), None, u, temperature=1.0, replicas=1)["merges"]
            opt.zero_grad()
            loss, _ = reinforce_loss(batch, agent, PhyInferEnv(cfgs, dev), m.cpu().numpy(), np.array([1.0], np.float32), 0.5)
            loss.backward()
            torch.nn.utils.clip_grad_value_(agent.parameters(), clip_value=1.0)
            opt.step()
            torch.cuda.synchronize(dev)
            if ep:
                times.append(time.perf_counter() - t0)
        return {"workload": f"Finetune episode, B=1, {T}x{L}: sampled rollout + differentiable replay + backward + Adam",
                "s_per_episode": float(np.median(times)),
                "peak_mem_gb": (torch.cuda.max_memory_allocated(dev) - held) / 2 ** 30,
                "peak_mem_note": "peak of the episodes themselves (allocations of the earlier bench legs that are still alive, "
                                 f"{held / 2 ** 30:.1f} GB, subtracted)",
                "note": "first, unfused fp32 path (DESIGN.md 14); gradients pinned to the reference's (tests/test_gpu_finetune.py)"}
    except Exception as e:                                    # pragma: no cover
        return {"error": f"{type(e).__name__}: {e}"}


def search_config5(cfgs, packed, dev, replicas=8):
    """BASELINE configs[4] beside the headline: infer_opt=Search at 200 taxa x 4096 sites.  Two figures:
      s_per_round               `replicas` sampled rollouts of ONE alignment (encoded once, nnj_rollout_sample) + the device-side
                                topology keys -- the sampling half of a round;
      s_per_round_with_scoring  a COMPLETE round of the reference's RL_Search (finetune_rl_search.py:338-427) through
                                neuralnj_amd.rollout.search_rollouts: sample, drop duplicate topologies, estimate the GTR+I+G
                                parameters (`opt_model=True`), optimise the branch lengths in three sweeps (`iters=3`), score
                                every distinct tree on the GPU (nnj_tree_optimize), rank, build the host trees."""
    try:
        g5 = Nnj(cfgs, dev)
        g5.load_weights(packed)
        T5, L5 = 200, 4096
        c5 = synth.synth_codes_tree(1, T5, L5, seed=4242)
        one = torch.from_numpy(c5).to(dev)
        u = torch.from_numpy(np.random.default_rng(5).random((replicas, T5 - 1)).astype(np.float32))
        g5.rollout_sample(one, None, u, temperature=1.0, replicas=replicas)["merges"].cpu()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        r = g5.rollout_sample(one, None, u, temperature=1.0, replicas=replicas)
        keys = g5.topology_hash(r["merges"]).cpu()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        g5.check_numeric()
        out = {"workload": f"Search mode, {T5} taxa x {L5} sites, {replicas} sampled rollouts of one alignment (BASELINE configs[4])",
               "s_per_round": dt, "rollouts_per_sec": replicas / dt, "distinct_topologies": int(torch.unique(keys).numel())}
        g5.close()
        from neuralnj_amd.environment import PhyInferEnv
        from neuralnj_amd.model import PhyloATTN
        from neuralnj_amd.rollout import search_rollouts
        agent = PhyloATTN(cfgs)
        sd = weights.seeded_state(cfgs, 0, "sharp")
        agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        agent = agent.to(dev)
        batch = {"codes": torch.from_numpy(c5), "seqs": [[""] * T5], "seq_keys": [[f"taxon{i + 1}" for i in range(T5)]],
                 "seq_weights": torch.ones((1, L5), dtype=torch.float32)}
        times = []
        for rep in range(2):                                  # (the first round also sizes the workspaces)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            det = {}
            _, best_ll, trees = search_rollouts(batch, agent, PhyInferEnv(cfgs, dev), replicas, seed=5, temperature=1.0,
                                                model="auto", sweeps=3, details=det)
            torch.cuda.synchronize(dev)
            times.append(time.perf_counter() - t0)
        out.update(s_per_round_with_scoring=times[-1], first_round_s=times[0], trees_scored=len(trees),
                   best_loglik=float(best_ll),
                   scoring="GTR+I+G by maximum likelihood on the most sampled topology (2 rounds of bounded Brent searches), "
                           "3 Gauss-Seidel sweeps of Newton branch lengths per tree, fp64 pruning (DESIGN.md 11)")
        return out
    except Exception as e:                                    # pragma: no cover
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="MSAs per GPU per step")
    ap.add_argument("--taxa", type=int, default=50)
    ap.add_argument("--sites", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle verification of a sample of the batch")
    ap.add_argument("--no-compat", action="store_true", help="skip the API-compatible (per-step) path figure")
    ap.add_argument("--no-single-msa", action="store_true",
                    help="skip the Batch=1 latency figure (keeps a rocprofv3 summary of this command to the timed workload)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP events pass")
    ap.add_argument("--data-rank", type=int, default=None,
                    help="single-process runs: generate the data shard of THIS rank of a multi-rank run (seed 1000 + rank)")
    ap.add_argument("--dump-merges", default=None,
                    help="rank 0 writes the gathered merge lists of the last timed step, int32 [world, B, T-1, 2], to this .npy")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("NNJ_BENCH_STREAMS", "2")),
                    help="sub-batches of a rollout that run on streams of their own (nnj_set_concurrency); 1 = one stream")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # (a bare `python bench.py --gpus N` never gets here: _self_launch() started N ranks under torch.distributed.run)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the run asked for")
    # Rehearsal knobs (one-GPU box): NNJ_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # NNJ_BENCH_BACKEND=gloo runs the (tiny) collectives on CPU tensors.  The real multi-GPU run uses
    # one GPU per rank and RCCL ("nccl").
    share = os.environ.get("NNJ_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("NNJ_BENCH_BACKEND", "nccl")
    dev_index = 0 if share else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} (local rank {local_rank}) has no GPU of its own: "
                         f"{torch.cuda.device_count()} visible")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where collective buffers live
    dist = None
    # NNJ_BENCH_FORCE_DIST=1: run the collectives at world = 1 too (the RCCL branch on a one-GPU box: process group on
    # the device, weight broadcast, all_gather, all_reduce MIN / MAX and the barriers all execute on GPU tensors)
    force_dist = os.environ.get("NNJ_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:
            import socket
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            s_.close()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
    # weights: rank 0's copy is broadcast once over RCCL so every rank provably runs the same model
    wt = torch.from_numpy(packed).to(cdev)
    if dist is not None:
        dist.broadcast(wt, src=0)
    g = Nnj(cfgs, dev)
    g.load_weights(wt.cpu().numpy())
    g.set_concurrency(args.streams)

    B, T, L = args.batch, args.taxa, args.sites
    data_rank = rank if args.data_rank is None else args.data_rank
    codes = torch.from_numpy(synth.synth_codes(B, T, L, seed=1000 + data_rank, gap_frac=0.2)).to(dev)
    mask = torch.zeros((B, L), dtype=torch.uint8, device=dev)
    g.workspace(B, T, L)

    def step():
        r = g.rollout_argmax(codes, mask)
        return r["merges"].cpu()          # merge lists to the host (ends the step)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    # ---- the timed region: K steps, NO per-kernel events (they are collected in a separate pass)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        merges = step()
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    g.check_numeric()                       # outside the timed region: no score table of the run was non-finite
    # ---- separate profiled pass: every launch bracketed by HIP events on its stream (same work, same inputs)
    prof, prof_steps = {}, 0
    if not args.no_profile:
        prof_steps = min(args.steps, 3)
        g.set_concurrency(1)          # whole-batch launches on one stream: a launch's duration is its own, not a share of the chip
        g.profile_enable(True)
        for _ in range(prof_steps):
            m2 = step()
        torch.cuda.synchronize(dev)
        prof = g.profile_read()
        g.profile_enable(False)
        g.set_concurrency(args.streams)
        # (sub-batches run other launch geometries than the whole batch: partial sums meet in another order, so a
        # near-tie may flip between the two passes; the verification below is of the TIMED pass)
        same_as_timed = bool(torch.equal(m2, merges))
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # merge lists of all shards gathered on every rank (the only result exchange of the path)
        gathered = [torch.empty_like(merges, device=cdev) for _ in range(world)]
        dist.all_gather(gathered, merges.to(cdev))
        assert all(tuple(x.shape) == (B, T - 1, 2) for x in gathered)
        # per-rank throughput and the device each rank really ran on (distinct ordinals = one GPU per rank)
        info = torch.tensor([B * args.steps / elapsed_local, float(torch.cuda.current_device())],
                            dtype=torch.float64, device=cdev)
        infos = [torch.empty_like(info) for _ in range(world)]
        dist.all_gather(infos, info)
        per_rank = {"trees_per_sec": [float(x[0]) for x in infos], "device_ordinal": [int(x[1]) for x in infos],
                    "distinct_devices": len({int(x[1]) for x in infos}), "backend": backend,
                    "collectives_on": "device tensors over RCCL" if backend == "nccl" else "host tensors (rehearsal)"}
    # ---- every rank verifies >= 8 of ITS trees against the oracle, outside the timed region; pass flags are reduced
    verified = None
    if not args.no_verify:
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        verified = verify_sample(g, cfgs, packed, codes, merges, T, L, k=8, threads=max(1, min(16, cores // world)))
        if dist is not None:
            okf = torch.tensor([1.0 if verified["ok"] else 0.0], dtype=torch.float64, device=cdev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            errs = torch.tensor([verified["score_err_rel_vs_fp32_oracle"], verified["score_err_rel_vs_fp64"]],
                                dtype=torch.float64, device=cdev)
            dist.all_reduce(errs, op=dist.ReduceOp.MAX)
            hardf = torch.tensor([0.0 if verified["rf_gate_ok"] and verified["same_merges_as_timed_run"] else 1.0],
                                 dtype=torch.float64, device=cdev)
            dist.all_reduce(hardf, op=dist.ReduceOp.MAX)
            verified["all_ranks_rf_gate_ok"] = bool(hardf.item() == 0.0)
            verified["all_ranks_ok"] = bool(okf.item() == 1.0)
            verified["worst_rank_score_err_rel_vs_fp32_oracle"] = float(errs[0])
            verified["worst_rank_score_err_rel_vs_fp64"] = float(errs[1])
            verified["ranks_verified"] = world
    if args.dump_merges and rank == 0:
        allm = torch.stack([x.cpu() for x in gathered]) if dist is not None else merges[None]
        np.save(args.dump_merges, allm.numpy().astype(np.int32))
    trees = world * B * args.steps
    out = {
        "metric": "trees/sec (Argmax) on 50-taxa x 1024-site MSAs",
        "value": trees / elapsed, "unit": "trees/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "streams_per_gpu": args.streams,
        "precision": "fp32 results (GEMM operands split into two fp16 pieces on the fp16 matrix pipe, three piece "
                     "products per fp32 product, fp32 accumulation; score tables as close to an fp64 evaluation as "
                     "the reference's own fp32 tables: profiles/r02/parity_margin.json, profiles/r03/golden_noise.json)",
        "config": {"workload": f"Batch={B} synthetic {T}x{L} MSAs per GPU, Argmax rollout (BASELINE configs[2]; "
                               f"configs[3] when sharded over 8 GPUs)",
                   "batch_per_gpu": B, "taxa": T, "sites": L, "gap_frac": 0.2,
                   "hyperparameters": "embed_dim 64, 8 heads, 6 layers, patch 1 (the reference's shipped configuration)",
                   "weights": "seeded random (no checkpoint ships with the reference)"},
    }
    if per_rank is not None:
        out["per_rank"] = per_rank
    if rank == 0:
        roof = None
        if prof:
            models = kernel_models(B, T, L, int(cfgs.model.num_enc_layers))
            name, (ms, cnt) = max(prof.items(), key=lambda kv: kv[1][0])
            total_ms = sum(v[0] for v in prof.values())
            tpath = os.path.join(REPO, "profiles", "traffic.json")
            tfile = json.load(open(tpath)) if os.path.exists(tpath) and (B, T, L) == (256, 50, 1024) else {}

            def measured_traffic(kind):
                # HBM bytes per launch from the committed rocprofv3 --pmc passes of this round's build (profiles/traffic.json
                # names the passes); PMC counters cannot be collected from inside the process, so a figure is shown only
                # when that file was made from the same source hash as the library running now
                from neuralnj_amd import build as nbuild
                if not tfile:
                    return "not measured for this workload (profiles/traffic.json is the 256 x 50 x 1024 batch)"
                if tfile.get("source_hash") != nbuild.source_hash():
                    return "stale: the library changed since profiles/traffic.json was measured (tools/profile_round.sh re-measures)"
                return tfile.get("per_kind", {}).get(kind, {}).get("hbm_bytes_per_launch")
            if name in models and cnt > 0:
                avg_s = ms / cnt / 1e3
                ach = models[name]["flops"] / avg_s / 1e12
                peak = models[name]["peak"]
                roof = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak,
                        "unit": "TFLOP/s", "frac": ach / peak, "traffic": measured_traffic(name),
                        "peak_basis": ("dense fp16 MFMA peak / 3 (fp32 GEMM as three fp16 piece products, fp32 accumulate)"
                                       if peak == PEAK_F32_VIA_F16X3_TFLOPS else "dense fp32 MFMA peak"),
                        # the same kernel against the HBM roofline, by its algorithmic bytes
                        "hbm": {"achieved": models[name]["bytes"] / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": models[name]["bytes"] / avg_s / 1e9 / PEAK_HBM_GBS},
                        "avg_launch_ms": ms / cnt, "launches": cnt, "profiled_steps": prof_steps,
                        "share_of_kernel_time": ms / total_ms if total_ms else None,
                        "algorithmic_flops_per_launch": models[name]["flops"],
                        "algorithmic_bytes_per_launch": models[name]["bytes"]}
            out["kernel_ms_per_step"] = {k: round(v[0] / prof_steps, 3) for k, v in prof.items() if v[1]}
            out["kernel_events"] = ("separate pass of %d rollouts after the timed region, same inputs, ONE stream (whole-batch "
                                    "launches); merge lists equal to the timed pass: %s" % (prof_steps, same_as_timed))
            if roof is not None:
                # the BASELINE "NJ Q-matrix kernel" quantity: all per-step kernels of the NJ loop against
                # the HBM roofline, bytes = sum over steps of (n+1) rows + the score tables = 0.334 GB per 50x1024 tree
                step_kinds = NJ_STEP_KINDS
                step_ms = sum(prof[k][0] for k in step_kinds if k in prof) / prof_steps
                C, D = L, 64
                P = lambda n: n * (n - 1) // 2  # noqa: E731
                step_bytes = B * sum((n + 1) * C * D * 4 + 4 * (P(n + 1) + n + P(n)) for n in range(T - 1, 1, -1))
                if step_ms > 0:
                    gbs = step_bytes / (step_ms / 1e3) / 1e9
                    meas = [measured_traffic(k) for k in step_kinds if k in prof and prof[k][1]]
                    meas_total = (sum(mt * prof[k][1] / prof_steps for k, mt in
                                      zip([k for k in step_kinds if k in prof and prof[k][1]], meas))
                                  if meas and all(isinstance(x, (int, float)) for x in meas) else None)
                    # What is REACHABLE: the loop carries sum_n [n C (8 D^2 + 4 n D) + 2 C D^2] flops of
                    # contraction per tree beside its bytes -- at the f16x3 matrix peak that alone takes mfma_floor_ms; and
                    # its pair chains (gate, mix, GELU, three operand splits: ~360 vector + 96 transcendental + ~90 MFMA
                    # issue slots per 16-pair tile and site and pass) bound it at vector_issue_floor_ms on
                    # 1024 SIMDs at 2.4 GHz with no padding and perfect overlap.  The 0.40 target is below both floors.
                    loop_flops = B * sum(n * C * (8.0 * D * D + 4.0 * n * D) + 2.0 * C * D * D for n in range(T - 1, 1, -1))
                    mfma_floor_ms = loop_flops / (PEAK_F32_VIA_F16X3_TFLOPS * 1e12) * 1e3
                    tile_sites = B * C * sum(n - 1 for n in range(T - 1, 1, -1)) / 16.0
                    issue_cycles = 360 * 4 + 96 * 8 + 90 * 8
                    vec_floor_ms = 2 * tile_sites * issue_cycles / (1024 * 2.4e9) * 1e3
                    reachable = {"loop_flops_per_rollout": loop_flops, "mfma_floor_ms": mfma_floor_ms,
                                 "frac_at_mfma_floor": step_bytes / (mfma_floor_ms / 1e3) / 1e9 / PEAK_HBM_GBS,
                                 "vector_issue_floor_ms": vec_floor_ms,
                                 "frac_at_vector_issue_floor": step_bytes / (vec_floor_ms / 1e3) / 1e9 / PEAK_HBM_GBS,
                                 "note": "the BASELINE target of 0.40 of HBM would need the loop in %.1f ms; the matrix pipe "
                                         "alone needs %.1f ms, the vector issue of the pair chains about %.0f ms"
                                         % (step_bytes / (0.40 * PEAK_HBM_GBS * 1e9) * 1e3, mfma_floor_ms, vec_floor_ms)}
                    roof["nj_loop_hbm"] = {"kernels": [k for k in step_kinds if k in prof and prof[k][1]],
                                           "reachable": reachable,
                                           "ms_per_rollout": step_ms,
                                           "algorithmic_bytes_per_rollout": step_bytes, "achieved": gbs,
                                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                           "measured_bytes_per_rollout": meas_total,
                                           "measured_over_algorithmic": (meas_total / step_bytes) if meas_total else None}
                # the memory-only kernel of the step on its own: table assemble + argmax + state/live
                # update; bytes = 4*(P(n+1) + n + 2 P(n)) per step and alignment (read old table + new scores, write the
                # table, read it back for the runner-up)
                if "k_assemble_argmax" in prof and prof["k_assemble_argmax"][1]:
                    a_ms, a_cnt = prof["k_assemble_argmax"]
                    a_bytes = B * sum(4 * (P(n + 1) + n + 2 * P(n)) for n in range(T, 1, -1)) / max(1, T - 1)
                    a_s = a_ms / a_cnt / 1e3
                    roof["assemble_hbm"] = {"kernel": "k_assemble_argmax", "avg_launch_ms": a_ms / a_cnt, "launches": a_cnt,
                                            "algorithmic_bytes_per_launch": a_bytes, "achieved": a_bytes / a_s / 1e9,
                                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a_bytes / a_s / 1e9 / PEAK_HBM_GBS,
                                            "note": "1.5 MB per launch at B=256: a launch-latency-sized kernel, far too "
                                                    "small to approach the HBM roofline (8 TB/s x 5 us = 40 MB)"}
        out["roofline"] = roof
        if world == 1 and not args.no_single_msa:
            # BASELINE configs[1] beside the batched figure: ONE 50 x 1024 alignment per rollout (latency bound)
            one = codes[:1].contiguous()
            g.profile_enable(False)
            r1 = None
            with torch.cuda.stream(torch.cuda.Stream(dev)):          # a stream of its own: the launches of a repeated call
                for _ in range(4):                                   # are captured once and replayed as a hipGraph
                    r1 = g.rollout_argmax(one, None, out=r1)
                    r1["merges"].cpu()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                reps = 20
                for _ in range(reps):
                    r1 = g.rollout_argmax(one, None, out=r1)
                    m1 = r1["merges"].cpu()
                torch.cuda.synchronize(dev)
            ms1 = 1e3 * (time.perf_counter() - t1) / reps
            out["single_msa"] = {"workload": f"Batch=1, {T}x{L} (BASELINE configs[1])", "ms_per_tree": ms1,
                                 "trees_per_sec": 1e3 / ms1,
                                 "same_tree_as_in_the_batch": bool(torch.equal(m1[0], merges[0])),
                                 "launches": "hipGraph replay of the rollout's launches (captured on the second call)"}
        if world == 1 and not args.no_compat:
            cp, m_api = compat_path(cfgs, packed, codes, T, L, dev)
            cp["merges_equal_fused_path_on_all_trees"] = bool(np.array_equal(m_api, merges.numpy()))
            out["compat_path"] = cp
            g.set_concurrency(1)                      # the step API works on the whole batch: compare like with like
            whole = g.rollout_argmax(codes, mask)["merges"].cpu()
            out["step_api_path"] = step_api_path(g, codes, mask, T, whole)
            g.set_concurrency(args.streams)
        if world == 1 and not args.no_compat:
            out["finetune_episode"] = finetune_episode(cfgs, T, L, dev)
            out["search_200x4096"] = search_config5(cfgs, packed, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfgs, packed, T, L)
        if verified is not None:
            out["verified"] = verified
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    # exit code (ADVICE r3 / VERDICT r3 item 5): the run fails -- exit 3 -- exactly when the JSON says ok = false: a merge list
    # that fails the RF gate, a sampled tree that differs from the timed run, or a scale-relative score error against the
    # fp64 evaluation beyond 1x the tolerance, on any rank.  Headroom: the run is deterministic (seeded data,
    # fixed launch geometry, no float atomics), so the gate cannot flake from run to run; what it can do is fail after a
    # kernel change that moves the rounding sequence.  The 8 sampled trees of every rank seed 1000..1007 sit at 4.5e-5 ..
    # 9.6e-5 (profiles/r05/e64_scan.txt; the plain-fp32 oracle, i.e. the reference's arithmetic, at 2.2e-5 .. 7.3e-5 on
    # the same trees): re-run tools/e64_scan.py after every change of a kernel on the <= 64-row path.
    if verified is not None:
        hard = not verified["ok"]
        if dist is not None:
            hard = hard or not verified.get("all_ranks_ok", True)
        if hard:
            sys.exit(3)
    # one GPU per rank: an RCCL run whose ranks did not sit on `world` distinct devices is not the
    # run that was asked for (the shared-GPU rehearsal says so in its environment and is exempt)
    if per_rank is not None and backend == "nccl" and not share and per_rank["distinct_devices"] != world:
        print(f"bench.py: {world} ranks ran on {per_rank['distinct_devices']} distinct device(s): {per_rank['device_ordinal']}",
              file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()
