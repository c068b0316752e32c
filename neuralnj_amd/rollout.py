"""Argmax inference drivers (counterparts of reference finetune_rl_search.py:78-189,
430-509).  `reinforce_rollout_argmax` repeats the reference's call sequence step by step
through the model/environment API; `argmax_rollout` is the device-resident fast path
(one nnj_rollout_argmax call, merge list replayed on the host trees)."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import utils
from .environment import PhyInferEnv
from .model import PhyloATTN


def reinforce_rollout_argmax(batch, agent, env, device=None):
    """eval=True, argmax=True, branch_optimize=False branch of the reference's
    reinforce_rollout, call for call.  Returns (scores, best_tree, merges)."""
    device = device or next(agent.parameters()).device
    arr = batch["data"].to(device)
    mask = batch["seq_weights"].to(device) == 0
    agent.eval()
    ij_prev, logits_prev, merges = None, None, []
    with torch.no_grad():
        # the encoder is queued BEFORE the host builds the leaf trees (the reference calls env.init_states first,
        # finetune_rl_search.py:108-112; the two are independent): 12,800 Python objects at a batch of 256 are built
        # while the device encodes
        state0 = agent.encode_zxr(arr, mask)
        env.init_states(batch["seqs"], batch["seq_keys"], arr)
        env.state_tensor = state0
        while True:
            B, n = env.state_tensor.shape[:2]
            # The reference computes the old->new table map on the host here (utils.get_score_indices_to_prev,
            # finetune_rl_search.py:121-123) and hands it to decode_zxr.  This package's decode_zxr derives the same map
            # on the device from ij_prev (nnj_score_index_map) and ignores the argument, so the host copy -- 2 ms and a
            # blocking 2.5 MB upload per step at a batch of 256 -- is not built; utils.get_score_indices_to_prev stays
            # available for callers that want it (tests/test_host.py pins it to the reference's).
            logits = agent.decode_zxr(env.state_tensor, mask, (ij_prev, None, logits_prev))["logits"]
            actions = torch.argmax(logits, dim=-1)
            # ONE device->host copy per step for the whole batch (the reference reads the actions element by element,
            # finetune_rl_search.py:159-160: 256 synchronising .item() calls per step were 64 % of this loop's host time)
            pairs = env.tree_pairs_dict[n]
            ij = [pairs[a] for a in actions.tolist()]
            merges.append(ij)
            ij_prev = utils.upload(ij, torch.int32, device)
            if env.step(actions, [(None, None)] * B, branch_optimize=False, agent=agent):
                break
            logits_prev = logits
    scores, _, _, best = env.evaluate_loglikelihood()
    return scores, best, np.array(merges, dtype=np.int32).transpose(1, 0, 2)


def reinforce_rollout_reference_pattern(batch, agent, env, device=None):
    """The same eval + argmax rollout driven the way the reference's UNMODIFIED loop drives it
    (finetune_rl_search.py:100-189) -- for the figure bench.py reports beside `reinforce_rollout_argmax`: the host trees
    first, then the encoder (:108-112); the old->new table map built on the host and uploaded every step (:121-123); a
    log-softmax of every table (:140); the actions read back one `.item()` per alignment (:159-160); the merged pairs
    uploaded from a numpy array (:160).  Nothing of the device path changes -- decode_zxr / env.step are the same calls --
    only the host pattern around them, which is what a drop-in user who edits nothing gets."""
    device = device or next(agent.parameters()).device
    arr = batch["data"].to(device)
    mask = batch["seq_weights"].to(device) == 0
    env.init_states(batch["seqs"], batch["seq_keys"], arr)
    agent.eval()
    ij_prev, logits_prev, merges = None, None, []
    with torch.no_grad():
        env.state_tensor = agent.encode_zxr(env.init_state_tensor, mask)
        while True:
            B, n = env.state_tensor.shape[:2]
            idx = None
            if ij_prev is not None:
                idx = torch.from_numpy(np.array(utils.get_score_indices_to_prev(ij_prev, env, n, B))).to(device)
            logits = agent.decode_zxr(env.state_tensor, mask, (ij_prev, idx, logits_prev))["logits"]
            torch.log_softmax(logits, dim=-1)
            actions = torch.argmax(logits, dim=-1)
            ij = [env.tree_pairs_dict[n][a.item()] for a in actions]
            merges.append(ij)
            ij_prev = torch.from_numpy(np.array(ij)).to(device).to(torch.int32)
            if env.step(actions, [(None, None)] * B, branch_optimize=False, agent=agent):
                break
            logits_prev = logits
    scores, _, _, best = env.evaluate_loglikelihood()
    return scores, best, np.array(merges, dtype=np.int32).transpose(1, 0, 2)


def argmax_rollout(batch, agent, env, device=None):
    """Fast path: whole rollout on the device, no per-step host round trip."""
    device = device or next(agent.parameters()).device
    codes = batch["codes"] if "codes" in batch else agent.onehot_to_codes(batch["data"].to(device))
    mask = batch["seq_weights"].to(device) == 0
    env.init_states(batch["seqs"], batch["seq_keys"], batch.get("data"))
    r = agent.rollout_argmax(codes.to(device), mask)
    merges = r["merges"].cpu().numpy()
    agent._context().check_numeric()          # raises if a score table was not finite (include/nnj.h)
    env.apply_merges(merges)
    scores, _, _, best = env.evaluate_loglikelihood()
    return scores, best, merges


def sample_rollouts(batch, agent, env, n_rollouts, seed=0, temperature=1.0, device=None):
    """Device part of the reference's RL_Search / "NeuralNJ-MC" (finetune_rl_search.py:338-427):
    `n_rollouts` sampled rollouts of ONE alignment, encoded once.  Returns the distinct sampled trees
    as Newick strings with their multiplicities.  Ranking them by likelihood (raxml-ng in the
    reference) is outside this package."""
    device = device or next(agent.parameters()).device
    codes = batch["codes"] if "codes" in batch else agent.onehot_to_codes(batch["data"].to(device))
    mask = batch["seq_weights"].to(device) == 0
    T = codes.shape[1]
    u = torch.from_numpy(np.random.default_rng(seed).random((n_rollouts, T - 1)).astype(np.float32))
    r = agent._context().rollout_sample(codes[:1].to(device), mask[:1], u, temperature=temperature,
                                        replicas=n_rollouts)
    # duplicate filter on the device (reference utils.py:76 keeps one tree per topo_repr): 64-bit topology keys,
    # only the first rollout of every distinct key is replayed into a host tree
    keys = agent._context().topology_hash(r["merges"])
    uniq, inverse, counts = torch.unique(keys, return_inverse=True, return_counts=True)
    first = torch.full((uniq.numel(),), n_rollouts, dtype=torch.long, device=keys.device)
    first.scatter_reduce_(0, inverse, torch.arange(n_rollouts, device=keys.device), reduce="amin")
    order = torch.argsort(first)                          # distinct trees in order of first appearance
    first, counts = first[order].cpu().numpy(), counts[order].cpu().numpy()
    merges = r["merges"].cpu().numpy()
    agent._context().check_numeric()
    k = len(first)
    env.init_states([batch["seqs"][0]] * k, [batch["seq_keys"][0]] * k, None)
    env.apply_merges(merges[first])
    trees = [(st.subtrees[0].utree_op_str, int(c)) for st, c in zip(env.states, counts)]
    return trees, merges


def search_rollouts(batch, agent, env, n_rollouts, seed=0, temperature=1.0, model=None, sweeps=3, device=None,
                    details=None):
    """One round of the reference's RL_Search / "NeuralNJ-MC" (finetune_rl_search.py:338-427) on the GPU: sample
    `n_rollouts` trees of ONE alignment (encoded once), drop duplicate topologies (device keys), optimise the branch
    lengths of the distinct trees and score them by log-likelihood under GTR+I+G (neuralnj_amd.likelihood, where the
    reference calls raxml-ng), keep the best.  Returns (best Newick with optimised lengths, its log-likelihood,
    [(Newick, log-likelihood, multiplicity)] of the distinct trees, best first)."""
    from . import likelihood as lk
    device = device or next(agent.parameters()).device
    ctx = agent._context()
    codes = batch["codes"] if "codes" in batch else agent.onehot_to_codes(batch["data"].to(device))
    codes = codes[:1].to(device)
    mask = (batch["seq_weights"].to(device) == 0)[:1]
    T = codes.shape[1]
    u = torch.from_numpy(np.random.default_rng(seed).random((n_rollouts, T - 1)).astype(np.float32))
    r = ctx.rollout_sample(codes, mask, u, temperature=temperature, replicas=n_rollouts)
    keys = ctx.topology_hash(r["merges"])
    uniq, inverse, counts = torch.unique(keys, return_inverse=True, return_counts=True)
    first = torch.full((uniq.numel(),), n_rollouts, dtype=torch.long, device=keys.device)
    first.scatter_reduce_(0, inverse, torch.arange(n_rollouts, device=keys.device), reduce="amin")
    merges = r["merges"][first]                               # one merge list per distinct topology
    ctx.check_numeric()
    if isinstance(model, str) and model == "auto":
        # the reference's opt_model=True (environment.py:373-377): GTR+I+G parameters by maximum likelihood, here once per
        # round on the most frequently sampled topology and shared by the trees scored together
        model, _, _ = lk.optimize_model(ctx, codes, merges[torch.argmax(counts):][:1], None, None, mask=mask, rounds=2, sweeps=sweeps)
    ll, br = lk.tree_optimize(ctx, codes, merges, None, model, mask=mask, sweeps=sweeps)
    order = torch.argsort(ll, descending=True)
    m_np, br_np, ll_np = merges[order].cpu().numpy(), br[order].cpu().numpy(), ll[order].cpu().numpy()
    cnt = counts[order].cpu().numpy()
    k = len(ll_np)
    env.init_states([batch["seqs"][0]] * k, [batch["seq_keys"][0]] * k, None)
    env.apply_merges(m_np, br_np, ll_np)
    trees = [(st.subtrees[0].utree_op_str, float(s), int(c)) for st, s, c in zip(env.states, ll_np, cnt)]
    if details is not None:
        # what the round scored, best first (tests feed it to the likelihood oracle): merge lists, optimised branch
        # lengths, log-likelihoods, multiplicities and the substitution model they were scored under
        details.update(merges=m_np, brlen=br_np, loglik=ll_np, counts=cnt, model=model, rollouts=int(n_rollouts))
    return trees[0][0], trees[0][1], trees


def argmax_inference(cfgs, test_dir, write_dir, device="cuda", fast=True):
    """Counterpart of the reference's Argmax_inference (finetune_rl_search.py:478-509):
    one tree per *.phy file in test_dir, written as <name>.tre."""
    from .phydata import load_pi_instance
    env = PhyInferEnv(cfgs, device)
    agent = PhyloATTN(cfgs).to(device)
    if cfgs.reload_checkpoint_path:
        ckpt = torch.load(cfgs.reload_checkpoint_path, map_location="cpu")
        agent.load_state_dict(ckpt["model_state_dict"])
    os.makedirs(write_dir, exist_ok=True)
    out = {}
    for fname in sorted(os.listdir(test_dir)):
        if not fname.endswith(".phy"):
            continue
        batch = load_pi_instance(os.path.join(test_dir, fname))
        # expand to env.batch_size replicas like Agmax_one_instance (:435-444)
        bs = int(cfgs.env.batch_size)
        if bs > 1:
            batch = {k: (v * bs if isinstance(v, list) else v.expand(bs, *v.shape[1:])) for k, v in batch.items()}
        fn = argmax_rollout if fast else reinforce_rollout_argmax
        _, best, _ = fn(batch, agent, env, torch.device(device))
        with open(os.path.join(write_dir, fname[:-4] + ".tre"), "w") as f:
            f.write(best)
        out[fname] = best
    return out


def search_inference(cfgs, test_dir, write_dir, stop_step=100, device="cuda", model=None, temperature=1.0):
    """Counterpart of the reference's Search_inference / RL_Search (finetune_rl_search.py:338-427, 512-541): for every
    *.phy file, `stop_step` rounds of env.batch_size sampled rollouts of the alignment (one round = one
    reinforce_rollout(eval=True, branch_optimize=True) of the reference), each round's distinct trees scored on the
    GPU (branch lengths optimised, log-likelihood under GTR+I+G), the best tree over all rounds written as <name>.tre.
    Returns {file: dict(the_best_tree, the_best_score, step_cur, distinct_trees)}."""
    from .phydata import load_pi_instance
    env = PhyInferEnv(cfgs, device)
    agent = PhyloATTN(cfgs).to(device)
    if cfgs.reload_checkpoint_path:
        ckpt = torch.load(cfgs.reload_checkpoint_path, map_location="cpu")
        agent.load_state_dict(ckpt["model_state_dict"])
    os.makedirs(write_dir, exist_ok=True)
    out = {}
    for fname in sorted(os.listdir(test_dir)):
        if not fname.endswith(".phy"):
            continue
        batch = load_pi_instance(os.path.join(test_dir, fname))
        best, best_score, seen = None, -np.inf, set()
        for step in range(1, int(stop_step) + 1):
            tree, score, trees = search_rollouts(batch, agent, env, int(cfgs.env.batch_size), seed=step,
                                                 temperature=temperature, model=model, device=torch.device(device))
            seen.update(t[0] for t in trees)
            if score > best_score:
                best, best_score = tree, score
        with open(os.path.join(write_dir, fname[:-4] + ".tre"), "w") as f:
            f.write(best)
        out[fname] = dict(the_best_tree=best, the_best_score=best_score, step_cur=int(stop_step), distinct_trees=len(seen))
    return out


def reinforce_loss(batch, agent, env, forced_merges, tree_scores, baseline, temperature=1.0, entropy_reg_strength=0.01,
                   device=None, replicas=None):
    """One episode of the reference's RL_finetuning with gradients (finetune_rl_search.py:78-189 with eval=False, and
    the loss of :292-307): the model / environment calls of the reference's loop, differentiable (train_model.py: every
    operator a HIP forward + backward kernel), then
        loss = mean_b(-(sum_t log p_t[a_t]) * (score_b - baseline)) + strength * (-sum_t mean_b H(p_t)).
    `forced_merges` int [B,T-1,2] replaces the reference's Categorical sampling (its RNG stream is the caller's
    business: sample with Nnj.rollout_sample and pass the merges here); `tree_scores` [B] are the rewards (the
    reference's raxml-ng log-likelihoods; likelihood.tree_optimize here).  The few table-sized operations of the
    reference's own driver (log_softmax, gather, the sums of the loss) are torch operations, as they are there.
    `replicas`: True = the batch is B copies of ONE alignment (the caller built it: rl_finetuning), False = it is not,
    None = find out by comparing the rows (2 (B-1) device comparisons, each a host synchronisation).
    Returns (loss, per-step tables)."""
    device = torch.device(device or next(agent.parameters()).device)
    if device.type == "cuda":
        # the operator kernels of libnnj_train_hip.so launch on the CURRENT device (train_ops._st checks it)
        with torch.cuda.device(device):
            return _reinforce_loss(batch, agent, env, forced_merges, tree_scores, baseline, temperature,
                                   entropy_reg_strength, device, replicas)
    return _reinforce_loss(batch, agent, env, forced_merges, tree_scores, baseline, temperature, entropy_reg_strength,
                           device, replicas)


def _reinforce_loss(batch, agent, env, forced_merges, tree_scores, baseline, temperature, entropy_reg_strength, device,
                    replicas):
    arr = batch["data"].to(device)
    mask = batch["seq_weights"].to(device) == 0
    env.init_states(batch["seqs"], batch["seq_keys"], arr)
    merges = np.asarray(forced_merges)
    B = merges.shape[0]
    if replicas is None:
        replicas = B > 1 and all(torch.equal(arr[0], arr[b]) for b in range(1, B)) \
            and all(torch.equal(mask[0], mask[b]) for b in range(1, B))
    if B > 1 and replicas and agent._wants_grad():
        # replicas of ONE alignment (the batch of the reference's Search / Finetune loops): encoded once, the gradient
        # of the shared encoding is the sum over the replicas
        from .train_model import ExpandBatch
        env.state_tensor = ExpandBatch.apply(agent.encode_zxr(env.init_state_tensor[:1], mask[:1]), B)
    else:
        env.state_tensor = agent.encode_zxr(env.init_state_tensor, mask)
    ij_prev, logits_prev, selected, log_ps, tables, step = None, None, [], [], [], 0
    while True:
        n = env.state_tensor.shape[1]
        idx = None
        if ij_prev is not None:
            idx = utils.upload(utils.get_score_indices_to_prev(merges[:, step - 1], env, n, B), torch.int64, device)
        logits = agent.decode_zxr(env.state_tensor, mask, (ij_prev, idx, logits_prev))["logits"]
        tables.append(logits)
        log_p = torch.log_softmax(logits / temperature, dim=-1)
        ij = [tuple(int(v) for v in merges[b, step]) for b in range(B)]
        # the actions are known on the host (forced): the environment gets the host copy (its bookkeeping reads them
        # back), the gather below the device copy -- no step of the loop waits for the device
        actions_host = torch.tensor([env.action_indices_dict[n][p] for p in ij])
        actions = utils.upload(actions_host, torch.int64, device)
        ij_prev = utils.upload(ij, torch.int32, device)
        if env.step(actions_host, [(None, None)] * B, branch_optimize=False, agent=agent):
            break
        step += 1
        selected.append(torch.gather(log_p, 1, actions.unsqueeze(1)))
        log_ps.append(log_p)
        logits_prev = logits
    selected = torch.cat(selected, dim=1)
    scores = torch.as_tensor(tree_scores, dtype=torch.float32, device=device)
    policy_loss = (-(selected.sum(dim=1)) * (scores - baseline)).mean()
    entropy_reg = -sum([-torch.sum(torch.exp(lp) * lp, dim=1).mean() for lp in log_ps])
    agent.__dict__["_train_keys"] = None          # (the key cache of train_model holds the last state: the graph keeps what it needs)
    return policy_loss + entropy_reg * entropy_reg_strength, tables


def episode_rng(seed, rank):
    """Uniform stream of rank `rank`'s sampled episodes (rl_finetuning)."""
    return np.random.default_rng([int(seed), int(rank) + 1, 1])


def baseline_rng(seed):
    """Uniform stream of the first, rank-independent baseline rollout (rl_finetuning): never equal to an episode stream."""
    return np.random.default_rng([int(seed), 0, 2])


def rl_finetuning(cfgs, batch, agent, optimizer, env, stop_step=20, seed=0, model=None, temperature=1.0, device=None,
                  dist=None):
    """Counterpart of the reference's RL_finetuning (finetune_rl_search.py:192-335) for one alignment: per episode one
    sampled rollout of the current policy (nnj_rollout_sample: the fused inference kernels), its tree scored by
    log-likelihood on the GPU (likelihood.tree_optimize where the reference calls raxml-ng), the episode replayed with
    gradients (reinforce_loss), loss.backward(); after `cfgs.num_episodes` episodes the gradients are clipped by value
    and the optimizer steps.  Baseline as in the reference (finetune_rl_search.py:265-279): the first epoch's is the
    score of ONE SAMPLED rollout of the policy (its reinforce_rollout(eval=True) leaves argmax=False), later ones the
    running maximum of the epoch means.  The reference's replay buffer re-injects nothing (its `sample`
    returns None, utils.py:99), so every action here is sampled too; utils.ReplayBuffer keeps the same surface.
    `dist` (an initialised torch.distributed, one process per GPU): the episodes of an epoch are split over the ranks
    (every rank samples with its own stream), the gradients are summed by one all-reduce of one flat bucket per
    optimizer step (sharding.allreduce_gradients) and every rank takes the same step; the best tree is each rank's own.
    Returns dict(the_best_tree, the_best_score, step_cur, losses)."""
    from . import likelihood as lk
    from . import sharding
    device = device or next(agent.parameters()).device
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    ctx = agent._context()
    codes = (batch["codes"] if "codes" in batch else agent.onehot_to_codes(batch["data"].to(device)))[:1].to(device)
    mask = (batch["seq_weights"].to(device) == 0)[:1]
    T = codes.shape[1]
    # Two families of uniform streams with DIFFERENT non-zero tags: SeedSequence zero-pads its entropy, so [seed] and
    # [seed, 0] are the same stream -- rank 0's first episode would replay the baseline rollout (advantage exactly 0, no
    # gradient: ADVICE r4).  Episodes: [seed, rank + 1, 1]; the rank-independent baseline: [seed, 0, 2].
    rng = episode_rng(seed, rank)
    agent.eval()

    def score(merges):
        ll, br = lk.tree_optimize(agent._context(), codes, merges, None, model, mask=mask)
        return ll.to(torch.float32), br

    with torch.no_grad():
        # the first baseline is ONE sampled rollout shared by all episodes of the epoch (finetune_rl_search.py:213-222):
        # drawn from a rank-independent stream so that every rank starts from the same baseline and the all-reduced
        # gradient equals the single-process run's
        u0 = torch.from_numpy(baseline_rng(seed).random((1, T - 1)).astype(np.float32))
        first = ctx.rollout_sample(codes, mask, u0, temperature=temperature, replicas=1)["merges"]
        sc0, br0 = score(first)
        baseline_val = float(sc0[0])
    # the baseline rollout is a scored tree like any other: it seeds the best tree, so a rank that is given no episode
    # (more ranks than episodes) still returns a tree -- the same one on every rank -- instead of None
    env.init_states([batch["seqs"][0]], [batch["seq_keys"][0]], None)
    env.apply_merges(first[:1].cpu().numpy(), br0[:1].cpu().numpy(), sc0[:1].cpu().numpy())
    best_tree, best_score = env.states[0].subtrees[0].utree_op_str, baseline_val
    losses, batch_scores, step_cur = [], [], 0
    lo, hi = sharding.shard_bounds(int(cfgs.num_episodes), world, rank)
    E = hi - lo                              # a rank without an episode samples nothing: it joins the all-reduce with zeros
    reps = {k: (v[:1] * E if isinstance(v, list) else v[:1].expand(E, *v.shape[1:])) for k, v in batch.items()}
    for epoch in range(1, int(cfgs.num_epoch) + 1):
        if epoch > 1 and batch_scores:
            baseline_val = max(baseline_val, sum(batch_scores) / len(batch_scores))
            batch_scores = []
        optimizer.zero_grad()
        # The reference runs the epoch's episodes one after the other against the same weights and baseline and adds
        # their gradients up (loss.backward() per episode, one optimizer step per epoch).  Here they run as ONE batch of
        # E replicas: sampled together, scored together, replayed with gradients together (the alignment encoded
        # once); E x the batch-mean loss is the sum of the E episode losses.
        if E == 0:                           # (allreduce_gradients zero-fills parameters without a gradient)
            step_cur += int(cfgs.num_episodes)
            sharding.allreduce_gradients(agent.parameters(), dist)
            torch.nn.utils.clip_grad_value_(agent.parameters(), clip_value=float(cfgs.clip_value))
            optimizer.step()
            if step_cur >= int(stop_step):
                break
            continue
        with torch.no_grad():
            u = torch.from_numpy(rng.random((E, T - 1)).astype(np.float32))
            merges = agent._context().rollout_sample(codes, mask, u, temperature=temperature, replicas=E)["merges"]
            agent._context().check_numeric()
            sc, br = score(merges)
        loss, _ = reinforce_loss(reps, agent, env, merges.cpu().numpy(), sc, baseline_val, temperature,
                                 float(cfgs.entropy_reg_strength), device, replicas=True)
        (loss * E).backward()
        losses.append(float(loss.detach()))
        batch_scores.extend(float(v) for v in sc)
        k = int(torch.argmax(sc))
        if float(sc[k]) > best_score:
            best_score = float(sc[k])
            env.init_states([batch["seqs"][0]], [batch["seq_keys"][0]], None)
            env.apply_merges(merges[k:k + 1].cpu().numpy(), br[k:k + 1].cpu().numpy(), sc[k:k + 1].cpu().numpy())
            best_tree = env.states[0].subtrees[0].utree_op_str
        step_cur += int(cfgs.num_episodes) if world > 1 else E
        sharding.allreduce_gradients(agent.parameters(), dist)
        torch.nn.utils.clip_grad_value_(agent.parameters(), clip_value=float(cfgs.clip_value))
        optimizer.step()
        if step_cur >= int(stop_step):
            break
    return dict(the_best_tree=best_tree, the_best_score=best_score, step_cur=step_cur, losses=losses,
                baseline=baseline_val)


def finetune_inference(cfgs, test_dir, write_dir, stop_step=20, device="cuda", model=None):
    """Counterpart of the reference's finetune_inference (finetune_rl_search.py:544-577): the checkpoint's policy and
    one Adam optimizer are fine-tuned on every *.phy file in turn (rl_finetuning; as in the reference the network is
    NOT reset between files) and the best tree of each file is written as <name>.tre."""
    from .phydata import load_pi_instance
    os.makedirs(write_dir, exist_ok=True)
    agent = PhyloATTN(cfgs).to(device)
    if cfgs.reload_checkpoint_path:
        ckpt = torch.load(cfgs.reload_checkpoint_path, map_location="cpu")
        agent.load_state_dict(ckpt["model_state_dict"])
    optimizer = torch.optim.Adam(agent.parameters(), lr=float(cfgs.lr))
    env = PhyInferEnv(cfgs, device)
    out = {}
    for fname in sorted(os.listdir(test_dir)):
        if not fname.endswith(".phy"):
            continue
        batch = load_pi_instance(os.path.join(test_dir, fname))
        res = rl_finetuning(cfgs, batch, agent, optimizer, env, stop_step=stop_step, model=model, device=torch.device(device))
        with open(os.path.join(write_dir, fname[:-4] + ".tre"), "w") as f:
            f.write(res["the_best_tree"])
        out[fname] = res
    return out


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="NeuralNJ Argmax inference on MI355X (neuralnj_amd)")
    ap.add_argument("--config_path", type=str, required=True)
    ap.add_argument("--infer_opt", type=str, default="Argmax")
    ap.add_argument("--output", type=str, default=None)
    ap.add_argument("--stop_step", type=int, default=100)
    args = ap.parse_args(argv)
    if args.infer_opt not in ("Argmax", "Search", "Finetune"):
        raise SystemExit("--infer_opt is one of Argmax, Search, Finetune")
    cfgs = utils.empty_config()
    cfgs.merge_from_file(args.config_path)
    base = os.path.dirname(os.path.abspath(args.config_path))
    test_dir = cfgs.instance_path if os.path.isabs(cfgs.instance_path) else os.path.join(base, "..", cfgs.instance_path)
    if cfgs.reload_checkpoint_path and not os.path.isabs(cfgs.reload_checkpoint_path):
        cfgs.reload_checkpoint_path = os.path.join(base, "..", cfgs.reload_checkpoint_path)
    name = os.path.basename(os.path.normpath(test_dir))
    write_dir = args.output or f"output/{args.infer_opt}_dim{cfgs.model.embed_dim}_patch{cfgs.model.patch_size}/{name}"
    if args.infer_opt == "Search":
        search_inference(cfgs, test_dir, write_dir, stop_step=args.stop_step)
    elif args.infer_opt == "Finetune":
        finetune_inference(cfgs, test_dir, write_dir, stop_step=args.stop_step)
    else:
        argmax_inference(cfgs, test_dir, write_dir)


if __name__ == "__main__":
    main()
