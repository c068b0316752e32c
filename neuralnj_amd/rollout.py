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
            actions = torch.argmax(logits, dim=-1)
            ij = [env.tree_pairs_dict[n][a.item()] for a in actions]
            merges.append(ij)
            ij_prev = torch.tensor(ij, dtype=torch.int32, device=device)
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


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="NeuralNJ Argmax inference on MI355X (neuralnj_amd)")
    ap.add_argument("--config_path", type=str, required=True)
    ap.add_argument("--infer_opt", type=str, default="Argmax")
    ap.add_argument("--output", type=str, default=None)
    args = ap.parse_args(argv)
    if args.infer_opt != "Argmax":
        raise SystemExit("only --infer_opt Argmax is implemented (Search / Finetune are outside the hot path)")
    cfgs = utils.empty_config()
    cfgs.merge_from_file(args.config_path)
    base = os.path.dirname(os.path.abspath(args.config_path))
    test_dir = cfgs.instance_path if os.path.isabs(cfgs.instance_path) else os.path.join(base, "..", cfgs.instance_path)
    if cfgs.reload_checkpoint_path and not os.path.isabs(cfgs.reload_checkpoint_path):
        cfgs.reload_checkpoint_path = os.path.join(base, "..", cfgs.reload_checkpoint_path)
    name = os.path.basename(os.path.normpath(test_dir))
    write_dir = args.output or f"output/Argmax_dim{cfgs.model.embed_dim}_patch{cfgs.model.patch_size}/{name}"
    argmax_inference(cfgs, test_dir, write_dir)


if __name__ == "__main__":
    main()
