#!/usr/bin/env python3
"""Golden GRADIENTS of NeuralNJ's Finetune loss from the REFERENCE itself.

Runs only in the build container (needs /root/reference).  Imports the reference's own model.py / environment.py /
utils.py (same placeholder modules as gen_golden.py) and drives them exactly as reinforce_rollout does with
eval=False (reference finetune_rl_search.py:78-189): encode_zxr, then per step get_score_indices_to_prev ->
decode_zxr -> log_softmax(logits / T) -> env.step(actions, agent=agent), all with gradients; then the loss of
RL_finetuning (finetune_rl_search.py:292-307)
    policy_loss = (-(sum_t log p_t[a_t]) * (score - baseline)).mean(),   entropy_reg = -sum_t mean_b H(p_t)
    loss = policy_loss + entropy_reg * strength
and loss.backward().  The actions are FORCED (a fixed merge list stored with the fixture): the reference samples them,
and its RNG stream is not part of the contract.  The tree scores (raxml-ng in the reference) are inputs too.

Stored: inputs (codes, mask, merges, scores, baseline, temperature, strength), weight seed / style, the loss, the per-step
tables, and d loss / d parameter for all 172 tensors, flat in state_dict order.  Nothing of the reference is copied.

Usage: python tests/golden/gen_golden_grad.py      Output: tests/golden/grad_<case>.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def run_case(name, codes, mask, merges, wseed, style, layers, ref, tree_scores, baseline=0.25, temperature=1.0,
             strength=0.01, train=False, dim=64, heads=8, patch=1):
    import torch
    import torch.nn.functional as F
    from neuralnj_amd import synth, weights

    frs, utils_mod, PGPI, PhyInferEnv = ref
    cfgs = gg.make_cfg(utils_mod, layers)
    cfgs.model.embed_dim, cfgs.model.num_enc_heads, cfgs.model.patch_size = dim, heads, patch
    agent = PGPI(cfgs)
    st = weights.seeded_state(cfgs, wseed, style)
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    agent.eval()                                   # the mode the reference's Finetune loop is in (see train_model.py)
    masks, real_dropout = [], F.dropout
    if train:
        # train() mode (reference train.py:435): nn.Dropout draws from torch's generator, whose stream is not part of
        # the contract -- so the masks are drawn HERE (seeded) and recorded, by standing in for the one function
        # nn.Dropout.forward calls; the reference's modules decide where, in which order and with which p it is called.
        agent.train()
        gen = torch.Generator().manual_seed(20240)

        def recorded(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            m = torch.rand(x.shape, generator=gen) >= p
            masks.append((m.numpy().reshape(-1).copy(), float(p)))
            return x * m.to(x.dtype) / (1.0 - p)
        F.dropout = recorded
    B, T, L = codes.shape
    onehot = torch.from_numpy(synth.codes_to_onehot(codes))
    seqs = [synth.codes_to_seqs(codes[b]) for b in range(B)]
    keys = [[f"taxon{i + 1}" for i in range(T)] for _ in range(B)]
    seq_mask = torch.from_numpy(mask)
    env = PhyInferEnv(cfgs, torch.device("cpu"))
    env.init_states(seqs, keys, onehot)
    env.state_tensor = agent.encode_zxr(env.init_state_tensor, seq_mask)
    enc = env.state_tensor.detach().numpy().copy()
    ij_prev, logits_prev = None, None
    selected, log_ps, tables = [], [], []
    step = 0
    while True:
        B_, n = env.state_tensor.shape[:2]
        idx = None
        if ij_prev is not None:
            idx = torch.from_numpy(np.array(utils_mod.get_score_indices_to_prev(ij_prev, env, n, B_)))
        logits = agent.decode_zxr(env.state_tensor, seq_mask, (ij_prev, idx, logits_prev))["logits"]
        tables.append(logits.detach().numpy().copy())
        log_p = torch.log_softmax(logits / temperature, dim=-1)
        ij = [tuple(int(v) for v in merges[b, step]) for b in range(B)]
        actions = torch.tensor([env.action_indices_dict[n][p] for p in ij])
        ij_prev = torch.tensor(ij, dtype=torch.int32)
        done = env.step(actions, [(None, None)] * B, branch_optimize=False, agent=agent)
        if done:
            break
        step += 1
        selected.append(torch.gather(log_p, 1, actions.unsqueeze(1)))
        log_ps.append(log_p)
        logits_prev = logits
    selected = torch.cat(selected, dim=1)
    scores = torch.from_numpy(tree_scores.astype(np.float32))
    policy_loss = (-(selected.sum(dim=1)) * (scores - baseline)).mean()
    entropy_reg = -sum([-torch.sum(torch.exp(lp) * lp, dim=1).mean() for lp in log_ps])
    loss = policy_loss + entropy_reg * strength
    agent.zero_grad()
    loss.backward()
    F.dropout = real_dropout
    grads = np.concatenate([p.grad.detach().numpy().reshape(-1) for p in agent.state_dict(keep_vars=True).values()])
    out = os.path.join(HERE, f"grad_{name}.npz")
    extra = {}
    if train:
        assert masks and len({p for _, p in masks}) == 1
        extra = dict(drop_bits=np.packbits(np.concatenate([m for m, _ in masks])),
                     drop_sizes=np.array([m.size for m, _ in masks], dtype=np.int64), drop_p=np.float32(masks[0][1]))
    np.savez_compressed(out, codes=codes, mask=mask, merges=merges, tree_scores=tree_scores.astype(np.float32),
                        baseline=np.float32(baseline), temperature=np.float32(temperature), strength=np.float32(strength),
                        wseed=np.int64(wseed), style=np.array(style), layers=np.int64(layers), loss=np.float32(loss.item()),
                        dim=np.int64(dim), heads=np.int64(heads), patch=np.int64(patch),
                        policy_loss=np.float32(policy_loss.item()), entropy_reg=np.float32(entropy_reg.item()),
                        grads=grads.astype(np.float32), enc=enc[:, :, ::8].astype(np.float32),
                        tables=np.concatenate([t.reshape(B, -1) for t in tables], axis=1).astype(np.float32),
                        selected=selected.detach().numpy().astype(np.float32), **extra)
    print(f"{name}: loss {loss.item():.6f}  |grad| max {np.abs(grads).max():.3e}  {grads.size} values -> {out}")


def main():
    from neuralnj_amd import synth
    gg._install_placeholders()
    import environment as ref_env      # noqa: E402  (the reference's)
    import model as ref_model          # noqa: E402
    import utils as ref_utils          # noqa: E402
    ref = (None, ref_utils, ref_model.PhyloATTN, ref_env.PhyInferEnv)
    # merges: the reference's own Argmax trees of the matching forward fixtures (any valid merge list would do)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, src, layers, pad in (("b2_t8_l128_s0", "synth_b2_t8_l128_s0", 6, 0), ("b2_t6_l48_pad", None, 2, 5),
                                   ("b1_t20_l256_s1", "synth_b1_t20_l256_s1", 6, 0),
                                   ("b1_t50_l1024_s0", "synth_b1_t50_l1024_s0", 6, 0),       # the bench shape (minutes of CPU)
                                   ("train_b2_t6_l48_pad", None, 2, 5),                      # train() mode: dropout 0.4
                                   # the reference's DEFAULT model (utils.py:45-52): 32 features, 4 heads, 3 layers, patch 4
                                   ("dim32_b2_t6_l48_pad", None, 3, 8)):
        if only and name != only:
            continue
        if src is not None:
            z = np.load(os.path.join(HERE, src + ".npz"), allow_pickle=True)
            codes, mask, merges, wseed, style = z["codes"], z["mask"], z["merges"], int(z["wseed"]), str(z["style"])
        else:
            codes = synth.synth_codes_tree(2, 6, 48, seed=21)
            mask = np.zeros((2, 48), dtype=bool)
            mask[0, -pad:] = True
            codes[0, :, -pad:] = 5
            rng = np.random.default_rng(3)
            merges = np.zeros((2, 5, 2), dtype=np.int32)
            for b in range(2):
                for s, n in enumerate(range(6, 1, -1)):
                    i, j = sorted(rng.choice(n, size=2, replace=False))
                    merges[b, s] = (i, j)
            wseed, style = 17, "plain"
        scores = np.array([0.8, -0.4], dtype=np.float32)[:codes.shape[0]]
        narrow = dict(dim=32, heads=4, patch=4) if name.startswith("dim32_") else {}
        run_case(name, codes, mask, merges, wseed, style, layers, ref, scores, train=name.startswith("train_"), **narrow)


if __name__ == "__main__":
    main()
