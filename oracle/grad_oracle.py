"""CPU oracle of the Finetune loss and its gradients.  TEST INFRASTRUCTURE ONLY: imported by tests/ and
never by the product (neuralnj_amd/ does not reference oracle/).

A restatement of the reference's forward pass in plain torch operations on the CPU, in float64 or float32, each function
citing the reference lines it follows; gradients come from torch.autograd on this restatement.  PINNED: against the
golden gradients captured from the reference itself (tests/golden/grad_*.npz, gen_golden_grad.py) in
tests/test_grad_oracle.py -- the float32 build reproduces the reference's gradients to fp32 rounding, the float64 build is
the arbiter that says how far fp32 rounding alone moves a gradient.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def _lin(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _ln(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)    # msa_modules.py:107


def _keep(x, drop, kind):
    """nn.Dropout in train() mode with a GIVEN mask: drop(kind, shape) returns (keep mask, p) -- tests hand the masks the
    device path drew -- or None in eval mode."""
    if drop is None:
        return x
    m, p = drop(kind, tuple(x.shape))
    return x * m.to(x.dtype) / (1.0 - p)


def row_attention(x, sd, pre, pad, drop=None, heads=8):
    """axial_attention.py:66-138 (grad mode: no chunking).  x [R,C,B,D]; pad bool [B,C]."""
    R, C, B, D = x.shape
    H, dh = heads, D // heads
    q = _lin(x, sd, pre + ".q_proj").view(R, C, B, H, dh) * (dh ** -0.5 / math.sqrt(R))        # :31-33, 77
    k = _lin(x, sd, pre + ".k_proj").view(R, C, B, H, dh)
    v = _lin(x, sd, pre + ".v_proj").view(R, C, B, H, dh)
    if pad is not None:
        q = q * (1 - pad.t().to(q.dtype))[None, :, :, None, None]                                # :78-82
    w = torch.einsum("rinhd,rjnhd->hnij", q, k)                                                   # :97
    if pad is not None:
        w = w.masked_fill(pad[None, :, None, :], -10000.0)                                        # :99-103
    p = _keep(w.softmax(-1), drop, "row_probs")                                                   # :135-136
    ctx = torch.einsum("hnij,rjnhd->rinhd", p, v).reshape(R, C, B, D)                             # :114
    return _lin(ctx, sd, pre + ".out_proj")


def column_attention(x, sd, pre, pad, drop=None, heads=8):
    """axial_attention.py:190-240."""
    R, C, B, D = x.shape
    H, dh = heads, D // heads
    q = _lin(x, sd, pre + ".q_proj").view(R, C, B, H, dh) * dh ** -0.5                            # :214
    k = _lin(x, sd, pre + ".k_proj").view(R, C, B, H, dh)
    v = _lin(x, sd, pre + ".v_proj").view(R, C, B, H, dh)
    w = torch.einsum("icnhd,jcnhd->hcnij", q, k)                                                  # :216
    if pad is not None:
        w = w.masked_fill(pad.t()[None, :, :, None, None], -10000.0)                              # :220-224
    p = _keep(w.softmax(-1), drop, "col_probs")                                                   # :232-233
    ctx = torch.einsum("hcnij,jcnhd->icnhd", p, v).reshape(R, C, B, D)                            # :234
    return _lin(ctx, sd, pre + ".out_proj")


def encode(sd, onehot, pad, layers, drop=None, heads=8, patch=1):
    """model.py:67-88 + msa_modules.py:62-125.  drop None: eval mode (dropout = identity); else see _keep.
    pad: the TOKEN mask (the caller takes every patch-th site, model.py:81)."""
    if patch > 1:                                                                                 # model.py:76: 'b r (c k) e -> b r c (k e)'
        B_, R_, L_, V_ = onehot.shape
        onehot = onehot.reshape(B_, R_, L_ // patch, patch * V_)
    x = _lin(F.gelu(_lin(onehot, sd, "embed.0")), sd, "embed.2")                                  # model.py:39-43
    x = x.permute(1, 2, 0, 3)
    for l in range(layers):
        pre = f"seq_emb_layers.{l}."
        y = row_attention(_ln(x, sd, pre + "row_self_attention.layer_norm"), sd, pre + "row_self_attention.layer", pad, drop, heads)
        x = x + _keep(y, drop, "out")                                                              # msa_modules.py:119-120
        y = column_attention(_ln(x, sd, pre + "column_self_attention.layer_norm"), sd, pre + "column_self_attention.layer", pad, drop, heads)
        x = x + _keep(y, drop, "out")
        y = _ln(x, sd, pre + "feed_forward_layer.layer_norm")
        y = _keep(F.gelu(_lin(y, sd, pre + "feed_forward_layer.layer.fc1")), drop, "act")          # msa_modules.py:148-149
        x = x + _keep(_lin(y, sd, pre + "feed_forward_layer.layer.fc2"), drop, "out")
    return x.permute(2, 0, 1, 3)


def aggregate(sd, state, x_i, x_j, i_idx, j_idx, patch_num):
    """model.py:102-155.  state [B,n,C,D]; x_i, x_j [B,p,C,D]; i_idx, j_idx [B,p]."""
    B, n, C, D = state.shape
    z = torch.sigmoid(_lin(x_i - x_j, sd, "h_linear_last"))
    x = z * x_i + (1 - z) * x_j
    if n <= 2:                                                                                    # :111
        return x
    q = _lin(x, sd, "g_attn_q")
    k = _lin(state, sd, "g_attn_k")
    alpha = torch.einsum("bncd,brcd->bnr", q, k) / math.sqrt(D * patch_num)                       # :118
    r = torch.arange(n).view(1, 1, n)
    alpha = alpha.masked_fill((r == i_idx.unsqueeze(-1)) | (r == j_idx.unsqueeze(-1)), float("-inf"))   # :120-144
    alpha = torch.softmax(alpha, dim=-1)
    xg = torch.einsum("bnr,brcd->bncd", alpha, state)                                             # :148
    w = torch.sigmoid(_lin(xg, sd, "g_linear_last"))
    return (1 - w) * x + w * xg                                                                   # :150-153


def decode_gg(sd, state, x_i, x_j, keep, i_idx, j_idx, patch_num):
    """model.py:90-99: masked SUM over the sites of s_out(aggregate(...))."""
    x = aggregate(sd, state, x_i, x_j, i_idx, j_idx, patch_num)
    s = _lin(F.gelu(_lin(x, sd, "s_out.0")), sd, "s_out.2").squeeze(-1)
    return (s * keep[:, None, :]).sum(-1)


def _rows(state, idx):
    return torch.gather(state, 1, idx[:, :, None, None].expand(-1, -1, state.shape[2], state.shape[3]))


def reinforce_loss(sd, onehot, pad, merges, tree_scores, baseline, temperature, strength, layers, dtype=torch.float64,
                   drop=None, heads=8, patch=1):
    """The loop of reinforce_rollout with eval=False (finetune_rl_search.py:78-189) on forced actions and the loss of
    RL_finetuning (:292-307).  sd: {name: tensor requiring grad}.  Returns (loss, tables)."""
    from neuralnj_amd import utils
    onehot = torch.as_tensor(onehot).to(dtype)
    pad = torch.as_tensor(pad).bool()
    B, T, L, _ = onehot.shape
    if patch > 1:
        pad = pad[:, ::patch]                                                                     # model.py:81
        L = -(-L // patch)                                                                        # patch_num (model.py:72)
    state = encode(sd, onehot, pad, layers, drop, heads, patch)
    keep = (~pad).to(dtype)
    merges = np.asarray(merges)
    table, tables, selected, ents = None, [], [], []
    for step, n in enumerate(range(T, 1, -1)):
        if table is None:
            row, col = torch.triu_indices(n, n, offset=1)
            i_idx, j_idx = row[None].expand(B, -1), col[None].expand(B, -1)
            table = decode_gg(sd, state, _rows(state, i_idx), _rows(state, j_idx), keep, i_idx, j_idx, L)
        else:
            ip = torch.as_tensor(merges[:, step - 1, 0]).long()
            r = torch.arange(n)[None].expand(B, n)
            i_idx, j_idx = torch.minimum(ip[:, None], r), torch.maximum(ip[:, None], r)              # model.py:186-190
            new = decode_gg(sd, state, _rows(state, i_idx), _rows(state, j_idx), keep, i_idx, j_idx, L)
            idx = torch.from_numpy(utils.index_map_batch(n, merges[:, step - 1]))                     # utils.py:213-251
            table = torch.gather(torch.cat([table, new], -1), 1, idx)                                 # model.py:199-201
        tables.append(table)
        log_p = torch.log_softmax(table / temperature, dim=-1)
        ij = torch.as_tensor(merges[:, step]).long()
        act = ij[:, 0] * n - ij[:, 0] * (ij[:, 0] + 1) // 2 + (ij[:, 1] - ij[:, 0] - 1)
        if n == 2:
            break                                                                                    # done: nothing appended (:164-167)
        selected.append(log_p.gather(1, act[:, None]))
        ents.append(-(log_p.exp() * log_p).sum(1).mean())
        # env.step (environment.py:760-835)
        i_idx, j_idx = ij[:, 0:1], ij[:, 1:2]
        new_row = aggregate(sd, state, _rows(state, i_idx), _rows(state, j_idx), i_idx, j_idx, L)
        r = torch.arange(n - 1)[None].expand(B, n - 1)
        base = r + (r >= j_idx)
        base = torch.where(r == i_idx, torch.full_like(base, n), base)
        state = _rows(torch.cat([state, new_row], 1), base)
    scores = torch.as_tensor(tree_scores).to(dtype)
    policy = (-(torch.cat(selected, 1).sum(1)) * (scores - baseline)).mean()
    return policy + (-sum(ents)) * strength, tables
