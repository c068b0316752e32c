"""Differentiable forward pass of PhyloATTN for the Finetune mode: the reference's model.py /
msa_modules.py / axial_attention.py forward, operator for operator, on the operators of train_ops.py (hand-written
gfx950 forward and backward kernels; torch.autograd keeps the graph).  Used by model.PhyloATTN when gradients are
enabled; inference (torch.no_grad) runs the fused kernels of libnnj_hip.so instead.

Dropout: the reference's Finetune loop runs the policy in eval mode -- its first baseline rollout calls agent.eval()
(finetune_rl_search.py:110,119) and nothing switches back -- so dropout is the identity there.  In train() mode (the
reference's train.py:435) the encoder applies nn.Dropout(model.dropout = 0.4) where the reference does: on the attention
probabilities (axial_attention.py:56,136,233), after the feed-forward GELU (msa_modules.py:149) and on every sublayer's
output before the residual add (msa_modules.py:119) -- train_ops.Dropout, a counter-based generator seeded by
torch.initial_seed().  The pair scorer has no dropout (model.py).
In grad mode the reference does not chunk the attention (axial_attention.py:127,243: `and not torch.is_grad_enabled()`),
so padded keys are filled with -10000 once.
"""
from __future__ import annotations

import math

import torch

from . import train_ops as T


def _lin(x, mod):
    return T.Linear.apply(x, mod.weight, mod.bias)


def _ln(x, mod):
    return T.LayerNorm.apply(x, mod.weight, mod.bias)


def _f32(t, dev):
    return t.to(device=dev, dtype=torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------ encoder
def row_attention(att, x, pad, p_drop=0.0):
    """RowSelfAttention.forward, tied over rows (axial_attention.py:66-138).  x [R,C,B,D]; pad bool [B,C] or None."""
    R, C, B, D = x.shape
    H = int(att.num_heads)
    dh = D // H
    scaling = dh ** -0.5 / math.sqrt(R)
    q = _lin(x, att.q_proj)
    k = _lin(x, att.k_proj)
    v = _lin(x, att.v_proj)
    # q *= scaling; q *= 1 - padding mask (zero at padded sites): one row scaling, token (r, c, b) -> s[c, b]
    s = torch.full((C, B), scaling, dtype=torch.float32, device=x.device)
    if pad is not None:
        s = s.masked_fill(pad.t(), 0.0)
    s = s.unsqueeze(0).expand(R, C, B).contiguous().view(-1)
    q = T.RowScale.apply(q, s)
    # 'rinhd,rjnhd->hnij' per (b, h): [C, R*dh] x [C, R*dh]^T
    qp = T.Permute.apply(q.view(R, C, B, H, dh), (2, 3, 1, 0, 4)).view(B * H, C, R * dh)
    kp = T.Permute.apply(k.view(R, C, B, H, dh), (2, 3, 1, 0, 4)).view(B * H, C, R * dh)
    vp = T.Permute.apply(v.view(R, C, B, H, dh), (2, 3, 1, 0, 4)).view(B * H, C, R * dh)
    logits = T.Bmm.apply(qp, kp, True, 1.0)                                  # [B*H, C(i), C(j)]
    if pad is not None:
        sel = pad.to(torch.uint8).contiguous()                                # key j of alignment b is padding
        logits = T.FillWhere.apply(logits, sel, -10000.0, H * C, B)
    probs = T.dropout(T.Softmax.apply(logits, None), p_drop)                  # axial_attention.py:136
    ctx = T.Bmm.apply(probs, vp, False, 1.0)                                  # [B*H, C, R*dh]
    ctx = T.Permute.apply(ctx.view(B, H, C, R, dh), (3, 2, 0, 1, 4)).view(R, C, B, D)
    return _lin(ctx, att.out_proj)


def column_attention(att, x, pad, p_drop=0.0):
    """ColumnSelfAttention.compute_attention_update (axial_attention.py:190-240).  x [R,C,B,D]."""
    R, C, B, D = x.shape
    H = int(att.num_heads)
    dh = D // H
    if R == 1:
        return _lin(_lin(x, att.v_proj), att.out_proj)
    q = T.Axpby.apply(_lin(x, att.q_proj), None, dh ** -0.5, 0.0)
    k = _lin(x, att.k_proj)
    v = _lin(x, att.v_proj)
    qp = T.Permute.apply(q.view(R, C, B, H, dh), (1, 2, 3, 0, 4)).view(C * B * H, R, dh)
    kp = T.Permute.apply(k.view(R, C, B, H, dh), (1, 2, 3, 0, 4)).view(C * B * H, R, dh)
    vp = T.Permute.apply(v.view(R, C, B, H, dh), (1, 2, 3, 0, 4)).view(C * B * H, R, dh)
    logits = T.Bmm.apply(qp, kp, True, 1.0)                                   # [C*B*H, R(i), R(j)]
    if pad is not None:
        # a padded column: every key gets -10000 (mask [B,R,C] is the site mask repeated over the rows)
        sel = pad.t().to(torch.uint8).unsqueeze(-1).expand(C, B, R).contiguous().view(C * B, R)
        logits = T.FillWhere.apply(logits, sel, -10000.0, H * R, C * B)
    probs = T.dropout(T.Softmax.apply(logits, None), p_drop)                   # axial_attention.py:233
    ctx = T.Bmm.apply(probs, vp, False, 1.0)                                   # [C*B*H, R, dh]
    ctx = T.Permute.apply(ctx.view(C, B, H, R, dh), (3, 0, 1, 2, 4)).view(R, C, B, D)
    return _lin(ctx, att.out_proj)


def encode(model, onehot, pad):
    """PhyloATTN.encode_zxr (model.py:67-88) with gradients.  onehot [B,R,L,4]; pad bool [B,L] -> [B,R,C,D]."""
    dev = next(model.parameters()).device
    x = _f32(onehot, dev)
    B, R, L, V = x.shape
    # Any model the reference's constructor accepts runs here: the operators are strided, batched products and
    # row-wise kernels of any width (the 64 x 64 weight-gradient kernel is a fast path taken by shape, train_ops.Linear),
    # so the narrow models the inference path pads onto its 64-feature kernels -- the reference's defaults (utils.py:45-52:
    # 32 features, 4 heads, patch 4) included -- need no padding here.
    P = int(model.patch_size)
    if L % P:
        raise ValueError(f"{L} sites are not a multiple of patch_size {P} (the reference's rearrange fails the same way, model.py:76)")
    model.patch_num = L // P
    if P > 1:
        x = x.view(B, R, L // P, P * V)                                        # 'b r (c k) e -> b r c (k e)'
        pad = None if pad is None else pad[:, ::P]                              # model.py:81
        L = L // P
    pad = None if pad is None else pad.to(dev).contiguous()
    # activations the per-step scorer calls of a whole episode would keep (~14 tensors of [B, n, C, D] per step, twice:
    # decode and merge): beyond a third of the device they run under activation checkpointing too (see decode)
    import os
    steps_bytes = 2 * 14 * B * (R * (R + 1) // 2) * L * 64 * 4
    model._train_ckpt = bool(int(os.environ.get("NNJ_TRAIN_PAIR_CHUNK", "0"))) or \
        steps_bytes > torch.cuda.get_device_properties(dev).total_memory // 3
    model.__dict__["_train_keys"] = None                                       # (keys of an earlier episode's states)
    x = _lin(T.Gelu.apply(_lin(x, model.embed[0])), model.embed[2])           # [B,R,C,D]
    x = T.Permute.apply(x, (1, 2, 0, 3))                                       # 'b r c d -> r c b d'
    pd = float(model.dropout) if model.training else 0.0                        # nn.Dropout: identity in eval mode
    for layer in model.seq_emb_layers:
        blk = layer.row_self_attention
        x = T.add(x, T.dropout(row_attention(blk.layer, _ln(x, blk.layer_norm), pad, pd), pd))
        blk = layer.column_self_attention
        x = T.add(x, T.dropout(column_attention(blk.layer, _ln(x, blk.layer_norm), pad, pd), pd))
        blk = layer.feed_forward_layer
        y = _ln(x, blk.layer_norm)
        y = T.dropout(T.Gelu.apply(_lin(y, blk.layer.fc1)), pd)                # msa_modules.py:148-149
        x = T.add(x, T.dropout(_lin(y, blk.layer.fc2), pd))                    # msa_modules.py:119
    return T.Permute.apply(x, (2, 0, 1, 3))                                    # 'r c b d -> b r c d'


# ------------------------------------------------------------------------------------------------ scorer
def _state_keys(model, state):
    """g_attn_k(state) [B,n,C,D] (model.py:117).  The reference transforms every row of the state in each aggregate call
    -- twice per NJ step (decode_zxr, env.step), although a step changes one row.  nn.Linear acts on rows, so the keys
    of the next state are the gathered keys of this one plus the transformed merged row (env_step); autograd carries
    the gradient through the gathers exactly as through the state itself.  The cache is keyed on the state TENSOR
    (the caller hands env.state_tensor back, as the reference's loop does); any other tensor is transformed afresh."""
    cache = model.__dict__.get("_train_keys")
    if _keys_valid(model, cache, state):
        return cache[1]
    k = _lin(state, model.g_attn_k)
    # (under activation checkpointing the closures recompute from the state: nothing is carried over)
    if not getattr(model, "_train_ckpt", False) and not model.__dict__.get("_train_nocache"):
        model.__dict__["_train_keys"] = (state, k, _keys_stamp(model, state))
    return k


def _keys_stamp(model, state):
    # version counters: an in-place write to the state, or to the weights (optimizer.step, load_state_dict), outdates the keys
    return (state._version, model.g_attn_k.weight._version, model.g_attn_k.bias._version)


def _keys_valid(model, cache, state):
    return cache is not None and cache[0] is state and cache[2] == _keys_stamp(model, state)


def aggregate(model, state, x_i, x_j, i_idx, j_idx):
    """PhyloATTN.aggregate (model.py:102-155).  state [B,n,C,D] (the stashed batch_input); x_i, x_j [B,p,C,D];
    i_idx, j_idx int64 [B,p]: the rows excluded from each pair's context."""
    B, n, C, D = state.shape
    p = x_i.shape[1]
    h = _lin(T.sub(x_i, x_j), model.h_linear_last)
    x = T.Gate.apply(h, x_i, x_j)                                              # z x_i + (1 - z) x_j
    if n <= 2:
        return x
    q = _lin(x, model.g_attn_q)
    k = _state_keys(model, state)
    alpha = T.Bmm.apply(q.view(B, p, C * D), k.view(B, n, C * D), True, 1.0 / math.sqrt(model.embed_dim * model.patch_num))
    r = torch.arange(n, device=state.device).view(1, 1, n)
    keep = ((r != i_idx.unsqueeze(-1)) & (r != j_idx.unsqueeze(-1))).to(torch.uint8).contiguous()
    alpha = T.Softmax.apply(alpha, keep)                                       # [B,p,n]
    xg = T.Bmm.apply(alpha, state.view(B, n, C * D), False, 1.0).view(B, p, C, D)
    g = _lin(xg, model.g_linear_last)
    return T.Gate.apply(g, xg, x)                                              # (1 - w) x + w x_g


def decode_gg(model, state, x_i, x_j, seq_keep, i_idx, j_idx):
    """PhyloATTN.decode_gg (model.py:90-99): masked site sum of s_out(aggregate(...)).  seq_keep float [B,C]."""
    B, p, C, D = x_i.shape
    x = aggregate(model, state, x_i, x_j, i_idx, j_idx)
    s = _lin(T.Gelu.apply(_lin(x, model.s_out[0])), model.s_out[2])            # [B,p,C,1]
    return T.Bmm.apply(s.view(B, p, C), seq_keep.view(B, C, 1), False, 1.0).view(B, p)


def decode(model, state, pad, info):
    """PhyloATTN.decode_zxr (model.py:158-209) with gradients."""
    dev = state.device
    B, n, C, D = state.shape
    actions_ij_prev, score_indices_to_prev, logits_prev = info
    if pad is not None and int(model.patch_size) > 1:
        pad = pad[:, :: int(model.patch_size)]                                  # the tokens' mask (model.py:163)
    keep = torch.ones((B, C), dtype=torch.float32, device=dev) if pad is None else (~pad.to(dev)).to(torch.float32)
    keep = keep.contiguous()
    state = state.contiguous()
    if logits_prev is None:
        # The unfused path keeps every activation of the all-pairs step: ~14 tensors of [B, pairs, C, D] floats.  Above
        # a budget (a third of the device, or NNJ_TRAIN_PAIR_CHUNK pairs) the pairs go through in chunks under
        # torch.utils.checkpoint: a chunk's activations are dropped after its forward and recomputed in its backward
        # (one more scorer forward), so BASELINE configs[4] (200 x 4096: 19,900 pairs, ~270 GB unchunked) fits.
        import os
        row, col = torch.triu_indices(n, n, offset=1, device=dev)
        P = row.numel()
        per_pair = 14 * B * C * D * 4
        budget = torch.cuda.get_device_properties(dev).total_memory // 3
        chunk = int(os.environ.get("NNJ_TRAIN_PAIR_CHUNK", "0")) or max(64, budget // per_pair)
        if chunk >= P:
            i_idx = row.unsqueeze(0).expand(B, -1).contiguous()
            j_idx = col.unsqueeze(0).expand(B, -1).contiguous()
            return decode_gg(model, state, T.GatherRows.apply(state, i_idx), T.GatherRows.apply(state, j_idx), keep, i_idx, j_idx)
        from torch.utils.checkpoint import checkpoint

        def part(st, lo, hi):
            i_idx = row[lo:hi].unsqueeze(0).expand(B, -1).contiguous()
            j_idx = col[lo:hi].unsqueeze(0).expand(B, -1).contiguous()
            return decode_gg(model, st, T.GatherRows.apply(st, i_idx), T.GatherRows.apply(st, j_idx), keep, i_idx, j_idx)

        out = None
        model.__dict__["_train_nocache"], model.__dict__["_train_keys"] = True, None
        try:
            for lo in range(0, P, chunk):
                piece = checkpoint(part, state, lo, min(P, lo + chunk), use_reentrant=False)
                out = piece if out is None else _cat_last(out, piece)
        finally:
            model.__dict__["_train_nocache"] = False
        return out
    ip = torch.as_tensor(actions_ij_prev).to(dev)[:, 0].to(torch.int64)
    r = torch.arange(n, device=dev, dtype=torch.int64).unsqueeze(0).expand(B, n)
    i_idx = torch.minimum(ip.unsqueeze(1), r).contiguous()                     # sort((i_prev, r))
    j_idx = torch.maximum(ip.unsqueeze(1), r).contiguous()
    def new_scores(st):
        return decode_gg(model, st, T.GatherRows.apply(st, i_idx), T.GatherRows.apply(st, j_idx), keep, i_idx, j_idx)

    if getattr(model, "_train_ckpt", False):
        from torch.utils.checkpoint import checkpoint
        new = checkpoint(new_scores, state, use_reentrant=False)
    else:
        new = new_scores(state)                                                # [B,n]
    table = _cat_last(logits_prev, new)
    idx = torch.as_tensor(score_indices_to_prev).to(dev).to(torch.int64)
    return T.GatherRows.apply(table.unsqueeze(-1), idx).squeeze(-1)           # gather(cat(prev, new), 1, idx)


class _CatLast(torch.autograd.Function):
    """torch.cat([a, b], -1) of two [B, *] tables (model.py:199) as two strided copies; the gradient splits back."""

    @staticmethod
    def forward(ctx, a, b):
        B, na = a.shape
        nb = b.shape[1]
        out = torch.empty((B, na + nb), dtype=torch.float32, device=a.device)
        out[:, :na].copy_(a)
        out[:, na:].copy_(b)
        ctx.na = na
        return out

    @staticmethod
    def backward(ctx, d):
        return d[:, :ctx.na].contiguous(), d[:, ctx.na:].contiguous()


def _cat_last(a, b):
    return _CatLast.apply(a.contiguous(), b.contiguous())


class _CatRows(torch.autograd.Function):
    """torch.cat((state, new), dim=1) of [B,n,...] and [B,m,...] (environment.py:833) as two copies."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]) + tuple(a.shape[2:]), dtype=torch.float32, device=a.device)
        out[:, :a.shape[1]].copy_(a)
        out[:, a.shape[1]:].copy_(b)
        ctx.na = a.shape[1]
        return out

    @staticmethod
    def backward(ctx, d):
        return d[:, :ctx.na].contiguous(), d[:, ctx.na:].contiguous()


def env_step(model, state, ij):
    """Tensor half of PhyInferEnv.step with gradients (environment.py:760-835): merge rows (i, j) with the other rows
    as context, put the merged row at position i, drop position j."""
    dev = state.device
    B, n, C, D = state.shape
    state = state.contiguous()
    ij = torch.as_tensor(ij).to(dev).to(torch.int64)
    i_idx, j_idx = ij[:, 0:1].contiguous(), ij[:, 1:2].contiguous()
    def merged(st):
        return aggregate(model, st, T.GatherRows.apply(st, i_idx), T.GatherRows.apply(st, j_idx), i_idx, j_idx)

    if getattr(model, "_train_ckpt", False):
        from torch.utils.checkpoint import checkpoint
        new = checkpoint(merged, state, use_reentrant=False)
    else:
        new = merged(state)
    r = torch.arange(n - 1, device=dev, dtype=torch.int64).unsqueeze(0).expand(B, n - 1)
    base = r + (r >= j_idx)                                  # positions of the old rows once j is gone ...
    base = torch.where(r == i_idx, torch.full_like(base, n), base).contiguous()   # ... the merged row (index n) at i
    out = T.GatherRows.apply(_CatRows.apply(state, new), base)
    cache = model.__dict__.get("_train_keys")
    if _keys_valid(model, cache, state) and n - 1 > 2:
        # keys of the next state: the old rows' keys move with their rows, the merged row is transformed alone
        keys = T.GatherRows.apply(_CatRows.apply(cache[1], _lin(new, model.g_attn_k)), base)
        model.__dict__["_train_keys"] = (out, keys, _keys_stamp(model, out))
    else:
        model.__dict__["_train_keys"] = None
    return out


class ExpandBatch(torch.autograd.Function):
    """x [1, ...] -> [B, ...] (B replicas of one encoded alignment: the reference re-encodes the same alignment for every
    replica of a batch); the gradient is the sum over the replicas (nnjt_sum_rows)."""

    @staticmethod
    def forward(ctx, x, B):
        ctx.shape = tuple(x.shape)
        return x.expand(B, *x.shape[1:]).contiguous()

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        B = d.shape[0]
        out = torch.empty(ctx.shape, dtype=torch.float32, device=d.device)
        T._chk(T.load_library().nnjt_sum_rows(T._p(d), T._p(out), B, d.numel() // B, T._st(d)))
        return out, None
