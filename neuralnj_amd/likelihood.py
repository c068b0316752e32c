"""Tree likelihood on the GPU (include/nnj.h: nnj_tree_loglik, nnj_tree_optimize): Felsenstein pruning under
GTR+I+G for batches of trees given as merge lists.  It stands where the reference calls raxmlpy.optimize_brlen to
score sampled trees (reference environment.py:365-441, 625-672; finetune_rl_search.py:401-411).  Model parameters
are inputs (defaults: Jukes-Cantor exchangeabilities and frequencies with gamma shape 1, four categories, no
invariant sites); only branch lengths are optimised."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import Nnj, NnjSubstModel, _p


def subst_model(rates=(1, 1, 1, 1, 1, 1), freqs=(0.25, 0.25, 0.25, 0.25), alpha=1.0, pinv=0.0, ncat=4) -> NnjSubstModel:
    m = NnjSubstModel()
    for i, r in enumerate(rates):
        m.rates[i] = float(r)
    for i, f in enumerate(freqs):
        m.freqs[i] = float(f)
    m.alpha, m.pinv, m.ncat = float(alpha), float(pinv), int(ncat)
    return m


def _lik_ws(ctx: Nnj, B, nA, T, L, ncat):
    need = C.c_size_t()
    rc = ctx.lib.nnj_lik_workspace_bytes(B, nA, T, L, ncat, C.byref(need))
    if rc != 0:
        raise RuntimeError("nnj_lik_workspace_bytes: bad argument")
    if getattr(ctx, "_lik_ws_t", None) is None or ctx._lik_ws_t.numel() < need.value:
        ctx._lik_ws_t = None
        ctx._lik_ws_t = torch.empty(need.value, dtype=torch.uint8, device=ctx.device)
    return ctx._lik_ws_t


def tree_loglik(ctx: Nnj, codes, merges, brlen=None, model=None, mask=None):
    """codes uint8 [1 or B, T, L]; merges int32 [B, T-1, 2]; brlen float [B, T-1, 2] or None -> float64 [B] (device)."""
    model = model or subst_model()
    codes = ctx._u8(codes)
    merges = ctx._i32(merges)
    nA, T, L = codes.shape
    B = merges.shape[0]
    mask = ctx._u8(mask)
    br = None if brlen is None else ctx._f32(brlen)
    out = torch.empty((B,), dtype=torch.float64, device=ctx.device)
    ws = _lik_ws(ctx, B, nA, T, L, model.ncat)
    ctx._chk(ctx.lib.nnj_tree_loglik(ctx.h, _p(codes), nA, _p(mask), _p(merges), _p(br), C.byref(model), B, T, L,
                                     _p(out), _p(ws), ws.numel(), ctx._stream()))
    return out


def tree_optimize(ctx: Nnj, codes, merges, brlen=None, model=None, mask=None, sweeps=3):
    """Branch-length optimisation (sweeps = the reference's iters=3), then the log-likelihood.
    -> (loglik float64 [B], brlen float32 [B, T-1, 2])."""
    model = model or subst_model()
    codes = ctx._u8(codes)
    merges = ctx._i32(merges)
    nA, T, L = codes.shape
    B = merges.shape[0]
    mask = ctx._u8(mask)
    br = None if brlen is None else ctx._f32(brlen)
    out = torch.empty((B,), dtype=torch.float64, device=ctx.device)
    br_out = torch.empty((B, T - 1, 2), dtype=torch.float32, device=ctx.device)
    ws = _lik_ws(ctx, B, nA, T, L, model.ncat)
    ctx._chk(ctx.lib.nnj_tree_optimize(ctx.h, _p(codes), nA, _p(mask), _p(merges), _p(br), C.byref(model), int(sweeps),
                                       B, T, L, _p(br_out), _p(out), _p(ws), ws.numel(), ctx._stream()))
    return out, br_out
