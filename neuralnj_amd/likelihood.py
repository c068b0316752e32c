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


def model_probe(ctx: Nnj, model: NnjSubstModel, t: float = 0.1):
    """(Q [4,4], category rates [ncat], P [ncat,4,4]): the normalised rate matrix and gamma rates the library builds
    from `model`, and the transition matrices of one branch of length t as the device kernel forms them
    (nnj_lik_model_probe; pinned to IQ-TREE's printed values in tests/test_likelihood.py)."""
    import numpy as np
    Q = np.zeros(16, np.float64)
    rates = np.zeros(8, np.float64)
    P = np.zeros(int(model.ncat) * 16, np.float64)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    ctx._chk(ctx.lib.nnj_lik_model_probe(ctx.h, C.byref(model), C.c_double(float(t)), dp(Q), dp(rates), dp(P)))
    return Q.reshape(4, 4), rates[:int(model.ncat)].copy(), P.reshape(int(model.ncat), 4, 4)


def empirical_freqs(codes):
    """Base frequencies of an alignment (codes uint8 [*, T, L]; gaps / N / padding ignored), with a pseudo-count."""
    c = torch.as_tensor(codes).reshape(-1).to(torch.int64)
    cnt = torch.bincount(c[c < 4], minlength=4).to(torch.float64) + 1.0
    return (cnt / cnt.sum()).tolist()


def optimize_model(ctx: Nnj, codes, merges, brlen=None, model=None, mask=None, rounds=3, sweeps=3, optimize_freqs=True,
                   verbose=False):
    """Maximum-likelihood estimate of the substitution-model parameters on ONE tree -- the `opt_model=True` of the
    reference's raxmlpy.optimize_brlen(..., model='GTR+I+G', opt_model=True) call (environment.py:373-377): `rounds` of
    (branch lengths by nnj_tree_optimize; then one bounded Brent search per parameter -- gamma shape, proportion of
    invariable sites, the five free exchange rates, in that order -- on nnj_tree_loglik with the lengths held).
    Frequencies start from the model's, or from the alignment's empirical ones when `model` is None, and are
    optimised too (raxml-ng's default for GTR is ML frequencies; optimize_freqs=False keeps them fixed).
    codes uint8 [1, T, L], merges int32 [1, T-1, 2].  Returns (model, loglik float, brlen float32 [1, T-1, 2]).
    Trees of the same alignment scored together share the result (the reference re-estimates per tree; the parameters
    belong to the alignment far more than to the topology) -- rollout.search_rollouts(model="auto")."""
    import numpy as np
    from scipy.optimize import minimize_scalar
    codes = ctx._u8(codes)[:1]
    merges = ctx._i32(merges)[:1]
    if model is None:
        model = subst_model(freqs=empirical_freqs(codes.cpu()), alpha=1.0, pinv=0.1, ncat=4)
    m = NnjSubstModel()
    C.memmove(C.byref(m), C.byref(model), C.sizeof(NnjSubstModel))
    # an invariable-sites share above the share of constant columns has likelihood 0 for the variable ones
    cc = codes[0].to(torch.int64)
    obs = torch.where(cc < 4, cc, torch.full_like(cc, -1))
    first = obs.max(dim=0).values
    const = ((obs == first[None]) | (obs < 0)).all(dim=0)
    if mask is not None:
        keep = ~ctx._u8(mask)[0].to(torch.bool)
        pmax = float((const & keep).sum()) / max(1.0, float(keep.sum()))
    else:
        pmax = float(const.to(torch.float64).mean())
    pmax = min(0.99, max(pmax, 1e-3))
    ll, br = tree_optimize(ctx, codes, merges, brlen, m, mask=mask, sweeps=sweeps)
    best = float(ll[0])

    def search(get, put, lo, hi, log):
        nonlocal best
        x0 = get()

        def f(x):
            put(float(np.exp(x)) if log else float(x))
            return -float(tree_loglik(ctx, codes, merges, br, m, mask=mask)[0])
        a, b_ = (np.log(lo), np.log(hi)) if log else (lo, hi)
        res = minimize_scalar(f, bounds=(a, b_), method="bounded", options=dict(xatol=1e-3, maxiter=40))
        if -res.fun >= best:
            put(float(np.exp(res.x)) if log else float(res.x))
            best = -float(res.fun)
        else:
            put(x0)

    def attr(name):
        return (lambda: getattr(m, name)), (lambda v: setattr(m, name, v))

    def rate(k):
        return (lambda: m.rates[k]), (lambda v: m.rates.__setitem__(k, v))

    def freq(k):                                              # ratio to the last frequency; renormalised by the library
        return (lambda: m.freqs[k]), (lambda v: m.freqs.__setitem__(k, v))

    def rate_scale():
        # all five free rates times one factor (= the G-T rate against the others): the direction along which the
        # one-at-a-time searches only crawl (each rate can move no further than the fixed G-T rate lets it)
        base = [m.rates[k] for k in range(5)]

        def put(v):
            for k in range(5):
                m.rates[k] = base[k] * v
        return (lambda: 1.0), put

    for rd in range(int(rounds)):
        if m.alpha > 0:
            search(*attr("alpha"), 0.02, 100.0, True)
        search(*attr("pinv"), 0.0, pmax, False)
        search(*rate_scale(), 0.05, 20.0, True)
        for k in range(5):
            search(*rate(k), 1e-3, 1e3, True)
        if optimize_freqs:
            for k in range(3):
                search(*freq(k), 1e-3, 1e3, True)
            tot = sum(m.freqs[k] for k in range(4))
            for k in range(4):
                m.freqs[k] = m.freqs[k] / tot
        ll, br = tree_optimize(ctx, codes, merges, br, m, mask=mask, sweeps=sweeps)
        best = float(ll[0])
        if verbose:
            print(f"round {rd}: loglik {best:.4f} alpha {m.alpha:.4f} pinv {m.pinv:.4f} rates {[round(m.rates[k], 4) for k in range(6)]}")
    return m, best, br
