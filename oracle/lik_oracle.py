"""CPU oracle of the tree-likelihood entry points (nnj_tree_loglik / nnj_tree_optimize).  TEST INFRASTRUCTURE ONLY.

The reference scores trees with raxml-ng / libpll through raxmlpy.optimize_brlen (environment.py:365-441, 625-672):
third-party code that is not in /root/reference (cloned at build time, RAxMLpy/setup.py:20,37), with no expected values
in the reference's own tests -- PARITY UNPINNED.  This file restates the published algorithm independently of the
HIP implementation so that the two can be compared:
  * transition matrices by scipy.linalg.expm of the rate matrix (the HIP code uses an eigen-decomposition);
  * discrete-gamma category rates from scipy.stats / scipy.special (the HIP code has its own incomplete-gamma code);
  * Felsenstein pruning by plain recursion over the merge list, in numpy float64;
  * a brute-force sum over all internal-state assignments for tiny trees (no pruning at all).
"""
from __future__ import annotations

import itertools

import numpy as np
from scipy.linalg import expm
from scipy.special import gammainc
from scipy.stats import gamma as gamma_dist

PAIR = {(0, 1): 0, (0, 2): 1, (0, 3): 2, (1, 2): 3, (1, 3): 4, (2, 3): 5}      # AC AG AT CG CT GT


def rate_matrix(rates, freqs):
    pi = np.asarray(freqs, float) / np.sum(freqs)
    Q = np.zeros((4, 4))
    for (i, j), k in PAIR.items():
        Q[i, j] = rates[k] * pi[j]
        Q[j, i] = rates[k] * pi[i]
    np.fill_diagonal(Q, -Q.sum(1))
    mu = -(pi * np.diag(Q)).sum()
    return Q / mu, pi


def gamma_rates(alpha, ncat):
    """Mean rate of each of ncat equiprobable categories of Gamma(alpha, rate alpha) (Yang 1994)."""
    if not alpha > 0 or ncat <= 1:
        return np.ones(max(ncat, 1))
    cuts = gamma_dist.ppf(np.arange(1, ncat) / ncat, alpha, scale=1.0 / alpha)
    cdf1 = np.concatenate([[0.0], gammainc(alpha + 1.0, cuts * alpha), [1.0]])
    return np.diff(cdf1) * ncat


def tip_vectors(codes_col):
    t = np.zeros((len(codes_col), 4))
    for r, c in enumerate(codes_col):
        if c > 3:
            t[r] = 1.0
        else:
            t[r, c] = 1.0
    return t


def program(merges, T):
    ids = list(range(T))
    prog = []
    for s, (i, j) in enumerate(merges):
        prog.append((ids[i], ids[j]))
        ids[i] = T + s
        ids.pop(j)
    return prog


def tree_loglik(codes, merges, brlen, model, mask=None):
    """codes [T,L] uint8, merges [T-1,2], brlen [T-1,2] (None: 0.1), model dict(rates, freqs, alpha, pinv, ncat)."""
    T, L = codes.shape
    Q, pi = rate_matrix(model["rates"], model["freqs"])
    rates = gamma_rates(model["alpha"], model["ncat"])
    pinv = model["pinv"]
    prog = program(merges, T)
    if brlen is None:
        brlen = np.full((T - 1, 2), 0.1)
    total = 0.0
    P = {}
    for s, (a, b) in enumerate(prog):
        for side, v in enumerate((a, b)):
            P[v] = [expm(Q * r * float(brlen[s][side])) for r in rates]
    for c in range(L):
        if mask is not None and mask[c]:
            continue
        tips = tip_vectors(codes[:, c])
        lik = 0.0
        for ci, _ in enumerate(rates):
            part = {}
            for s, (a, b) in enumerate(prog):
                xa = tips[a] if a < T else part[a]
                xb = tips[b] if b < T else part[b]
                part[T + s] = (P[a][ci] @ xa) * (P[b][ci] @ xb)
            lik += (pi * part[2 * T - 2]).sum()
        lik *= (1.0 - pinv) / len(rates)
        poss = np.ones(4, bool)
        for code in codes[:, c]:
            if code <= 3:
                poss &= np.arange(4) == code
        lik += pinv * pi[poss].sum()
        total += np.log(lik)
    return total


def tree_loglik_sites(codes, merges, brlen, model, mask=None):
    """The same pruning recursion as tree_loglik with the SITES as an array axis (numpy float64): what makes one
    200 x 4096 tree a second of CPU instead of minutes.  Pinned to the per-site loop above in tests/test_likelihood.py."""
    codes = np.asarray(codes).astype(np.int64)
    T, L = codes.shape
    Q, pi = rate_matrix(model["rates"], model["freqs"])
    rates = gamma_rates(model["alpha"], model["ncat"])
    pinv = model["pinv"]
    prog = program(merges, T)
    if brlen is None:
        brlen = np.full((T - 1, 2), 0.1)
    tips = np.zeros((T, L, 4))
    for k in range(4):
        tips[:, :, k] = (codes == k) | (codes > 3)
    lik = np.zeros(L)
    for r in rates:
        part = {}
        for s, (a, b) in enumerate(prog):
            Pa, Pb = expm(Q * r * float(brlen[s][0])), expm(Q * r * float(brlen[s][1]))
            xa = tips[a] if a < T else part.pop(a)
            xb = tips[b] if b < T else part.pop(b)
            part[T + s] = (xa @ Pa.T) * (xb @ Pb.T)
        lik += part[2 * T - 2] @ pi
    lik *= (1.0 - pinv) / len(rates)
    obs = np.where(codes <= 3, codes, -1)
    first = obs.max(0)
    const = ((obs == first[None]) | (obs < 0)).all(0)            # every taxon shows the same state or a gap
    inv = np.where(const, np.where(first >= 0, pi[np.maximum(first, 0)], 1.0), 0.0)
    lik += pinv * inv
    keep = np.ones(L, bool) if mask is None else ~np.asarray(mask, bool)
    return float(np.log(lik[keep]).sum())


def brute_force_loglik(codes, merges, brlen, model):
    """Sum over every assignment of states to the internal nodes (tiny trees only): no pruning, no recursion."""
    T, L = codes.shape
    assert T <= 5
    Q, pi = rate_matrix(model["rates"], model["freqs"])
    rates = gamma_rates(model["alpha"], model["ncat"])
    prog = program(merges, T)
    root = 2 * T - 2
    total = 0.0
    for c in range(L):
        tips = tip_vectors(codes[:, c])
        lik = 0.0
        for r in rates:
            Pm = {}
            for s, (a, b) in enumerate(prog):
                Pm[a] = expm(Q * r * float(brlen[s][0]))
                Pm[b] = expm(Q * r * float(brlen[s][1]))
            for states in itertools.product(range(4), repeat=T - 1):          # internal nodes T .. 2T-2
                st = {T + k: x for k, x in enumerate(states)}
                p = pi[st[root]]
                for s, (a, b) in enumerate(prog):
                    for v in (a, b):
                        if v < T:
                            p *= (Pm[v][st[T + s]] * tips[v]).sum()
                        else:
                            p *= Pm[v][st[T + s], st[v]]
                lik += p
        lik *= (1.0 - model["pinv"]) / len(rates)
        poss = np.ones(4, bool)
        for code in codes[:, c]:
            if code <= 3:
                poss &= np.arange(4) == code
        lik += model["pinv"] * pi[poss].sum()
        total += np.log(lik)
    return total
