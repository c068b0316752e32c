"""Tree likelihood.  Parity with the reference is UNPINNED (raxml-ng / libpll are not in the mount and the
reference's tests at that boundary print without asserting), so these are known-answer tests: closed forms, a
brute-force sum over internal states, invariances of the likelihood, and the HIP kernels against the independent
numpy/scipy evaluation of oracle/lik_oracle.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lik_oracle as LO  # noqa: E402

GTR = dict(rates=[1.3, 4.1, 0.7, 1.1, 5.2, 1.0], freqs=[0.31, 0.19, 0.23, 0.27], alpha=0.6, pinv=0.15, ncat=4)
JC = dict(rates=[1] * 6, freqs=[0.25] * 4, alpha=0.0, pinv=0.0, ncat=1)


def _random_tree(T, rng):
    merges, n = [], T
    while n > 1:
        i, j = sorted(rng.choice(n, 2, replace=False))
        merges.append((int(i), int(j)))
        n -= 1
    return np.array(merges, np.int32)


def test_oracle_two_taxa_jc69_closed_form():
    t = 0.37
    codes = np.array([[0, 1, 2, 3, 0, 2], [0, 1, 3, 3, 1, 2]], np.uint8)
    same, diff = 0.25 * (0.25 + 0.75 * np.exp(-4 * t / 3)), 0.25 * (0.25 - 0.25 * np.exp(-4 * t / 3))
    want = 4 * np.log(same) + 2 * np.log(diff)
    got = LO.tree_loglik(codes, np.array([[0, 1]]), np.array([[0.3 * t, 0.7 * t]]), JC)
    assert abs(got - want) < 1e-12


def test_oracle_pruning_equals_brute_force_and_gamma_rates():
    rng = np.random.default_rng(1)
    r = LO.gamma_rates(0.6, 4)
    assert abs(r.mean() - 1.0) < 1e-12 and (np.diff(r) > 0).all()
    # Yang (1994) table: alpha = 0.5, four categories, mean rates
    np.testing.assert_allclose(LO.gamma_rates(0.5, 4), [0.03338775, 0.25191592, 0.82026848, 2.89442785], rtol=1e-6)
    codes = rng.integers(0, 5, size=(4, 7)).astype(np.uint8)
    merges = np.array([[1, 3], [0, 2], [0, 1]], np.int32)
    br = rng.uniform(0.02, 0.6, size=(3, 2))
    a = LO.tree_loglik(codes, merges, br, GTR)
    b = LO.brute_force_loglik(codes, merges, br, GTR)
    assert abs(a - b) < 1e-10
    # pulley principle: only the SUM of the two root edges matters (time-reversible model)
    br2 = br.copy()
    br2[2] = [br[2].sum() * 0.9, br[2].sum() * 0.1]
    assert abs(LO.tree_loglik(codes, merges, br2, GTR) - a) < 1e-10


def test_oracle_site_vectorised_form_equals_the_site_loop():
    """tree_loglik_sites (sites as an array axis: the oracle of the configs[4] Search test) == the per-site recursion."""
    rng = np.random.default_rng(11)
    for (T, L, model) in ((2, 9, JC), (6, 50, GTR), (13, 41, dict(GTR, ncat=6)), (9, 33, dict(GTR, pinv=0.0, alpha=0.0, ncat=1))):
        codes = rng.integers(0, 6, size=(T, L)).astype(np.uint8)
        codes[:, :5] = codes[:1, :5]                           # constant columns (the +I term), some of them gaps only
        codes[:, 5] = 4
        mask = np.zeros(L, bool)
        mask[-4:] = True
        merges = _random_tree(T, rng)
        br = rng.uniform(0.01, 0.5, size=(T - 1, 2))
        for mk in (None, mask):
            a = LO.tree_loglik(codes, merges, br, model, mk)
            b = LO.tree_loglik_sites(codes, merges, br, model, mk)
            assert abs(a - b) <= 1e-10 * max(1.0, abs(a)), (T, L, a, b)


@pytest.mark.gpu
def test_hip_loglik_matches_numpy_oracle():
    import torch
    from neuralnj_amd import likelihood as lk, utils
    from neuralnj_amd._lib import Nnj
    rng = np.random.default_rng(7)
    g = Nnj(utils.shipped_config(), "cuda:0")
    for (T, L, model) in ((2, 6, JC), (5, 40, GTR), (12, 65, GTR), (9, 33, dict(GTR, pinv=0.0, alpha=0.0, ncat=1)),
                          (70, 20, dict(GTR, ncat=6))):
        B = 3
        codes = rng.integers(0, 5, size=(B, T, L)).astype(np.uint8)
        codes[:, :, :4] = codes[:, :1, :4]                     # a few constant columns (the +I component)
        mask = np.zeros((B, L), bool)
        mask[:, -3:] = True
        merges = np.stack([_random_tree(T, rng) for _ in range(B)])
        br = rng.uniform(0.01, 0.5, size=(B, T - 1, 2)).astype(np.float32)
        got = lk.tree_loglik(g, codes, merges, br, lk.subst_model(**model), mask=mask).cpu().numpy()
        want = [LO.tree_loglik(codes[b], merges[b], br[b].astype(np.float64), model, mask[b]) for b in range(B)]
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-9)
        # n_align = 1: B trees over ONE alignment
        got1 = lk.tree_loglik(g, codes[:1], merges, br, lk.subst_model(**model), mask=mask[:1]).cpu().numpy()
        want1 = [LO.tree_loglik(codes[0], merges[b], br[b].astype(np.float64), model, mask[0]) for b in range(B)]
        np.testing.assert_allclose(got1, want1, rtol=1e-10, atol=1e-9)
    g.close()


@pytest.mark.gpu
def test_hip_malformed_merge_list_is_reported():
    """merge lists reach nnj_tree_loglik from user buffers.  A pair with i >= j (or a position outside the live
    list) used to yield a silently wrong program; now the kernel that reads the list sets NNJ_STATUS_BAD_MERGE and
    check_numeric raises -- and a well-formed list afterwards is clean again (the flag is cleared by the read)."""
    from neuralnj_amd import likelihood as lk, utils
    from neuralnj_amd._lib import Nnj
    rng = np.random.default_rng(3)
    g = Nnj(utils.shipped_config(), "cuda:0")
    T, L = 7, 30
    codes = rng.integers(0, 4, size=(1, T, L)).astype(np.uint8)
    good = _random_tree(T, rng)[None]
    for bad_step, bad_pair in ((2, (3, 1)), (1, (2, 2)), (0, (0, T))):
        bad = good.copy()
        bad[0, bad_step] = bad_pair
        lk.tree_loglik(g, codes, bad, None, lk.subst_model(**GTR))
        with pytest.raises(ValueError, match="BAD_MERGE"):
            g.check_numeric()
    lk.tree_loglik(g, codes, good, None, lk.subst_model(**GTR))
    g.check_numeric()
    g.close()


@pytest.mark.gpu
def test_hip_branch_length_optimisation():
    """Data simulated down a known tree: the optimiser must (1) never lower the likelihood, (2) reach the value a
    generic scipy optimiser finds for the same tree, (3) recover branch lengths near the simulated ones, and
    (4) rank the true topology above a random one."""
    import torch
    from scipy.optimize import minimize
    from neuralnj_amd import likelihood as lk, utils
    from neuralnj_amd._lib import Nnj
    rng = np.random.default_rng(11)
    T, L = 6, 3000
    merges = np.array([[0, 1], [1, 2], [0, 1], [1, 2], [0, 1]], np.int32)      # ((0,1),((2,3),(4,5)))-like
    true_br = rng.uniform(0.05, 0.3, size=(T - 1, 2))
    Q, pi = LO.rate_matrix(JC["rates"], JC["freqs"])
    from scipy.linalg import expm
    prog = LO.program(merges, T)
    seq = {2 * T - 2: rng.choice(4, size=L, p=pi)}
    for s in range(T - 2, -1, -1):
        for side, v in enumerate(prog[s]):
            P = expm(Q * true_br[s][side])
            cum = P[seq[T + s]].cumsum(1)
            seq[v] = (rng.random(L)[:, None] > cum).sum(1).clip(0, 3)
    codes = np.stack([seq[i] for i in range(T)]).astype(np.uint8)[None]
    g = Nnj(utils.shipped_config(), "cuda:0")
    m = lk.subst_model(**JC)
    start = np.full((1, T - 1, 2), 0.1, np.float32)
    ll0 = lk.tree_loglik(g, codes, merges[None], start, m).item()
    lls = [ll0]
    for sw in (1, 2, 3, 6, 12):                              # (Jacobi sweeps: all edges at once; 12 reach the optimum to 1e-3)
        ll, br = lk.tree_optimize(g, codes, merges[None], start, m, sweeps=sw)
        lls.append(ll.item())
    assert all(b >= a - 1e-9 for a, b in zip(lls, lls[1:])), lls
    br = br.cpu().numpy()[0].astype(np.float64)
    assert abs(lk.tree_loglik(g, codes, merges[None], br[None].astype(np.float32), m).item() - lls[-1]) < 1e-6 * abs(lls[-1])

    def neg(x):
        b = np.abs(x).reshape(T - 1, 2)
        return -LO.tree_loglik(codes[0], merges, b, JC)
    ref = minimize(neg, np.full(2 * (T - 1), 0.1), method="L-BFGS-B", bounds=[(1e-6, 5)] * (2 * (T - 1)))
    assert lls[-1] >= -ref.fun - 0.05, (lls[-1], -ref.fun)          # as good as a generic optimiser (log units)
    est, tru = br.copy(), true_br.copy()
    est[-1], tru[-1] = est[-1].sum(), tru[-1].sum()                 # the root's two edges are one edge
    assert np.abs(est - tru).max() < 0.06
    wrong = np.array([[0, 4], [0, 2], [1, 2], [0, 2], [0, 1]], np.int32)
    llw, _ = lk.tree_optimize(g, codes, wrong[None], None, m, sweeps=3)
    assert lls[-1] > llw.item() + 10
    g.close()


@pytest.mark.gpu
def test_hip_root_edge_is_one_edge():
    """The two child edges of the last join are one edge of the unrooted tree.  Two taxa under JC69: with p the share
    of differing sites the ML distance is -3/4 ln(1 - 4p/3) in closed form; the optimiser must reach it after ONE
    sweep and stay there (a Jacobi update of the two halves mirrors the sum about the optimum instead), from a
    symmetric and from a lopsided start, and the exported halves must add up to it.  Then the same on the root edge of
    a four-taxon tree against scipy."""
    from scipy.optimize import minimize_scalar
    from neuralnj_amd import likelihood as lk, utils
    from neuralnj_amd._lib import Nnj
    rng = np.random.default_rng(3)
    L = 2000
    a = rng.integers(0, 4, size=L)
    b = a.copy()
    flip = rng.random(L) < 0.2
    b[flip] = (a[flip] + rng.integers(1, 4, size=int(flip.sum()))) % 4
    codes = np.stack([a, b]).astype(np.uint8)[None]
    p = float((a != b).mean())
    d = -0.75 * np.log(1.0 - 4.0 * p / 3.0)
    g = Nnj(utils.shipped_config(), "cuda:0")
    m = lk.subst_model(**JC)
    merges = np.array([[[0, 1]]], np.int32)
    want = LO.tree_loglik(codes[0], merges[0], np.array([[0.5 * d, 0.5 * d]]), JC)
    for start in ([[0.1, 0.1]], [[0.9, 0.02]], [[0.01, 0.01]]):
        for sweeps in (1, 3):
            ll, br = lk.tree_optimize(g, codes, merges, np.array([start], np.float32), m, sweeps=sweeps)
            br = br.cpu().numpy()[0, 0].astype(np.float64)
            assert abs(br.sum() - d) < 2e-5 * d, (start, sweeps, br, d)
            assert abs(ll.item() - want) < 1e-6, (start, sweeps, ll.item(), want)
    # four taxa, GTR+I+G: all lengths but the root pair's held at their values, the sum of the pair scanned by scipy
    T = 4
    def mutate(seq, p):
        hit = rng.random(seq.size) < p
        return np.where(hit, (seq + rng.integers(1, 4, size=seq.size)) % 4, seq)
    anc = rng.integers(0, 4, size=600)
    left, right = mutate(anc, 0.08), mutate(anc, 0.12)       # the root edge: about 0.2 substitutions per site
    codes4 = np.stack([mutate(left, 0.1), mutate(left, 0.15), mutate(right, 0.1), mutate(right, 0.2)]).astype(np.uint8)[None]
    mg = np.array([[[0, 1], [1, 2], [0, 1]]], np.int32)
    mod = dict(GTR)
    ll1, br1 = lk.tree_optimize(g, codes4, mg, None, lk.subst_model(**mod), sweeps=1)
    ll6, br6 = lk.tree_optimize(g, codes4, mg, None, lk.subst_model(**mod), sweeps=40)
    brn = br6.cpu().numpy()[0].astype(np.float64)

    def neg(s):
        bb = brn.copy()
        bb[-1] = [0.5 * s, 0.5 * s]
        return -LO.tree_loglik(codes4[0], mg[0], bb, mod)
    best = minimize_scalar(neg, bounds=(1e-6, 3.0), method="bounded", options=dict(xatol=1e-9))
    assert abs(brn[-1].sum() - best.x) < 3e-3 * max(best.x, 1e-2), (brn[-1], best.x)
    assert ll6.item() >= ll1.item() - 1e-9
    g.close()


# ---------------------------------------------------------------- pins the reference holds: IQ-TREE's simulation logs
IQTREE = os.path.join(ROOT, "tests", "golden", "iqtree_models.npz")


def _close_3sf(got, want, what):
    """IQ-TREE prints three significant digits: |got - want| within half a unit of the third digit of `want`."""
    got, want = np.asarray(got, float), np.asarray(want, float)
    digit = 10.0 ** (np.floor(np.log10(np.maximum(np.abs(want), 1e-300))) - 2)
    bad = np.abs(got - want) > 0.5001 * digit + 1e-12
    assert not bad.any(), f"{what}: {got[bad][:4]} vs printed {want[bad][:4]}"


def test_oracle_matches_iqtree_logs():
    """The numpy oracle's normalised GTR rate matrix and discrete-gamma category rates against what IQ-TREE printed
    for each of the 1,152 simulated alignments of the reference's bundled test set (tests/golden/gen_iqtree_models.py).
    IQ-TREE reports the category rates of +I+G rescaled by 1 / (1 - pinv) (mean rate 1 over ALL sites); the oracle and
    the library keep mean 1 over the gamma part -- a convention that only rescales branch lengths."""
    z = np.load(IQTREE)
    assert len(z["names"]) == 1152
    for i in range(len(z["names"])):
        Q, pi = LO.rate_matrix(z["rates"][i], z["freqs"][i])
        _close_3sf(Q, z["Q"][i], f"Q of {z['names'][i]}")
        r = LO.gamma_rates(float(z["alpha"][i]), 4) / (1.0 - float(z["pinv"][i]))
        _close_3sf(r, z["cat_rate"][i][1:], f"category rates of {z['names'][i]}")
        _close_3sf([float(z["pinv"][i])] + [(1.0 - float(z["pinv"][i])) / 4] * 4, z["cat_prop"][i], "category proportions")


@pytest.mark.gpu
def test_hip_model_matches_iqtree_logs():
    """The library's model (nnj_lik_model_probe: host-side eigen system, own incomplete-gamma code) and the DEVICE
    transition matrices against the same 1,152 logs: Q and the category rates to the three printed digits; the
    derivative of the device's P(t) at t -> 0 is Q x rate; P(t) at t = 0.3 equals expm(Q rate t) of the pinned Q's
    parameters to 1e-12."""
    from scipy.linalg import expm
    from neuralnj_amd import likelihood as lk, utils
    from neuralnj_amd._lib import Nnj
    z = np.load(IQTREE)
    g = Nnj(utils.shipped_config(), "cuda:0")
    for i in range(len(z["names"])):
        m = lk.subst_model(rates=z["rates"][i], freqs=z["freqs"][i], alpha=float(z["alpha"][i]), pinv=float(z["pinv"][i]), ncat=4)
        deep = i % 16 == 0
        Q, rates, P = lk.model_probe(g, m, 1e-7 if deep else 0.3)
        _close_3sf(Q, z["Q"][i], f"Q of {z['names'][i]}")
        _close_3sf(rates / (1.0 - float(z["pinv"][i])), z["cat_rate"][i][1:], f"category rates of {z['names'][i]}")
        Qo, _ = LO.rate_matrix(z["rates"][i], z["freqs"][i])
        np.testing.assert_allclose(Q, Qo, atol=1e-12)
        if deep:
            for c in range(4):
                np.testing.assert_allclose((P[c] - np.eye(4)) / 1e-7, Qo * rates[c], atol=2e-6 * max(1.0, rates[c]))
        else:
            for c in range(4):
                np.testing.assert_allclose(P[c], expm(Qo * rates[c] * 0.3), atol=1e-12)
    g.close()


@pytest.mark.gpu
def test_hip_model_optimisation_recovers_the_simulation():
    """likelihood.optimize_model (the reference's opt_model=True, environment.py:373-377) on three alignments of the
    reference's bundled test set, each on its generating tree (tests/golden/gen_lik_fixtures.py): starting from
    Jukes-Cantor-like values the optimiser must reach at least the likelihood of the parameters IQ-TREE simulated the
    data under (branch lengths optimised under both), agree with the numpy oracle on that likelihood, and land near
    those parameters (finite-sample scatter: 256 .. 1024 sites)."""
    from neuralnj_amd import likelihood as lk, utils
    from neuralnj_amd._lib import Nnj
    z = np.load(os.path.join(ROOT, "tests", "golden", "lik_fixtures.npz"))
    g = Nnj(utils.shipped_config(), "cuda:0")
    for k in range(int(z["n"])):
        codes, merges, br = z[f"codes_{k}"][None], z[f"merges_{k}"][None], z[f"brlen_{k}"][None]
        true = dict(rates=list(z[f"rates_{k}"]), freqs=list(z[f"freqs_{k}"]), alpha=float(z[f"alpha_{k}"]),
                    pinv=float(z[f"pinv_{k}"]), ncat=4)
        ll_true, _ = lk.tree_optimize(g, codes, merges, br, lk.subst_model(**true), sweeps=12)
        m, ll_hat, br_hat = lk.optimize_model(g, codes, merges, None, None, rounds=4, sweeps=6)
        got = dict(rates=[m.rates[i] for i in range(6)], freqs=[m.freqs[i] for i in range(4)], alpha=m.alpha, pinv=m.pinv, ncat=4)
        want = LO.tree_loglik(codes[0], merges[0], br_hat.cpu().numpy()[0].astype(np.float64), got)
        assert abs(ll_hat - want) < 1e-6 * abs(want), (ll_hat, want)
        assert ll_hat >= ll_true.item() - 2.0, (str(z[f"name_{k}"]), ll_hat, ll_true.item())
        L = codes.shape[2]
        tol = 1.0 if L >= 1024 else 1.6                         # log-ratio scatter of the estimates
        assert abs(np.log(m.alpha / true["alpha"])) < tol, (m.alpha, true["alpha"])
        assert abs(m.pinv - true["pinv"]) < 0.2, (m.pinv, true["pinv"])
        lr = np.log(np.array(got["rates"][:5]) / np.array(true["rates"][:5]))
        assert np.abs(lr).max() < tol, (got["rates"], true["rates"])
        print(f"{z[f'name_{k}']}: loglik {ll_hat:.2f} (generating parameters {ll_true.item():.2f}); alpha {m.alpha:.3f} / "
              f"{true['alpha']:.3f}, pinv {m.pinv:.3f} / {true['pinv']:.3f}, rates {np.round(got['rates'][:5], 2)} / {np.round(true['rates'][:5], 2)}")
    g.close()
