"""Diagnostic: likelihood.optimize_model on the reference-data fixtures, round by round."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from neuralnj_amd import likelihood as lk, utils
from neuralnj_amd._lib import Nnj
z = np.load('tests/golden/lik_fixtures.npz')
g = Nnj(utils.shipped_config(), "cuda:0")
for k in (1,):
    codes, merges, br = z[f"codes_{k}"][None], z[f"merges_{k}"][None], z[f"brlen_{k}"][None]
    true = dict(rates=list(z[f"rates_{k}"]), freqs=list(z[f"freqs_{k}"]), alpha=float(z[f"alpha_{k}"]), pinv=float(z[f"pinv_{k}"]), ncat=4)
    print("true", true)
    for sw in (12, 40):
        print("generating params, br from truth, sweeps", sw, lk.tree_optimize(g, codes, merges, br, lk.subst_model(**true), sweeps=sw)[0].item())
        print("generating params, br from 0.1,  sweeps", sw, lk.tree_optimize(g, codes, merges, None, lk.subst_model(**true), sweeps=sw)[0].item())
    m, ll, brh = lk.optimize_model(g, codes, merges, None, None, rounds=8, sweeps=6, verbose=True)
    print("then 40 more sweeps:", lk.tree_optimize(g, codes, merges, brh, m, sweeps=40)[0].item())
