import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'oracle'); sys.path.insert(0,'tests')
import lik_oracle as LO
from scipy.linalg import expm
from scipy.optimize import minimize
from neuralnj_amd import likelihood as lk, utils
from neuralnj_amd._lib import Nnj
JC = dict(rates=[1] * 6, freqs=[0.25] * 4, alpha=0.0, pinv=0.0, ncat=1)
rng = np.random.default_rng(11)
T, L = 6, 3000
merges = np.array([[0, 1], [1, 2], [0, 1], [1, 2], [0, 1]], np.int32)
true_br = rng.uniform(0.05, 0.3, size=(T - 1, 2))
Q, pi = LO.rate_matrix(JC["rates"], JC["freqs"])
prog = LO.program(merges, T)
seq = {2 * T - 2: rng.choice(4, size=L, p=pi)}
for s in range(T - 2, -1, -1):
    for side, v in enumerate(prog[s]):
        P = expm(Q * true_br[s][side]); cum = P[seq[T + s]].cumsum(1)
        seq[v] = (rng.random(L)[:, None] > cum).sum(1).clip(0, 3)
codes = np.stack([seq[i] for i in range(T)]).astype(np.uint8)[None]
g = Nnj(utils.shipped_config(), "cuda:0")
m = lk.subst_model(**JC)
start = np.full((1, T - 1, 2), 0.1, np.float32)
print(os.environ.get("NNJ_LIK_NOFOLD"), [round(lk.tree_optimize(g, codes, merges[None], start, m, sweeps=sw)[0].item(), 3) for sw in (1,2,3,4,6,8,12,20)])
z = np.load('tests/golden/lik_fixtures.npz')
for k in range(3):
    true = dict(rates=list(z[f"rates_{k}"]), freqs=list(z[f"freqs_{k}"]), alpha=float(z[f"alpha_{k}"]), pinv=float(z[f"pinv_{k}"]), ncat=4)
    print(k, [round(lk.tree_optimize(g, z[f"codes_{k}"][None], z[f"merges_{k}"][None], None, lk.subst_model(**true), sweeps=sw)[0].item(), 3) for sw in (1,2,3,6,12,30)])
