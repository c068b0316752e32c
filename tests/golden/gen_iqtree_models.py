"""Substitution-model pins the reference itself holds: every alignment of its bundled test set
(/root/reference/data_gen/data/test/<len>/<taxa>/*.phy) was simulated by IQ-TREE's AliSim, and the log of each
simulation (<name>_raw.tre.log) prints the GTR+F+I+G4 parameters it used together with DERIVED quantities -- the
normalised rate matrix Q and the relative rates of the four discrete-gamma categories -- to three significant digits.
This script parses the 1,152 logs into one numeric fixture (tests/golden/iqtree_models.npz): inputs (exchange rates,
base frequencies, proportion of invariable sites, gamma shape) and expected outputs (Q, category rates and
proportions).  Data only; nothing of the reference's code is involved.    python tests/golden/gen_iqtree_models.py"""
import glob
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = "/root/reference/data_gen/data/test"
NUM = r"([-+0-9.eE]+)"


def parse(path):
    txt = open(path).read()
    m = re.search(r"-m GTR\{" + ",".join([NUM] * 5) + r"\}\+F\{" + ",".join([NUM] * 4) + r"\}\+I\{" + NUM + r"\}\+G\{" + NUM + r"\}", txt)
    if m is None:
        return None
    v = [float(x) for x in m.groups()]
    rates, freqs, pinv, alpha = v[:5] + [1.0], v[5:9], v[9], v[10]
    q = re.search(r"Rate matrix Q:\s*\n\s*\n((?:\s+[ACGT](?:\s+" + NUM + r"){4}\s*\n){4})", txt)
    Q = np.array([[float(x) for x in ln.split()[1:]] for ln in q.group(1).strip().split("\n")])
    cats = re.findall(r"^\s+([0-4])\s+" + NUM + r"\s+" + NUM + r"\s*$", txt, flags=re.M)
    cat = np.array([[float(a), float(b)] for _, a, b in cats])
    assert Q.shape == (4, 4) and cat.shape == (5, 2), path
    return rates, freqs, pinv, alpha, Q, cat


def main():
    rows, names = [], []
    for p in sorted(glob.glob(os.path.join(ROOT, "*", "*", "*_raw.tre.log"))):
        r = parse(p)
        if r is not None:
            rows.append(r)
            names.append(os.path.relpath(p, ROOT))
    out = dict(names=np.array(names), rates=np.array([r[0] for r in rows]), freqs=np.array([r[1] for r in rows]),
               pinv=np.array([r[2] for r in rows]), alpha=np.array([r[3] for r in rows]),
               Q=np.array([r[4] for r in rows]), cat_rate=np.array([r[5][:, 0] for r in rows]),
               cat_prop=np.array([r[5][:, 1] for r in rows]))
    np.savez_compressed(os.path.join(HERE, "iqtree_models.npz"), **out)
    print(len(names), "logs ->", os.path.join(HERE, "iqtree_models.npz"))


if __name__ == "__main__":
    main()
