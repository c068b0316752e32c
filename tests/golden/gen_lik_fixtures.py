"""Likelihood fixtures from DATA the reference holds: three simulated alignments of its bundled test set
(data_gen/data/test/...: the .phy alignment, the generating tree <name>.tre with its branch lengths) as 1-byte site
codes + merge list + branch lengths, next to the GTR+I+G parameters IQ-TREE simulated them under (taken from
tests/golden/iqtree_models.npz).  Used to check that the model optimiser reaches at least the likelihood of the
generating parameters and lands near them.      python tests/golden/gen_lik_fixtures.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from neuralnj_amd import phydata, synth, utils  # noqa: E402

ROOT = "/root/reference/data_gen/data/test"
CASES = ("len1024/taxa20/G_l_1024_n_20_0_0.01_", "len512/taxa50/G_l_512_n_50_0_0.01_", "len256/taxa20/G_l_256_n_20_0_0.01_")


def main():
    z = np.load(os.path.join(HERE, "iqtree_models.npz"))
    names = [str(n) for n in z["names"]]
    out = {}
    for k, prefix in enumerate(CASES):
        idx = next(i for i, n in enumerate(names) if n.startswith(prefix))
        stem = os.path.join(ROOT, names[idx][:-len("_raw.tre.log")])
        seqs, keys, n_taxa, n_sites = phydata.load_phy_file_multirow(stem + ".phy")
        codes = synth.seqs_to_codes(seqs)
        merges, brlen = utils.newick_to_merges(open(stem + ".tre").read(), keys)
        out[f"codes_{k}"] = codes
        out[f"merges_{k}"] = merges
        out[f"brlen_{k}"] = brlen
        out[f"name_{k}"] = names[idx]
        for f in ("rates", "freqs", "pinv", "alpha"):
            out[f"{f}_{k}"] = z[f][idx]
        print(names[idx], codes.shape, "gaps", float((codes > 3).mean()))
    out["n"] = len(CASES)
    np.savez_compressed(os.path.join(HERE, "lik_fixtures.npz"), **out)


if __name__ == "__main__":
    main()
