"""Synthetic MSA generation and the 1-byte site code used on the device.

Site codes (lossless re-encoding of the reference's 4-vector per site,
reference phydata.py:38-46,57-77): 0..3 = A,C,G,T one-hot; 4 = gap / N =
[1,1,1,1]; 5 = padding '*' = [0,0,0,0].
"""
from __future__ import annotations

import numpy as np

CODE_CHARS = "ACGT-*"
CODE_TO_ONEHOT = np.array(
    [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [1, 1, 1, 1], [0, 0, 0, 0]],
    dtype=np.int8,
)
_CHAR_TO_CODE = {"A": 0, "C": 1, "G": 2, "T": 3, "-": 4, "N": 4, "*": 5}


def synth_codes(batch: int, taxa: int, sites: int, seed: int, gap_frac: float = 0.2) -> np.ndarray:
    """i.i.d. codes over {A,C,G,T} with a gap fraction."""
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=(batch, taxa, sites), dtype=np.uint8)
    if gap_frac > 0:
        gaps = rng.random((batch, taxa, sites)) < gap_frac
        codes[gaps] = 4
    return codes


def synth_codes_tree(batch: int, taxa: int, sites: int, seed: int, mut: float = 0.08,
                     gap_frac: float = 0.1) -> np.ndarray:
    """Codes evolved down a random binary tree (JC-like substitutions), so that the
    MSA carries phylogenetic signal and the argmax decisions are not near-ties."""
    rng = np.random.default_rng(seed)
    out = np.empty((batch, taxa, sites), dtype=np.uint8)
    for b in range(batch):
        seqs = [rng.integers(0, 4, size=sites, dtype=np.uint8)]
        while len(seqs) < taxa:
            k = int(rng.integers(0, len(seqs)))
            parent = seqs.pop(k)
            for _ in range(2):
                child = parent.copy()
                hit = rng.random(sites) < mut * (0.5 + rng.random())
                child[hit] = rng.integers(0, 4, size=int(hit.sum()), dtype=np.uint8)
                seqs.append(child)
        order = rng.permutation(taxa)
        m = np.stack([seqs[i] for i in order])
        if gap_frac > 0:
            m[rng.random((taxa, sites)) < gap_frac] = 4
        out[b] = m
    return out


def codes_to_onehot(codes: np.ndarray) -> np.ndarray:
    return CODE_TO_ONEHOT[codes]


def onehot_to_codes(onehot: np.ndarray) -> np.ndarray:
    """Inverse of codes_to_onehot for the six vectors the reference can produce."""
    oh = np.asarray(onehot)
    s = oh.sum(-1)
    codes = np.where(s == 4, 4, np.where(s == 0, 5, oh.argmax(-1))).astype(np.uint8)
    if not np.array_equal(CODE_TO_ONEHOT[codes], oh.astype(np.int8)):
        raise ValueError("input is not one of the six site vectors of the reference's CHARS_DICT")
    return codes


def codes_to_seqs(codes2d: np.ndarray):
    return ["".join(CODE_CHARS[c] for c in row) for row in codes2d]


def seqs_to_codes(seqs) -> np.ndarray:
    return np.array([[_CHAR_TO_CODE[ch] for ch in s] for s in seqs], dtype=np.uint8)
