"""MSA readers for single-instance inference, mirroring the part of the reference's
phydata.py the Argmax driver calls: PHYLIP (sequential / interleaved) and FASTA readers
with the reference's character policy (phydata.py:499-548, 593-630), taxon ordering by
numeric suffix and the batch dictionary of load_pi_instance (phydata.py:1249-1289,
1222-1247).  Output `data` is the reference's int8 one-hot [1,T,L,4]; `codes` is the
1-byte form the device consumes."""
from __future__ import annotations

import re

import numpy as np
import torch

from .synth import CODE_TO_ONEHOT, seqs_to_codes

ALPHABET = "ACGT-N*"          # keys of the reference's CHARS_DICT (phydata.py:38-46)


def _clean(seq: str) -> str:
    """Upper-case; every symbol outside the alphabet becomes a gap (phydata.py:536-543)."""
    s = seq.replace(" ", "").upper()
    return "".join(ch if ch in ALPHABET else "-" for ch in s)


def load_phy_file_multirow(path):
    """PHYLIP, sequential or interleaved: first block has 'name sequence' lines, later
    blocks (after a blank line) continue the sequences in the same order."""
    with open(path, "r") as f:
        header = f.readline().split()
        n_taxa, n_sites = int(header[0]), int(header[1])
        names, chunks = [], {}
        lines = [ln.strip() for ln in f]
    it = iter(lines)
    for ln in it:                       # first block
        if not ln:
            break
        parts = ln.split(maxsplit=1)
        if len(parts) > 1:
            names.append(parts[0])
            chunks[parts[0]] = [parts[1]]
    k = 0
    for ln in it:                       # continuation blocks
        if ln:
            chunks[names[k]].append(ln)
            k += 1
        else:
            k = 0
    seqs = [_clean("".join(chunks[nm])) for nm in names]
    for nm, s in zip(names, seqs):
        if len(s) != n_sites:
            raise ValueError(f"sequence {nm} has {len(s)} sites, header says {n_sites}")
    if len(names) != n_taxa:
        raise ValueError(f"found {len(names)} taxa, header says {n_taxa}")
    return seqs, names, n_taxa, n_sites


def load_alignment_file(path):
    """FASTA (.fasta / .aln)."""
    names, parts = [], {}
    cur = None
    with open(path, "r") as f:
        for ln in f:
            ln = ln.strip()
            if not ln:
                continue
            if ln.startswith(">"):
                cur = ln[1:].split()[0]
                names.append(cur)
                parts[cur] = []
            elif cur is not None:
                parts[cur].append(ln)
    seqs = [_clean("".join(parts[nm])) for nm in names]
    n_sites = len(seqs[0]) if seqs else 0
    for nm, s in zip(names, seqs):
        if len(s) != n_sites:
            raise ValueError(f"sequence {nm} length differs from the first sequence")
    return seqs, names, len(names), n_sites


def load_pi_instance(path):
    """One MSA -> the batch dict consumed by the rollout (keys as the reference's
    infer_custom_collate_fn, plus 'codes')."""
    if path.endswith(".phy"):
        seqs, keys, n_taxa, n_sites = load_phy_file_multirow(path)
        m = re.match(r"^([a-zA-Z]+)([0-9]+)$", keys[0])
        if m is None:
            raise ValueError(f"taxon name {keys[0]!r} is not <letters><number> (reference phydata.py:1252-1256)")
        plen = len(m.group(1))
        order = {k: int(k[plen:]) - 1 for k in keys}
        s_sorted, k_sorted = [None] * n_taxa, [None] * n_taxa
        for s, k in zip(seqs, keys):
            s_sorted[order[k]], k_sorted[order[k]] = s, k
        seqs, keys = s_sorted, k_sorted
    elif path.endswith(".fasta") or path.endswith(".aln"):
        seqs, keys, n_taxa, n_sites = load_alignment_file(path)
    else:
        raise ValueError("expected a .phy, .fasta or .aln file")
    # only_padding_sample (phydata.py:98-123): all sequences already have equal length, so no
    # column is padded or sampled and every weight is 1
    codes = seqs_to_codes(seqs)[None]                      # [1,T,L] uint8
    data = CODE_TO_ONEHOT[codes]                           # int8 [1,T,L,4]
    weights = np.ones((1, n_sites), dtype=np.float32)
    return {
        "data": torch.from_numpy(data), "codes": torch.from_numpy(codes),
        "seqs": [seqs], "seq_keys": [keys], "seq_weights": torch.from_numpy(weights),
        "file_paths": [path], "taxa_nums": [n_taxa], "seq_lens": [n_sites],
    }


_PHY_UNKNOWN = str.maketrans({"?": "N", ".": "N"})


def load_phy_file(file_path):
    """Strictly sequential PHYLIP -- every record on ONE line, `name` then the sequence (blanks inside it allowed) --
    with the character policy of the reference's reader of that format (phydata.py:478-496): upper case, '?' and '.'
    read as N, nothing else rewritten (load_phy_file_multirow, which the drivers use, maps every symbol outside the
    alphabet to a gap instead).  A taxon name that occurs again replaces the earlier record but keeps its place.
    Raises AssertionError, like the reference, when the header's counts do not match the records."""
    order, seq_of = [], {}
    with open(file_path, "r") as f:
        want_taxa, want_sites = (int(tok) for tok in f.readline().split()[:2])
        for record in f:
            fields = record.split()
            if not fields:
                continue
            name = fields[0]
            if name not in seq_of:
                order.append(name)
            seq_of[name] = "".join(fields[1:]).upper().translate(_PHY_UNKNOWN)
    seqs = [seq_of[name] for name in order]
    assert len(order) == want_taxa, f"{file_path}: header says {want_taxa} taxa, file holds {len(order)}"
    assert seqs and len(seqs[0]) == want_sites, f"{file_path}: header says {want_sites} sites"
    return seqs, order, want_taxa, want_sites


def load_tree_file(file_path, device=None, pos=5):
    """Reference phydata.py:633-: a label tree for SUPERVISED training (Bio.Phylo -> PhyloTree with inner-node names);
    the inference and Finetune drivers import the name and never call it.  Here: the Newick string of the file; turn
    it into a merge list with utils.newick_to_merges to score it (environment.compute_raw_tree_log_score)."""
    with open(file_path, "r") as f:
        return f.read().strip()


class PhySampler:                      # reference phydata.py:442: batch sampler of the supervised training set
    def __init__(self, *a, **k):
        raise NotImplementedError("PhySampler belongs to supervised training (train.py), which is not part of this package")


def custom_collate_fn(batch):          # reference phydata.py:1031: collate of the supervised training set
    raise NotImplementedError("custom_collate_fn belongs to supervised training (train.py), which is not part of this package")
