"""Host helpers mirroring the surface of the reference's utils.py that the Argmax
path touches (reference utils.py:7-57 config defaults, 59-63 evolution model,
213-251 score index map, 255-261 RF distance).

`CfgNode` is a small attribute dictionary with the subset of fvcore's CfgNode API the
reference uses (`merge_from_file`, attribute access, nested nodes); fvcore itself is
not a dependency.
"""
from __future__ import annotations

import numpy as np
import yaml


class CfgNode(dict):
    """Attribute-style nested config (the reference uses fvcore.common.config.CfgNode)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                node = self.get(k)
                if not isinstance(node, CfgNode):
                    node = CfgNode()
                    self[k] = node
                node.merge_from_dict(v)
            else:
                self[k] = v
        return self

    def merge_from_file(self, path):
        with open(path, "r") as f:
            d = yaml.safe_load(f) or {}
        return self.merge_from_dict(d)

    def clone(self):
        c = CfgNode()
        for k, v in self.items():
            c[k] = v.clone() if isinstance(v, CfgNode) else v
        return c


def empty_config() -> CfgNode:
    """Same keys and defaults as the reference's utils.empty_config (utils.py:7-57)."""
    cfgs = CfgNode()
    cfgs.num_epoch = 1
    cfgs.num_episodes = 1
    cfgs.num_episodes_baseline = 1
    cfgs.lr = 0.01
    cfgs.clip_value = 0.1
    cfgs.entropy_reg_strength = 1.0
    cfgs.risk_epsilon = 0.1
    cfgs.replay_buffer_size = 128
    cfgs.replay_buffer_sample_size = 32
    cfgs.replay_buffer_score_bound = 10
    cfgs.loss = CfgNode(BALANCED_ELU_LOSS=False, ELU_LOSS=False)
    cfgs.summary_name = "Try"
    cfgs.summary_path = "tb_summary"
    cfgs.checkpoint_path = "checkpoints"
    cfgs.reload_checkpoint_path = ""
    cfgs.dataset_path = ""
    cfgs.val_dataset_path = ""
    cfgs.instance_path = ""
    cfgs.sequences_file = ""
    cfgs.raw_tree_file = ""
    cfgs.c_best_tree_file = ""
    cfgs.dataset_taxa_list = []
    cfgs.dataset_len_list = []
    cfgs.env = CfgNode(batch_size=8, sequence_type="DNA_WITH_GAP")
    cfgs.model = CfgNode(
        vocab_size=4, patch_size=4, fixed_length=1024, embed_dim=32,
        encoder_attn_layers=2, num_enc_heads=4, num_enc_layers=3,
    )
    cfgs.ratio_factor = 1.0
    return cfgs


def shipped_config() -> CfgNode:
    """Model hyper-parameters of the reference's shipped yaml
    (config/finetune_reinforce_search_example.yaml:24-30)."""
    cfgs = empty_config()
    cfgs.model.merge_from_dict(dict(vocab_size=4, patch_size=1, embed_dim=64,
                                    num_enc_heads=8, num_enc_layers=6))
    cfgs.env.batch_size = 1
    return cfgs


evolution_model = "GTR+I+G"


def set_evolution_model(name: str) -> None:
    global evolution_model
    evolution_model = name


def num_pairs(n: int) -> int:
    return n * (n - 1) // 2


def pair_index(n: int, i, j):
    """Flat index of pair (i,j), i<j, in itertools.combinations(range(n), 2) order
    (reference environment.py:457-462).  Works on ints and numpy arrays."""
    return i * n - i * (i + 1) // 2 + (j - i - 1)


def index_map_one(n: int, ii_prev: int, jj_prev: int) -> np.ndarray:
    """Old->new flat pair index map for ONE batch element, vectorised restatement of
    the branch table of reference utils.py:227-247.  `n` is the current number of rows
    (nb_seq); the result indexes cat(logits_prev[P(n+1)], new_scores[n])."""
    ii, jj = np.triu_indices(n, 1)
    ii = ii.astype(np.int64)
    jj = jj.astype(np.int64)
    lp = num_pairs(n + 1)
    sh_i = ii + (ii >= jj_prev)          # rows at/after the removed slot shift by one
    sh_j = jj + (jj >= jj_prev)
    old = pair_index(n + 1, sh_i, sh_j)
    touches_i = ii == ii_prev            # pair (merged, jj) -> new score of row jj
    touches_j = jj == ii_prev            # pair (ii, merged) -> new score of row ii
    return np.where(touches_i, lp + jj, np.where(touches_j, lp + ii, old))


_TRIU_CACHE = {}


def _triu(n: int):
    t = _TRIU_CACHE.get(n)
    if t is None:
        ii, jj = np.triu_indices(n, 1)
        t = _TRIU_CACHE[n] = (ii.astype(np.int64), jj.astype(np.int64))
    return t


def index_map_batch(n: int, ij_prev: np.ndarray) -> np.ndarray:
    """index_map_one for a whole batch at once: int64 [B, P(n)] (the pair table of n is computed once per n)."""
    ii, jj = _triu(int(n))
    ip = ij_prev[:, 0:1].astype(np.int64)
    jp = ij_prev[:, 1:2].astype(np.int64)
    lp = num_pairs(n + 1)
    old = pair_index(n + 1, ii[None, :] + (ii[None, :] >= jp), jj[None, :] + (jj[None, :] >= jp))
    return np.where(ii[None, :] == ip, lp + jj[None, :], np.where(jj[None, :] == ip, lp + ii[None, :], old))


def get_score_indices_to_prev(actions_ij_prev, env, nb_seq, batch_size):
    """Same signature and result as the reference's utils.get_score_indices_to_prev
    (utils.py:213-251); `env` is accepted for call compatibility and not consulted.  The reference builds the list
    element by element in Python (7 ms per tree at a batch of 256); here the whole batch is one numpy expression."""
    a = actions_ij_prev
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.asarray(a)[:batch_size]
    # an int64 array [B, P(n)] where the reference returns the same numbers as a list of lists: its only consumer
    # (finetune_rl_search.py:123) wraps the result in np.array, and converting 300k indices per step to Python ints
    # cost more than computing them
    return index_map_batch(int(nb_seq), a)


def upload(values, dtype, device):
    """Small host data (per-step index lists) to the device WITHOUT draining the stream: a copy from pageable host
    memory blocks the host until everything queued before it has finished -- once per NJ step that makes the host and the
    device take turns.  Staged through pinned memory (torch's caching host allocator keeps the block until the copy has
    run) the copy is just another entry of the stream."""
    import torch
    t = torch.as_tensor(values, dtype=dtype)
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def newick_to_merges(newick: str, keys):
    """A binary (or root-trifurcating) Newick tree over the taxa `keys` -> (merges int32 [T-1,2], brlen float32
    [T-1,2]) in the convention of the NJ loop (environment.py:764-768: the merged subtree takes position i, position j
    leaves the list; brlen[s] = lengths of the edges above rows i and j of merge s, 0.1 where the string has none).
    Lets a tree that did not come from a rollout (the reference's raw_tree_file / c_best_tree_file, a RAxML result) be
    scored by nnj_tree_loglik / nnj_tree_optimize."""
    s = newick.strip().rstrip(";").replace(" ", "")
    pos = 0

    def parse():
        nonlocal pos
        if s[pos] == "(":
            pos += 1
            kids = [parse()]
            while s[pos] == ",":
                pos += 1
                kids.append(parse())
            assert s[pos] == ")", "unbalanced Newick string"
            pos += 1
            node = kids
        else:
            node = None
        start = pos
        while pos < len(s) and s[pos] not in ",():":
            pos += 1
        label = s[start:pos]
        length = None
        if pos < len(s) and s[pos] == ":":
            pos += 1
            start = pos
            while pos < len(s) and s[pos] not in ",()":
                pos += 1
            length = float(s[start:pos])
        return (node if node is not None else label, length)

    root, _ = parse()
    if not isinstance(root, list):
        raise ValueError("a single-leaf tree has no merges")
    if len(root) == 3:                       # trifurcating (unrooted) root: join the last two first, like the reference
        b, a = root.pop(), root.pop()
        root.append(([a, b], 0.0))
    index = {k: i for i, k in enumerate(keys)}
    live = list(range(len(keys)))            # subtree ids by position; leaves are their taxon index
    merges, brlen = [], []
    next_id = [len(keys)]

    def build(node):
        kids, _ = node
        if not isinstance(kids, list):
            return index[kids]
        if len(kids) != 2:
            raise ValueError("only binary trees (and a trifurcating root) can be turned into a merge list")
        ids = [build(k) for k in kids]
        lens = [0.1 if k[1] is None else float(k[1]) for k in kids]
        pa, pb = live.index(ids[0]), live.index(ids[1])
        (i, li), (j, lj) = sorted(((pa, lens[0]), (pb, lens[1])))
        merges.append((i, j))
        brlen.append((li, lj))
        me = next_id[0]
        next_id[0] += 1
        live[i] = me
        del live[j]
        return me

    build((root, None))
    if len(merges) != len(keys) - 1:
        raise ValueError("the tree does not contain every taxon exactly once")
    return np.array(merges, dtype=np.int32), np.array(brlen, dtype=np.float32)


class ReplayBuffer:
    """Same surface as the reference's utils.ReplayBuffer (utils.py:67-100): the best `replay_buffer_size` distinct
    trees seen so far, distinct by `topo_repr` (the comparison nnj_topology_hash does on the device for whole batches).
    `sample` returns None exactly like the reference's, whose trajectory re-injection is switched off (utils.py:99:
    `return None`), so a rollout that consults the buffer samples every action itself."""

    def __init__(self, replay_buffer_size):
        self.replay_buffer_size = replay_buffer_size
        self.trees, self.scores = [], []

    def get_size(self):
        return len(self.trees)

    def add(self, trees, scores):
        for tree, score in zip(trees, scores):
            if any(t.topo_repr == tree.topo_repr for t in self.trees):
                continue
            if len(self.trees) < self.replay_buffer_size:
                self.trees.append(tree)
                self.scores.append(score)
            else:
                k = self.scores.index(min(self.scores))
                if score > self.scores[k]:
                    self.trees[k], self.scores[k] = tree, score

    def sample(self, sample_num):
        return None


# ---------------------------------------------------------------- RF distance
def _newick_splits(newick: str):
    """Bipartitions (as frozensets of leaf names, normalised against the full leaf
    set) of a Newick string; branch lengths and internal labels are ignored."""
    s = newick.strip()
    if s.endswith(";"):
        s = s[:-1]
    pos = 0
    leaves_all = []
    clades = []

    def parse():
        nonlocal pos
        while pos < len(s) and s[pos].isspace():
            pos += 1
        if s[pos] == "(":
            pos += 1
            members = []
            while True:
                members += parse()
                while pos < len(s) and s[pos].isspace():
                    pos += 1
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    break
                raise ValueError(f"bad newick at {pos}")
            _skip_label()
            clades.append(frozenset(members))
            return members
        start = pos
        while pos < len(s) and s[pos] not in ",():;":
            pos += 1
        name = s[start:pos].strip()
        _skip_label()
        leaves_all.append(name)
        return [name]

    def _skip_label():
        nonlocal pos
        while pos < len(s) and s[pos] not in ",()":
            pos += 1

    parse()
    full = frozenset(leaves_all)
    splits = set()
    for c in clades:
        if 1 < len(c) < len(full) - 1:
            other = full - c
            ref_leaf = min(full)
            splits.add(c if ref_leaf not in c else other)
    return full, splits


def rf_distance(newick_a: str, newick_b: str):
    """Unrooted Robinson-Foulds distance and its normalised form between two Newick
    strings (the reference delegates to ete3, utils.py:255-261)."""
    la, sa = _newick_splits(newick_a)
    lb, sb = _newick_splits(newick_b)
    if la != lb:
        raise ValueError("trees have different leaf sets")
    rf = len(sa ^ sb)
    max_rf = len(sa) + len(sb)
    return rf, (rf / max_rf if max_rf else 0.0)


def calculate_rf_distance(file1, file2):
    def read(x):
        try:
            with open(x, "r") as f:
                return f.readline().strip()
        except (OSError, ValueError):
            return x
    return rf_distance(read(file1), read(file2))
