"""PhyInferEnv: host mirror of the environment surface the Argmax path touches
(reference environment.py:444-898).  Tree bookkeeping is re-stated minimally: a subtree
is a small object with the attributes the reference's callers read (seq_indices,
left_tree_data/right_tree_data, name, log_score); Newick / topology strings follow
reference environment.py:263-304 exactly (", " separators, 0.12345 dummy lengths)."""
from __future__ import annotations

import itertools
from typing import List

import numpy as np
import torch

DUMMY_BRANCH = 0.12345          # reference environment.py:288
NO_SCORE = -111111              # reference environment.py:686

CHARACTERS_MAPS = {
    "DNA_WITH_GAP": {"A": [1., 0., 0., 0.], "C": [0., 1., 0., 0.], "G": [0., 0., 1., 0.],
                     "T": [0., 0., 0., 1.], "-": [1., 1., 1., 1.], "N": [1., 1., 1., 1.]},
}
CHARACTERS_MAPS["DNA"] = {k: v for k, v in CHARACTERS_MAPS["DNA_WITH_GAP"].items() if k != "-"}


class PhyloTree:
    """Rooted binary subtree; children ordered by their smallest leaf index
    (reference environment.py:79-83)."""

    def __init__(self, at_root, left_tree_data=None, right_tree_data=None, root_seq_data=None, name="",
                 device="cpu"):
        self.at_root = at_root
        self.name = name
        self.log_score = None
        if root_seq_data is not None:
            self.left_tree_data = self.right_tree_data = None
            self.seq_indices = list(root_seq_data)
            self.log_score = 0
        else:
            if left_tree_data["tree"].seq_indices[0] > right_tree_data["tree"].seq_indices[0]:
                left_tree_data, right_tree_data = right_tree_data, left_tree_data
            self.left_tree_data, self.right_tree_data = left_tree_data, right_tree_data
            self.seq_indices = sorted(left_tree_data["tree"].seq_indices + right_tree_data["tree"].seq_indices)
        self.min_seq_index = self.seq_indices[0]

    @property
    def is_leaf(self):
        return self.left_tree_data is None

    @property
    def is_internal(self):
        return self.left_tree_data is not None


class UnrootedPhyloTree(PhyloTree):
    """Final tree; same children, plus topo_repr (reference environment.py:203-221)."""

    def __init__(self, rooted: PhyloTree):
        self.at_root = True
        self.name = rooted.name
        self.left_tree_data, self.right_tree_data = rooted.left_tree_data, rooted.right_tree_data
        self.seq_indices = rooted.seq_indices
        self.min_seq_index = rooted.min_seq_index
        self.log_score = rooted.log_score
        self.topo_repr = format_rtree_topology(self, True, None)
        self._keys = None
        self._tuple = None

    # the nested (left, len, right, len) tuples (what raxmlpy.treestr_to_tuples yields in the reference) are built
    # when somebody reads them: evaluate_loglikelihood returns the best tree's only, and building them for every tree
    # of a batch of 256 was a tenth of the host time of a rollout
    @property
    def rtree_op_tuple(self):
        if self._tuple is None and self._keys is not None:
            self._tuple = _newick_tuple(self, self._keys)
        return self._tuple

    @rtree_op_tuple.setter
    def rtree_op_tuple(self, v):
        self._tuple = v

    utree_op_tuple = rtree_op_tuple


def format_rtree_topology(tree, at_root=False, sequence_keys=None):
    if tree.left_tree_data is None:
        k = sequence_keys[tree.seq_indices[0]] if sequence_keys else tree.seq_indices[0]
        return f"{k}"
    left = format_rtree_topology(tree.left_tree_data["tree"], False, sequence_keys)
    right = format_rtree_topology(tree.right_tree_data["tree"], False, sequence_keys)
    return f"({left}, {right});" if at_root else f"({left}, {right})"


def format_rtree(tree, at_root=False, branch_length=None, sequence_keys=None):
    b = branch_length if branch_length is not None else DUMMY_BRANCH
    if tree.left_tree_data is None:
        return f"{sequence_keys[tree.seq_indices[0]]}:{b}"
    left = format_rtree(tree.left_tree_data["tree"], False, tree.left_tree_data["branch_length"], sequence_keys)
    right = format_rtree(tree.right_tree_data["tree"], False, tree.right_tree_data["branch_length"], sequence_keys)
    return f"({left}, {right});" if at_root else f"({left}, {right}):{b}"


class PhylogeneticTreeState:
    def __init__(self, subtrees: List[PhyloTree], is_initial=None):
        self.subtrees = subtrees
        self.num_trees = len(subtrees)
        self.is_done = self.num_trees == 1
        self.last_state = isinstance(subtrees[0], UnrootedPhyloTree)
        self.log_score = subtrees[0].log_score if self.last_state else None
        # fixed at construction like the reference's flag (environment.py:334-338).  The callers on the hot path say
        # what they build (init_states: True, a merge: False) -- the walk over the subtrees the reference does per state
        # was half of env.step's host time at a batch of 256; anyone else gets the walk.
        if is_initial is None:
            is_initial = all(t.left_tree_data is None for t in subtrees) and not self.last_state
        self.is_initial = bool(is_initial)


class PhyInferEnv:
    def __init__(self, cfg, device):
        self.device = device
        self.chars_dict = CHARACTERS_MAPS[cfg.env.sequence_type]
        self.states = None
        self.state_tensor = None
        self.init_state_tensor = None
        self.label_trees = None
        self.batch_action_set_step = []

    # ------------------------------------------------------------------ reference API
    def init_states(self, batch_seqs, seq_keys, seq_arrays, label_trees=None, step_action=False):
        if label_trees is not None:
            raise NotImplementedError("label trees (supervised training) are outside the Argmax hot path")
        self.batch_seqs = batch_seqs
        self.seq_keys = seq_keys
        T = len(batch_seqs[0])
        self.tree_pairs_dict, self.action_indices_dict = {}, {}
        for n in range(2, T + 1):
            pairs = list(itertools.combinations(range(n), 2))
            self.tree_pairs_dict[n] = pairs
            self.action_indices_dict[n] = {p: i for i, p in enumerate(pairs)}
        self.batch_size = len(batch_seqs)
        self.states = [
            PhylogeneticTreeState([PhyloTree(False, root_seq_data=[i], name=seq_keys[b][i], device=self.device)
                                   for i in range(len(batch_seqs[b]))], is_initial=True)
            for b in range(self.batch_size)]
        self.init_state_tensor = seq_arrays
        self.state_tensor = None
        self._merge_log = []                     # (i, j) of every step: the merge lists branch_optimize=True scores

    def _merge_host(self, b, i, j, lengths=None, log_score=None):
        """Tree half of one merge for batch element b; returns True when the tree is complete.
        lengths: (edge to row i, edge to row j) from the GPU branch-length optimiser, or None (dummy lengths)."""
        st = self.states[b]
        unrooted = st.num_trees == 2
        li, lj = (None, None) if lengths is None else (float(lengths[0]), float(lengths[1]))
        new_tree = PhyloTree(unrooted, {"tree": st.subtrees[i], "branch_length": li},
                             {"tree": st.subtrees[j], "branch_length": lj})
        if unrooted:
            # optimize_branch_length_no_br (reference environment.py:674-686): dummy lengths, sentinel score -- or the
            # lengths and log-likelihood of the GPU optimiser (the reference's optimize_branch_length_* :625-672)
            keys = self.seq_keys[b]
            new_tree.log_score = NO_SCORE if log_score is None else float(log_score)
            ut = UnrootedPhyloTree(new_tree)
            ut.utree_op_str = format_rtree(new_tree, True, None, keys)
            ut._keys = keys                                    # rtree_op_tuple / utree_op_tuple: built on first read
            self.states[b] = PhylogeneticTreeState([ut], is_initial=False)
            return True
        # a new list: a state object the caller kept (the first state of a trajectory) is not changed behind its back
        trees = list(st.subtrees)
        trees[i] = new_tree
        trees.pop(j)
        self.states[b] = PhylogeneticTreeState(trees, is_initial=False)
        return False

    def step(self, actions, edge_actions=None, parallel=True, branch_optimize=False, agent=None,
             step_action=False):
        n = self.states[0].num_trees
        pairs = self.tree_pairs_dict[n]
        acts = actions.tolist() if hasattr(actions, "tolist") else list(actions)
        ij = [pairs[int(a)] for a in acts]
        self._merge_log.append(ij)
        done = n == 2
        if not done:
            # The tensor half (reference environment.py:760-835) is queued on the device FIRST: it needs the pairs only,
            # and the tree half below (Python objects, about a millisecond per step at a batch of 256) then runs on the
            # host while the device works -- the reference does the two in the other order and its GPU waits.
            self._step_device(ij, n, agent)
        for b, (i, j) in enumerate(ij):
            self._merge_host(b, i, j)
        if done and branch_optimize:
            # The reference scores the finished trees here with raxml-ng (optimize_branch_length_*, environment.py:
            # 625-672: branch lengths optimised, log-likelihood under GTR+I+G).  Same place, on the GPU: the merge lists
            # of the batch go through nnj_tree_optimize (likelihood.py; default model parameters, not optimised) and
            # the trees are rebuilt with their branch lengths and scores.
            self._score_finished_trees(agent)
        return done

    def _step_device(self, ij, n, agent):
        """env.step's tensor half: rows i and j of every alignment aggregated into position i, position j dropped."""
        if agent is None:
            raise NotImplementedError("the mean-aggregate fallback (agent=None) is not part of the hot path")
        from . import utils
        dev = self.state_tensor.device
        ij_t = utils.upload(ij, torch.long, dev)              # no wait for the queued device work (utils.upload)
        if hasattr(agent, "_wants_grad") and agent._wants_grad():
            # Finetune mode: the same step with gradients (train_model.env_step: differentiable gathers,
            # aggregate, concatenation)
            from . import train_model
            self.state_tensor = train_model.env_step(agent, self.state_tensor, ij_t)
            return
        if hasattr(agent, "_context") and getattr(agent, "batch_input", None) is self.state_tensor:
            # this package's PhyloATTN: aggregate + compaction as ONE device call (nnj_env_step).  When the state
            # is the tensor the preceding decode_zxr scored, the library continues its session: merged row in
            # place, no row transformed again, one gather for the dense tensor returned here (include/nnj.h,
            # "Sessions") -- instead of aggregate + cat + gather, i.e. two copies of the whole state per step
            fused = agent.fused_env_step(self.state_tensor, ij_t) if hasattr(agent, "fused_env_step") else None
            self.state_tensor = fused if fused is not None else agent._context().env_step(self.state_tensor, ij_t)
            return
        new = agent.aggregate(None, None, (ij_t[:, 0], ij_t[:, 1]), batchwise_ij_indices=True)
        base = []
        for (i, j) in ij:
            idx = list(range(n))
            idx[i] = n
            idx.pop(j)
            base.append(idx)
        base = utils.upload(base, torch.long, dev)
        cat = torch.cat((self.state_tensor, new), dim=1)
        self.state_tensor = torch.gather(cat, 1, base[:, :, None, None].expand(-1, -1, cat.size(2), cat.size(3)))

    def _score_finished_trees(self, agent):
        if agent is None or not hasattr(agent, "_context"):
            raise NotImplementedError("branch_optimize=True scores the trees on the GPU and needs this package's agent")
        from . import likelihood as lk
        ctx = agent._context()
        arr = self.init_state_tensor
        if arr is None:
            raise RuntimeError("branch_optimize=True needs the alignment passed to init_states")
        codes = arr if arr.dim() == 3 else agent.onehot_to_codes(arr.to(ctx.device))
        merges = torch.tensor(self._merge_log, dtype=torch.int32).permute(1, 0, 2).contiguous()      # [B, T-1, 2]
        ll = torch.empty(merges.shape[0], dtype=torch.float64)
        br = torch.empty(tuple(merges.shape[:2]) + (2,), dtype=torch.float32)
        if all(torch.equal(codes[0], codes[b]) for b in range(1, codes.shape[0])):
            l, r = lk.tree_optimize(ctx, codes[:1], merges)          # replicas of one alignment (Search / Finetune)
            ll, br = l.cpu(), r.cpu()
        else:
            for b in range(merges.shape[0]):
                l, r = lk.tree_optimize(ctx, codes[b:b + 1], merges[b:b + 1])
                ll[b], br[b] = l[0].cpu(), r[0].cpu()
        seqs, keys, log = self.batch_seqs, self.seq_keys, self._merge_log
        self.init_states(seqs, keys, arr)
        self._merge_log = log
        self.apply_merges(merges.numpy(), br.numpy(), ll.numpy())

    def apply_merges(self, merges, brlen=None, log_scores=None):
        """Fast path: replay a device-produced merge list [B,T-1,2] on the host trees; brlen [B,T-1,2] and
        log_scores [B] (neuralnj_amd.likelihood.tree_optimize) give the trees real branch lengths and scores."""
        merges = np.asarray(merges)
        for b in range(merges.shape[0]):
            for s, (i, j) in enumerate(merges[b]):
                self._merge_host(b, int(i), int(j), None if brlen is None else brlen[b][s],
                                 None if log_scores is None else log_scores[b])

    def evaluate_loglikelihood(self, get_all_tree=False):
        scores = [s.log_score for s in self.states]
        assert all(s.is_done for s in self.states)
        t = torch.from_numpy(np.array(scores))
        if get_all_tree:
            return (t, [s.subtrees[0].rtree_op_tuple for s in self.states],
                    [s.subtrees[0].utree_op_tuple for s in self.states],
                    [s.subtrees[0].utree_op_str for s in self.states])
        best = self.states[scores.index(max(scores))].subtrees[0]
        return t, best.rtree_op_tuple, best.utree_op_tuple, best.utree_op_str

    def dump_end_trees(self):
        trees = [s.subtrees[0] for s in self.states]
        return trees, [t.log_score for t in trees]

    def get_current_trees(self):
        return [[format_rtree_topology(t, True, None) for t in s.subtrees] for s in self.states]


def _newick_tuple(tree, keys):
    """Nested (left, len, right, len) tuples, the shape raxmlpy.treestr_to_tuples yields for
    the dummy-length Newick (reference RAxMLpy/raxmlpy/core.py:21-26)."""
    if tree.left_tree_data is None:
        return keys[tree.seq_indices[0]]
    return (_newick_tuple(tree.left_tree_data["tree"], keys), DUMMY_BRANCH,
            _newick_tuple(tree.right_tree_data["tree"], keys), DUMMY_BRANCH)


def compute_raw_tree_log_score(env, rtree_str_batch, parallel=False, agent=None):
    """Reference environment.py:394-441: optimise the branch lengths of the given Newick trees of env's alignments and
    return their log-likelihoods -- there one raxml-ng call per tree (`pllpy.optimize_brlen(..., iters=3)`), here the
    trees become merge lists (utils.newick_to_merges) and go through nnj_tree_optimize on the GPU.  `agent`: this
    package's PhyloATTN (its device context); the reference's signature has no such argument because its scorer is a CPU
    library."""
    from . import likelihood as lk
    from . import utils
    from .phydata import seqs_to_codes
    if agent is None or not hasattr(agent, "_context"):
        raise NotImplementedError("compute_raw_tree_log_score scores on the GPU: pass agent=<neuralnj_amd PhyloATTN>")
    ctx = agent._context()
    out = []
    for b, tree in enumerate(rtree_str_batch):
        merges, br = utils.newick_to_merges(tree, env.seq_keys[b])
        codes = torch.from_numpy(seqs_to_codes(env.batch_seqs[b])[None])
        ll, _ = lk.tree_optimize(ctx, codes, torch.from_numpy(merges[None]), torch.from_numpy(br[None]))
        out.append(float(ll[0]))
    return out
