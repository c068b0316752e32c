"""Host-side logic (CPU): index map, config, tree strings, loaders, RF distance."""
import itertools
import os
import tempfile

import numpy as np
import pytest

from helpers import golden_names, load_golden
from neuralnj_amd import environment, phydata, synth, utils, weights


def _ref_index_map(n, ip, jp):
    """Dictionary form of the reference's branch table (utils.py:227-247), written
    independently of utils.index_map_one for cross-checking."""
    prev = {p: k for k, p in enumerate(itertools.combinations(range(n + 1), 2))}
    out = []
    for (ii, jj) in itertools.combinations(range(n), 2):
        if ii == ip:
            out.append(len(prev) + jj)
        elif jj == ip:
            out.append(len(prev) + ii)
        else:
            out.append(prev[(ii + (ii >= jp), jj + (jj >= jp))])
    return out


@pytest.mark.parametrize("n", [2, 3, 4, 7, 19, 49])
def test_index_map_all_merges(n):
    from oracle_lib import Oracle  # noqa: F401  (C restatement, literal branch table)
    import ctypes as C
    from oracle_lib import _load
    lib = _load("libnnj_oracle.so")
    for ip, jp in itertools.combinations(range(n + 1), 2):
        a = utils.index_map_one(n, ip, jp)
        assert a.tolist() == _ref_index_map(n, ip, jp)
        buf = np.empty(n * (n - 1) // 2, np.int64)
        lib.nnjo_index_map_one(C.c_int32(n), C.c_int32(ip), C.c_int32(jp), buf.ctypes.data_as(C.c_void_p))
        assert buf.tolist() == a.tolist()
    got = utils.get_score_indices_to_prev(np.array([[0, 1], [1, n]]), None, n, 2)
    assert np.asarray(got).dtype == np.int64 and np.asarray(got).shape == (2, n * (n - 1) // 2)
    assert list(got[0]) == utils.index_map_one(n, 0, 1).tolist() and list(got[1]) == utils.index_map_one(n, 1, n).tolist()


def test_config_defaults_and_yaml():
    c = utils.empty_config()
    assert (c.model.patch_size, c.model.embed_dim, c.model.num_enc_heads, c.model.num_enc_layers) == (4, 32, 4, 3)
    assert c.env.batch_size == 8 and c.env.sequence_type == "DNA_WITH_GAP"
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.write("env:\n  batch_size: 1\nmodel:\n  patch_size: 1\n  embed_dim: 64\n  num_enc_heads: 8\n  num_enc_layers: 6\n")
    c.merge_from_file(f.name)
    os.unlink(f.name)
    assert c.model.embed_dim == 64 and c.model.vocab_size == 4 and c.env.batch_size == 1
    assert weights.num_params(c) == 425857
    assert len(weights.param_spec(c)) == 172


def test_weights_pack_roundtrip_and_determinism():
    c = utils.shipped_config()
    a = weights.pack(c, weights.seeded_state(c, 3))
    b = weights.pack(c, weights.seeded_state(c, 3))
    assert weights.digest(a) == weights.digest(b)
    assert weights.digest(a) != weights.digest(weights.pack(c, weights.seeded_state(c, 4)))
    st = weights.unpack(c, a)
    assert np.array_equal(weights.pack(c, st), a)
    with pytest.raises(ValueError):
        weights.unpack(c, a[:-1])


@pytest.mark.parametrize("name", golden_names())
def test_tree_strings_match_reference(name):
    """Replaying the reference's merge list through the host trees reproduces its Newick
    and topology strings (pins child ordering, separators, dummy branch lengths)."""
    z, cfgs, _ = load_golden(name)
    B, T, L = z["codes"].shape
    env = environment.PhyInferEnv(cfgs, "cpu")
    keys = [list(k) for k in z["keys"]]
    env.init_states([synth.codes_to_seqs(z["codes"][b]) for b in range(B)], keys, None)
    assert env.tree_pairs_dict[T][0] == (0, 1) and len(env.tree_pairs_dict[T]) == T * (T - 1) // 2
    env.apply_merges(z["merges"])
    for b in range(B):
        t = env.states[b].subtrees[0]
        assert t.utree_op_str == str(z["newick"][b])
        assert t.topo_repr == str(z["topo"][b])
    scores, _, _, best = env.evaluate_loglikelihood()
    assert best == str(z["best_tree"]) and float(scores[0]) == -111111


def test_env_step_host_half_matches_apply_merges():
    z, cfgs, _ = load_golden("synth_b2_t8_l128_s0")
    B, T, L = z["codes"].shape
    env = environment.PhyInferEnv(cfgs, "cpu")
    env.init_states([synth.codes_to_seqs(z["codes"][b]) for b in range(B)], [list(k) for k in z["keys"]], None)
    for step in range(T - 1):
        n = T - step
        for b in range(B):
            i, j = z["merges"][b, step]
            done = env._merge_host(b, int(i), int(j))
        assert done == (n == 2)
        assert env.get_current_trees()[0] is not None
    assert env.states[1].subtrees[0].topo_repr == str(z["topo"][1])


def test_phylip_and_fasta_readers():
    seqs = ["ACGT-NAC", "acgtKkGG", "TTTTAAAA"]
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.phy")
        with open(p, "w") as f:       # interleaved, taxa out of order, lower case + foreign symbol
            f.write("3 8\ntaxon3   TTTT\ntaxon1   ACGT\ntaxon2   acgt\n\nAAAA\n-NAC\nKkGG\n")
        b = phydata.load_pi_instance(p)
        assert b["seq_keys"] == [["taxon1", "taxon2", "taxon3"]]
        assert b["seqs"][0] == ["ACGT-NAC", "ACGT--GG", "TTTTAAAA"]
        assert b["data"].shape == (1, 3, 8, 4) and b["data"].dtype.is_floating_point is False
        assert b["data"][0, 0, 4].tolist() == [1, 1, 1, 1] and b["data"][0, 0, 0].tolist() == [1, 0, 0, 0]
        assert np.array_equal(synth.onehot_to_codes(b["data"].numpy()), b["codes"].numpy())
        assert float(b["seq_weights"].sum()) == 8
        q = os.path.join(d, "y.fasta")
        with open(q, "w") as f:
            f.write(">b desc\nACGT\n-NAC\n>a\nTTTT\nAAAA\n")
        c = phydata.load_pi_instance(q)
        assert c["seq_keys"] == [["b", "a"]] and c["seqs"][0] == ["ACGT-NAC", "TTTTAAAA"]
        with open(p, "w") as f:
            f.write("2 4\ntaxon1 ACGT\ntaxon2 ACG\n")
        with pytest.raises(ValueError):
            phydata.load_pi_instance(p)
    with pytest.raises(ValueError):
        synth.onehot_to_codes(np.array([[[[1, 1, 0, 0]]]], dtype=np.int8))


def test_example_files_match_reference_loader_fixture(repo_root):
    """The two example MSAs shipped with the reference, read by our loader, give the codes and
    taxon order its own loader produced (fixture); skipped where /root/reference is absent."""
    ex = "/root/reference/examples/len1024taxa50"
    if not os.path.isdir(ex):
        pytest.skip("reference checkout not present on this machine")
    for f in sorted(os.listdir(ex)):
        if f.endswith(".phy"):
            z, _, _ = load_golden("example_" + f[:-4].replace(".", "p"))
            b = phydata.load_pi_instance(os.path.join(ex, f))
            assert np.array_equal(b["codes"].numpy(), z["codes"]) and b["seq_keys"][0] == list(z["keys"][0])


def test_rf_distance():
    a = "((A:1, B:1):1, (C:1, D:1):1, E:1);"
    b = "((A:1, C:1):1, (B:1, D:1):1, E:1);"
    assert utils.rf_distance(a, a) == (0, 0.0)
    rf, nrf = utils.rf_distance(a, b)
    assert rf == 4 and nrf == 1.0
    z, cfgs, _ = load_golden("synth_b1_t20_l256_s0")
    s = str(z["newick"][0])
    assert utils.rf_distance(s, s)[0] == 0
    # rooted binary vs its unrooted reading: same splits
    assert utils.rf_distance("((A, B), (C, (D, E)));", "(A, B, (C, (D, E)));")[0] == 0


def test_newick_to_merges_round_trip():
    """utils.newick_to_merges turns any binary Newick tree (trifurcating root included) into a merge list of the NJ
    loop's convention: replayed through the environment it gives the same topology and the same branch lengths."""
    from neuralnj_amd.environment import PhyInferEnv
    cfgs = utils.shipped_config()
    rng = np.random.default_rng(0)
    for _ in range(25):
        T = int(rng.integers(3, 14))
        keys = [f"taxon{i + 1}" for i in range(T)]
        merges = [tuple(sorted(rng.choice(n, size=2, replace=False))) for n in range(T, 1, -1)]
        br = rng.random((1, T - 1, 2)).astype(np.float32) + 0.01
        env = PhyInferEnv(cfgs, "cpu")
        env.init_states([[""] * T], [keys], None)
        env.apply_merges(np.array([merges]), br, np.array([-1.0]))
        nw = env.states[0].subtrees[0].utree_op_str
        m2, b2 = utils.newick_to_merges(nw, keys)
        env2 = PhyInferEnv(cfgs, "cpu")
        env2.init_states([[""] * T], [keys], None)
        env2.apply_merges(m2[None], b2[None], np.array([-1.0]))
        assert utils.rf_distance(nw, env2.states[0].subtrees[0].utree_op_str)[0] == 0
        assert sorted(np.round(br.reshape(-1), 5).tolist()) == sorted(np.round(b2.reshape(-1), 5).tolist())
    m, b = utils.newick_to_merges("(taxon1:0.1,taxon2:0.2,(taxon3:0.3,taxon4:0.4):0.5);", [f"taxon{i + 1}" for i in range(4)])
    assert m.tolist() == [[2, 3], [1, 2], [0, 1]]
    with pytest.raises(ValueError):
        utils.newick_to_merges("(taxon1,taxon2,taxon3,taxon4);", [f"taxon{i + 1}" for i in range(4)])


def test_reference_driver_import_lines_resolve():
    """The import lines of the reference's driver (finetune_rl_search.py:13-28) against the drop-in module names of
    neuralnj_amd/compat (in a process of its own: the names `model`, `utils` ... are too generic to put on this one's
    path)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from environment import PhyInferEnv, compute_raw_tree_log_score\n"
        "import environment\n"
        "from model import PhyloATTN as PGPI\n"
        "import utils\n"
        "from phydata import PhySampler, custom_collate_fn, load_pi_instance, load_tree_file, load_phy_file_multirow, load_phy_file\n"
        "c = utils.empty_config(); utils.set_evolution_model('GTR+I+G'); utils.ReplayBuffer(4); utils.get_score_indices_to_prev\n"
        "assert len(PGPI(utils.shipped_config()).state_dict()) == 172\n"
        "print('ok')\n") % (repo, os.path.join(repo, "neuralnj_amd", "compat"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_load_phy_file_sequential_policy(tmp_path):
    """phydata.load_phy_file: one record per line, '?' and '.' read as N, upper case, a repeated name replaces the
    earlier record in place, header mismatches raise AssertionError (reference phydata.py:478-496)."""
    from neuralnj_amd import phydata
    p = tmp_path / "a.phy"
    p.write_text("3 8\nt1 acgt ac?.\nt2 ACGTAC-N\n\nt3 TTTTTTTT\nt2 GGGGGGGG\n")
    seqs, keys, n, L = phydata.load_phy_file(str(p))
    assert (keys, n, L) == (["t1", "t2", "t3"], 3, 8)
    assert seqs == ["ACGTACNN", "GGGGGGGG", "TTTTTTTT"]
    p.write_text("4 8\nt1 ACGTACGT\nt2 ACGTACGT\n")
    with pytest.raises(AssertionError):
        phydata.load_phy_file(str(p))


def test_finetune_uniform_streams_never_coincide():
    """SeedSequence zero-pads its entropy, so default_rng([s]) and default_rng([s, 0]) are ONE stream -- rank 0's
    first episode used to replay the baseline rollout's uniforms (advantage exactly 0, no policy gradient).  The baseline
    stream and every rank's episode stream must differ, for every rank and seed, and the ranks among themselves."""
    from neuralnj_amd.rollout import baseline_rng, episode_rng
    for seed in (0, 1, 7):
        u0 = baseline_rng(seed).random(16)
        rows = [episode_rng(seed, r).random((3, 16)) for r in range(4)]
        for r, u in enumerate(rows):
            assert not np.array_equal(u0, u[0]), f"seed {seed}: rank {r}'s first episode replays the baseline"
        for a in range(4):
            for b in range(a + 1, 4):
                assert not np.array_equal(rows[a], rows[b])
        assert np.array_equal(baseline_rng(seed).random(16), u0)          # rank independent and reproducible
