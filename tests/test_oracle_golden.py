"""Pins the CPU oracle (oracle/nnj_oracle.c) to vectors captured from the reference
itself (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from helpers import (assert_logits_close, golden_names, load_golden, onehot_f32, split_trace)
from oracle_lib import Oracle

SMALL = golden_names(max_taxa=20)
BIG = golden_names(min_taxa=21)


def _check(name, precision="f32", rel=1e-4):
    z, cfgs, packed = load_golden(name)
    o = Oracle(cfgs, packed, precision)
    oh, mask = onehot_f32(z["codes"]), z["mask"]
    B, T, L = z["codes"].shape
    if "sub_embed" in z.files:
        enc, taps = o.encode(oh, mask, taps=True)
        np.testing.assert_allclose(taps[0], z["sub_embed"], atol=2e-6)
        for i, k in enumerate(("sub_l0_row", "sub_l0_col", "sub_l0_ffn")):
            np.testing.assert_allclose(taps[i + 1], z[k].transpose(2, 0, 1, 3), atol=3e-5)
    # teacher-forced: the oracle follows the reference's merges, every table must match
    r = o.rollout_argmax(oh, mask, forced_merges=z["merges"], want_state=True)
    if "enc" in z.files:
        np.testing.assert_allclose(r["state"], z["enc"], atol=1e-4 * np.abs(z["enc"]).max())
    else:
        np.testing.assert_allclose(r["state"][:, ::7, ::61, :], z["enc_slice"],
                                   atol=1e-4 * np.abs(z["enc_slice"]).max())
        assert abs(r["state"].astype(np.float64).sum() - float(z["enc_checksum"])) \
            <= 1e-5 * float(z["enc_abs_checksum"])
    assert_logits_close(r["logits"], z["logits"], rel=rel)
    # argmax of each table: must agree wherever the reference's own top-2 gap is decisive
    scale = np.abs(z["logits"]).max()
    decisive = z["top2_gap"] > 4e-4 * scale
    assert (r["merges"][decisive] == z["merges"][decisive]).all()
    return o, z, oh, mask


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_reference_small(name):
    o, z, oh, mask = _check(name)
    # free-running rollout reproduces the reference's merge list when no step is a near-tie
    scale = np.abs(z["logits"]).max()
    if (z["top2_gap"][:, :-1] > 4e-4 * scale).all():
        r = o.rollout_argmax(oh, mask)
        assert (r["merges"] == z["merges"]).all()


@pytest.mark.parametrize("name", BIG[:2])
def test_oracle_matches_reference_50x1024(name):
    _check(name)


# round 2: reference-scale weights and bench-style data at 50 x 1024, a ragged site count, more than 64 taxa
# (the reference's bundled 100-taxon evaluation alignments through its own loader)
@pytest.mark.parametrize("name", ["plain_b1_t50_l1024_s6", "benchlike_b1_t50_l1024_s0", "ragged_b1_t50_l1023_s10",
                                  "synth_b2_t70_l64_s12", "synth_b1_t100_l256_s11",
                                  "data_G_l_256_n_100_0_0p01_101"])
def test_oracle_matches_reference_round2_cases(name):
    if name == "synth_b1_t100_l256_s11":
        # 100 rows under the sharpened stress weights: the fp32 oracle's sequential sums are 1.2e-4 of the score
        # scale from an fp64 evaluation (the reference's blocked kernels: 4.4e-5).  The fp64 build carries the
        # 1e-4 pin of the semantics; the fp32 build is held to 2e-4 here.
        _check(name, "f64")
        _check(name, "f32", rel=2e-4)
    else:
        _check(name)


def test_oracle_matches_reference_round3_wide_cases():
    """Round 3: a second 100-taxon alignment under the stress weights (both builds within 1e-4 of the reference), and
    200 taxa x 256 sites.  At 200 rows the reference's OWN fp32 tables are 1.5e-4 (step 0) .. 4.6e-4 of the score
    scale from the fp64 evaluation -- fp32 rounding of the encoder output amplified by the six layers under these
    weights; the plain-fp32 oracle is 5.3e-4 from fp64, the HIP path 1.6e-4 (tools/golden_steps.py) -- so the pin of
    the semantics at this shape is the fp64 build within 6e-4 of the reference's tables and equal merges on every
    decisive step."""
    _check("synth_b1_t100_l256_s14", "f32")
    _check("synth_b1_t200_l256_s13", "f64", rel=6e-4)


@pytest.mark.parametrize("name", ["patch4_b2_t8_l128_s16", "patch4_b1_t20_l256_s17", "patch2_b1_t12_l96_s18",
                                  # narrower models, incl. the reference's default shape (32 features, 4 heads, 3 layers,
                                  # patch 4) at the size of its bundled examples
                                  "dim32_b1_t50_l1024_s22"])
def test_oracle_matches_reference_patched_tokens(name):
    """patch_size 4 and 2 (a token = several consecutive sites, model.py:72-79; the reference's utils.py default is 4):
    embed over the concatenated site vectors, the mask taken every patch_size-th site, alpha scaled by the patch count."""
    _check(name)


def test_oracle_free_run_rf_gate():
    """The RF = 0 gate of the GPU tests, exercised on the CPU: the fp32 oracle's free run against the
    reference's free run, fp64 build as arbiter where the merge lists part (helpers.free_run_verdict)."""
    from helpers import free_run_verdict
    for name in ("synth_b1_t20_l256_s2", "ragged_b2_t9_l30_s8"):
        z, cfgs, packed = load_golden(name)
        oh, mask = onehot_f32(z["codes"]), z["mask"]
        B, T, L = z["codes"].shape
        r = Oracle(cfgs, packed).rollout_argmax(oh, mask)
        mine, ref = split_trace(r["logits"], T), split_trace(z["logits"], T)
        for b in range(B):
            def truth(b=b):
                t = Oracle(cfgs, packed, "f64").rollout_argmax(oh[b:b + 1], mask[b:b + 1],
                                                                forced_merges=z["merges"][b:b + 1])
                return [x[0] for x in split_trace(t["logits"], T)]
            row = free_run_verdict(r["merges"][b], [x[b] for x in mine], z["merges"][b], [x[b] for x in ref],
                                   z["newick"][b], z["keys"][b], truth)
            assert row["rf"] == 0 or "first_divergent_step" in row


@pytest.mark.parametrize("name", SMALL[:3])
def test_oracle_f64_agrees(name):
    _check(name, "f64")


def test_component_entry_points_consistent():
    """decode_full / decode_incr / env_step / select_pair compose to the rollout."""
    z, cfgs, packed = load_golden("synth_b2_t8_l128_s1")
    o = Oracle(cfgs, packed)
    oh, mask = onehot_f32(z["codes"]), z["mask"]
    B, T, L = z["codes"].shape
    state = o.encode(oh, mask)
    ref_tables = split_trace(z["logits"], T)
    logits = o.pair_scores_full(state, mask)
    assert_logits_close(logits, ref_tables[0])
    for step, n in enumerate(range(T, 2, -1)):
        ij, _ = o.select_pair(logits, n)
        assert (ij == z["merges"][:, step]).all()
        state = o.env_step(state, ij)
        idx = o.score_index_map(ij, n - 1)
        logits_new, new = o.pair_scores_incr(state, mask, ij, logits, want_new=True)
        cat = np.concatenate([logits, new], 1)
        assert np.array_equal(np.take_along_axis(cat, idx, 1), logits_new)
        assert_logits_close(logits_new, ref_tables[step + 1])
        logits = logits_new


def test_oracle_sampling_mode_properties():
    """Sampling twin (finetune_rl_search.py:147): u -> 0 picks nothing below the first pair with
    mass, a very low temperature reproduces the argmax rollout, and merges are valid pairs."""
    z, cfgs, packed = load_golden("synth_b2_t8_l128_s0")
    o = Oracle(cfgs, packed)
    oh, mask = onehot_f32(z["codes"]), z["mask"]
    B, T, L = z["codes"].shape
    u = np.random.default_rng(3).random((B, T - 1)).astype(np.float32)
    cold = o.rollout_sample(oh, mask, u, temperature=1e-4)
    assert np.array_equal(cold["merges"], z["merges"])            # T -> 0: the sample is the argmax
    hot = o.rollout_sample(oh, mask, u, temperature=50.0)
    m = hot["merges"]
    assert (m[:, :, 0] < m[:, :, 1]).all() and (m[:, :, 1] < np.arange(T, 1, -1)[None, :]).all()
    assert not np.array_equal(m, z["merges"])                      # hot sampling explores
    again = o.rollout_sample(oh, mask, u, temperature=50.0)
    assert np.array_equal(again["merges"], m)                      # deterministic in the uniforms
    # inverse-CDF semantics on the first table, recomputed in numpy fp64
    t0 = hot["logits"][:, :T * (T - 1) // 2].astype(np.float64)
    e = np.exp((t0 - t0.max(1, keepdims=True)) / 50.0)
    cdf = np.cumsum(e, 1)
    k = np.array([int(np.searchsorted(cdf[b], float(u[b, 0]) * cdf[b, -1], side="right")) for b in range(B)])
    pairs = [(i, j) for i in range(T) for j in range(i + 1, T)]
    assert [pairs[x] for x in k] == [tuple(p) for p in m[:, 0]]
