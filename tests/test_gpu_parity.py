"""Parity of the HIP path (through the C ABI of libnnj_hip.so) against the CPU oracle, the
golden vectors captured from the reference, and size-independent properties at the
BASELINE sizes.  Tolerance: pair scores within 1e-4 of the table's scale (BASELINE.json:
"pairwise-distance floats within 1e-4 relative"); merge lists exact wherever the
reference's own top-2 gap is decisive (> 4e-4 of the scale), i.e. RF = 0."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_logits_close, flat_pair, golden_names, load_golden, onehot_f32, split_trace
from neuralnj_amd import synth, utils, weights

pytestmark = pytest.mark.gpu

RTOL = 1e-4
# No per-fixture tolerances.  Where a table misses 1e-4 of the REFERENCE's fp32 table (wide alignments under the
# sharpened stress weights: six encoder layers amplify fp32 rounding of the encoder output -- HIP's 1.6e-5, the fp32
# oracle's 1.4e-5 of its scale at 100 x 256 -- seven- to twenty-fold into the scores; every single HIP operation fed
# with exact inputs is within 3e-6, tools/step_ops_margin.py) the fp64 oracle arbitrates: two independent fp32
# evaluations of the fixture exist -- the reference's own tables and the plain-fp32 oracle -- and HIP must be no
# farther from the fp64 tables than twice the larger of THEIR distances (measured: 100 x 256 s11: HIP 1.7e-4, reference
# 4.4e-5, fp32 oracle 1.2e-4; 200 x 256 s13: HIP 1.6e-4, reference 4.6e-4, fp32 oracle 5.3e-4; 100 x 256 s14 / s15 and
# the reference's real 100-taxon alignments: all three within 3e-5).  The figures of a run go to
# gpurun_out/golden_noise.json.
NOISE_ROWS = []


def _fp64_arbitration(name, z, cfgs, packed, logits):
    """(hip vs fp64, reference vs fp64, fp32 oracle vs fp64), scale-relative, over all tables of the fixture."""
    from oracle_lib import Oracle
    oh = onehot_f32(z["codes"])
    t64 = Oracle(cfgs, packed, "f64").rollout_argmax(oh, z["mask"], forced_merges=z["merges"])["logits"]
    t32 = Oracle(cfgs, packed).rollout_argmax(oh, z["mask"], forced_merges=z["merges"])["logits"]
    scale = max(float(np.abs(t64).max()), 1.0)
    row = dict(fixture=name, hip_vs_fp64=float(np.abs(logits - t64).max()) / scale,
               reference_vs_fp64=float(np.abs(z["logits"] - t64).max()) / scale,
               fp32_oracle_vs_fp64=float(np.abs(t32 - t64).max()) / scale,
               hip_vs_reference=float(np.abs(logits - z["logits"]).max()) / scale)
    NOISE_ROWS.append(row)
    return row


@pytest.fixture(scope="module")
def ctx_cache():
    from neuralnj_amd._lib import Nnj
    cache = {}

    def get(cfgs, packed):
        key = (int(cfgs.model.num_enc_layers), weights.digest(packed))
        if key not in cache:
            g = Nnj(cfgs, "cuda:0")
            g.load_weights(packed)
            cache[key] = g
        return cache[key]
    yield get
    for g in cache.values():
        g.close()


def _oracle(cfgs, packed):
    from oracle_lib import Oracle
    return Oracle(cfgs, packed)


def _oracle_f64(cfgs, packed):
    from oracle_lib import Oracle
    return Oracle(cfgs, packed, "f64")


def _four_pass_ctx(cfgs, packed):
    """A handle whose rollouts run the four-pass NJ step (k_agg_alpha / k_agg_finish / k_inc_alpha16 / k_inc_score*), the
    kernels the step-by-step entry points (nnj_step, nnj_env_step, nnj_pair_scores_incr) run; the default rollout runs
    the two-pass step of csrc/nnj_step2.hpp (NNJ_TWO_PASS is read by nnj_create)."""
    import os
    from neuralnj_amd._lib import Nnj
    os.environ["NNJ_TWO_PASS"] = "0"
    try:
        g = Nnj(cfgs, "cuda:0")
    finally:
        del os.environ["NNJ_TWO_PASS"]
    g.load_weights(packed)
    return g


# fixtures on which HIP is known to exceed max(1e-4, 2 x the reference's own fp64 distance): name -> the ceiling the
# expected failure may not pass.  EMPTY since round 5: the one entry (synth_b1_t100_l256_s11, 1.75e-4 from fp64) is cured by
# the fp64 encoder of the > 64-row path (csrc/nnj_encoder64.hpp): 5.3e-6 (profiles/r05/noise_enc64.txt).
ABOVE_TOLERANCE = {}

RF_ROWS = []          # one row per (fixture, alignment): written to gpurun_out/rf_table.json at module teardown


@pytest.fixture(scope="module", autouse=True)
def _rf_table_dump():
    yield
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if RF_ROWS and os.path.isdir(out):
        summary = dict(alignments=len(RF_ROWS), rf_zero=sum(r["rf"] == 0 for r in RF_ROWS),
                       identical_merge_lists=sum(r["identical_merges"] for r in RF_ROWS),
                       arbitrated_near_ties=sum(not r["identical_merges"] for r in RF_ROWS))
        with open(os.path.join(out, "rf_table.json"), "w") as f:
            json.dump(dict(what="free-running HIP rollout against the reference's own tree, per golden alignment "
                                "(tests/test_gpu_parity.py::test_rollout_matches_reference_golden)",
                           summary=summary, rows=RF_ROWS), f, indent=1)
    if NOISE_ROWS and os.path.isdir(out):
        with open(os.path.join(out, "golden_noise.json"), "w") as f:
            json.dump(dict(what="fixtures whose HIP tables miss 1e-4 of the reference's fp32 tables: distances from the fp64 "
                                "oracle's tables (scale-relative, whole rollout, teacher-forced along the reference's merges)",
                           rows=NOISE_ROWS), f, indent=1)


@pytest.mark.parametrize("name", golden_names())
def test_rollout_matches_reference_golden(name, ctx_cache):
    """(1) Teacher-forced along the reference's merges: every per-step table within tolerance of what the
    reference computed, argmax equal on decisive steps.  (2) ALWAYS the free-running rollout: Robinson-Foulds
    distance of the HIP tree to the reference's tree; identical merge lists (RF = 0), or -- helpers.free_run_verdict --
    the first divergent step is a near-tie by the fp64 oracle and the HIP pick is the reference's runner-up."""
    from helpers import free_run_verdict
    z, cfgs, packed = load_golden(name)
    g = ctx_cache(cfgs, packed)
    codes, mask = torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"])
    B, T, L = z["codes"].shape
    r = g.rollout_argmax(codes, mask, forced_merges=z["merges"], want_trace=True, want_state=True)
    logits = r["logits"].cpu().numpy()
    scale_ = max(float(np.abs(z["logits"]).max()), 1.0)
    expected_failure = None                               # (reported at the END: every other check of the fixture still runs)
    if float(np.abs(logits - z["logits"]).max()) > RTOL * scale_:
        # Where a table misses 1e-4 of the reference's fp32 table the fp64 oracle arbitrates, and the bound is set by the
        # REFERENCE's own distance from fp64, not by this repo's fp32 oracle: HIP must be within
        # max(1e-4, 2 x reference_vs_fp64) of the fp64 tables.
        row = _fp64_arbitration(name, z, cfgs, packed, logits)
        assert row["reference_vs_fp64"] <= 1e-3, f"the fp64 oracle does not reproduce the reference: {row}"
        bound = max(RTOL, 2.0 * row["reference_vs_fp64"])
        if row["hip_vs_fp64"] > bound and name in ABOVE_TOLERANCE:
            # an EXPECTED FAILURE, shown as such in the test summary -- not a widened gate (no fixture is listed any more)
            assert row["hip_vs_fp64"] <= ABOVE_TOLERANCE[name], row
            expected_failure = (f"{name}: HIP {row['hip_vs_fp64']:.2e} from fp64 against a bound of {bound:.1e} "
                                f"(reference {row['reference_vs_fp64']:.2e}); known, measured, not cured: DESIGN.md section 2")
        else:
            assert row["hip_vs_fp64"] <= bound, \
            f"HIP is farther from the fp64 tables than max(1e-4, twice the reference's own distance): {row}"
    st = r["state"].cpu().numpy()
    if "enc" in z.files:
        np.testing.assert_allclose(st, z["enc"], atol=RTOL * np.abs(z["enc"]).max())
    else:
        np.testing.assert_allclose(st[:, ::7, ::61, :], z["enc_slice"], atol=RTOL * np.abs(z["enc_slice"]).max())
        assert abs(st.astype(np.float64).sum() - float(z["enc_checksum"])) <= 1e-5 * float(z["enc_abs_checksum"])
    scale = np.abs(z["logits"]).max()
    decisive = z["top2_gap"] > 4 * RTOL * scale
    merges = r["merges"].cpu().numpy()
    assert (merges[decisive] == z["merges"][decisive]).all()
    np.testing.assert_allclose(r["top2_gap"].cpu().numpy()[decisive], z["top2_gap"][decisive], atol=2 * RTOL * scale)
    # ---- free run, every fixture
    free = g.rollout_argmax(codes, mask, want_trace=True)
    g.check_numeric()
    fm = free["merges"].cpu().numpy()
    ft, gt = split_trace(free["logits"].cpu().numpy(), T), split_trace(z["logits"], T)
    oh = onehot_f32(z["codes"])
    for b in range(B):
        def truth(b=b):
            from oracle_lib import Oracle
            t = Oracle(cfgs, packed, "f64").rollout_argmax(oh[b:b + 1], z["mask"][b:b + 1],
                                                            forced_merges=z["merges"][b:b + 1])
            return [x[0] for x in split_trace(t["logits"], T)]
        row = free_run_verdict(fm[b], [x[b] for x in ft], z["merges"][b], [x[b] for x in gt], z["newick"][b],
                               z["keys"][b], truth)
        RF_ROWS.append(dict(fixture=name, alignment=b, taxa=T, sites=L, **row))
    if expected_failure:
        pytest.xfail(expected_failure)


@pytest.mark.parametrize("name", ["synth_b1_t8_l128_s0", "synth_b1_t8_l128_s2", "tiny_b2_t3_l64_s5"])
def test_encoder_taps_match_oracle(name, ctx_cache):
    z, cfgs, packed = load_golden(name)
    g = ctx_cache(cfgs, packed)
    o = _oracle(cfgs, packed)
    enc, taps = o.encode(onehot_f32(z["codes"]), z["mask"], taps=True)
    codes, mask = torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"])
    try:
        for stop, ref in ((1, taps[1]), (2, taps[2]), (0, enc)):
            g.debug_encoder_stop(stop)
            out = g.encode(codes, mask).cpu().numpy()
            np.testing.assert_allclose(out, ref, atol=2e-5 * max(1.0, np.abs(ref).max()))
    finally:
        g.debug_encoder_stop(0)


@pytest.mark.parametrize("name", ["synth_b2_t8_l128_s1", "padded_b2_t8_l128_s3", "plain_b1_t12_l96_s4",
                                  "synth_b1_t20_l256_s1", "synth_b2_t70_l64_s12", "patch4_b2_t8_l128_s16",
                                  # embed_dim 32 / 16 (zero-padded to the kernels' 64 features): dense tensors of the
                                  # model's own width at every entry point
                                  "dim32_b2_t10_l256_s19", "dim32_b1_t20_l160_s20", "dim16_b1_t9_l96_s21"])
def test_entry_points_match_oracle_step_by_step(name, ctx_cache):
    """The reference's call sequence (decode_zxr / argmax / env.step) through the separate
    C-ABI entry points, each compared with the oracle on the same inputs."""
    z, cfgs, packed = load_golden(name)
    g = ctx_cache(cfgs, packed)
    o = _oracle(cfgs, packed)
    B, T, L = z["codes"].shape
    mask = z["mask"]
    tm = torch.from_numpy(mask)
    state = o.encode(onehot_f32(z["codes"]), mask)
    logits = o.pair_scores_full(state, mask)
    got = g.pair_scores_full(torch.from_numpy(state), tm).cpu().numpy()
    assert_logits_close(got, logits, RTOL, "pair_scores_full")
    for step, n in enumerate(range(T, 2, -1)):
        ij, gap = o.select_pair(logits, n)
        gij, ggap = g.select_pair(torch.from_numpy(logits), n)
        assert np.array_equal(gij.cpu().numpy(), ij)                        # integer work: exact
        np.testing.assert_allclose(ggap.cpu().numpy(), gap, rtol=0, atol=0)
        agg = o.aggregate(state, ij)
        np.testing.assert_allclose(g.aggregate(torch.from_numpy(state), ij).cpu().numpy(), agg,
                                   atol=RTOL * np.abs(agg).max())
        nstate = o.env_step(state, ij)
        gs = g.env_step(torch.from_numpy(state), ij).cpu().numpy()
        np.testing.assert_allclose(gs, nstate, atol=RTOL * np.abs(nstate).max())
        untouched = np.ones(n - 1, bool)
        for b in range(B):                                                    # copied rows: bit exact
            untouched[:] = True
            untouched[ij[b, 0]] = False
            assert np.array_equal(gs[b, untouched], nstate[b, untouched])
        assert np.array_equal(g.score_index_map(ij, n - 1).cpu().numpy(), o.score_index_map(ij, n - 1))
        nl = o.pair_scores_incr(nstate, mask, ij, logits)
        gl = g.pair_scores_incr(torch.from_numpy(nstate), tm, ij, torch.from_numpy(logits)).cpu().numpy()
        assert_logits_close(gl, nl, RTOL, f"pair_scores_incr n={n - 1}")
        # cached entries are copied, not recomputed: bit exact
        idx = o.score_index_map(ij, n - 1)
        old = idx < logits.shape[1]
        assert np.array_equal(gl[old], np.take_along_axis(logits, np.minimum(idx, logits.shape[1] - 1), 1)[old])
        state, logits = nstate, nl


def test_seeded_random_inputs_vs_oracle(ctx_cache):
    """Fresh seeded inputs (not fixtures): HIP rollout vs oracle, ragged shapes."""
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 21, "sharp"))
    g = ctx_cache(cfgs, packed)
    o, o64 = _oracle(cfgs, packed), _oracle_f64(cfgs, packed)
    # shapes chosen to hit every kernel variant: row counts <= 16, 17..32, 33..48, 49..64 (the incremental scorer
    # changes tile shape at those points), odd T (zero-padded k-step of the row attention), L = 4 (one 16-column
    # block, mostly padding), L not a multiple of 16 / 256, L > 1024 (five 256-row operand blocks)
    for (B, T, L, seed) in ((3, 5, 36, 1), (1, 2, 64, 2), (2, 33, 100, 3), (1, 17, 260, 4), (1, 50, 128, 5),
                            (1, 64, 64, 6), (2, 57, 96, 7), (2, 9, 4, 8), (1, 20, 12, 9), (1, 40, 268, 10),
                            (1, 3, 1028, 11), (1, 48, 32, 12), (1, 49, 16, 13),
                            # more than 64 rows: 4 / 8 waves per column, e-chunked row context, star scorer kernels
                            # (128-row images up to 128 rows, 256-row images above), 256 = the largest row count covered
                            (1, 65, 32, 14), (2, 100, 48, 15), (1, 129, 40, 16), (1, 200, 36, 17), (1, 256, 20, 18)):
        codes = synth.synth_codes_tree(B, T, L, seed)
        mask = np.zeros((B, L), bool)
        if seed % 2:
            codes[:, :, L - 8:] = 5
            mask[:, L - 8:] = True
        free = g.rollout_argmax(torch.from_numpy(codes), torch.from_numpy(mask), want_trace=True)
        merges = free["merges"].cpu().numpy()
        # More than 64 rows: the HIP encoder runs in fp64 (round 5), so the checker must not be noisier than the code it
        # checks -- under these stress weights the plain-fp32 oracle is itself up to 1.2e-4 from the fp64 evaluation at 256
        # rows (that distance was 1e-4-class on BOTH sides before).  The fp64 build of the oracle is the reference there.
        ref = (o64 if T > 64 else o).rollout_argmax(synth.codes_to_onehot(codes).astype(np.float32), mask, forced_merges=merges)
        assert_logits_close(free["logits"].cpu().numpy(), ref["logits"], RTOL, f"{B}x{T}x{L}")
        scale = np.abs(ref["logits"]).max()
        decisive = ref["top2_gap"] > 4 * RTOL * scale
        assert (ref["merges"][decisive] == merges[decisive]).all()
        assert (merges[:, :, 0] < merges[:, :, 1]).all()
        assert (merges[:, :, 1] < np.arange(T, 1, -1)[None, :]).all()


def test_narrow_model_seeded_inputs_vs_oracle(ctx_cache):
    """A model of 32 features / 4 heads / 2 layers / patch 2 (zero-padded onto the 64-feature kernels) on fresh seeded
    inputs that reach every kernel family: 16-pair and 32-pair tiles, the two-pass step, the star kernels above 64 and
    above 128 rows, a padded tail, sampled replicas of one alignment (first table computed once), the step API."""
    cfgs = utils.shipped_config()
    cfgs.model.embed_dim, cfgs.model.num_enc_heads, cfgs.model.num_enc_layers, cfgs.model.patch_size = 32, 4, 2, 2
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 23, "sharp"))
    g = ctx_cache(cfgs, packed)
    o, o64 = _oracle(cfgs, packed), _oracle_f64(cfgs, packed)
    for (B, T, L, seed) in ((2, 9, 40, 1), (1, 33, 64, 2), (2, 50, 96, 3), (1, 70, 48, 4), (1, 130, 32, 5)):
        codes = synth.synth_codes_tree(B, T, L, 100 + seed)
        mask = np.zeros((B, L), bool)
        if seed % 2:
            codes[:, :, L - 6:] = 5
            mask[:, L - 6:] = True
        tc, tm = torch.from_numpy(codes), torch.from_numpy(mask)
        free = g.rollout_argmax(tc, tm, want_trace=True, want_state=True)
        assert free["state"].shape == (B, T, L // 2, 32)
        merges = free["merges"].cpu().numpy()
        ref = (o64 if T > 64 else o).rollout_argmax(synth.codes_to_onehot(codes).astype(np.float32), mask, forced_merges=merges)
        assert_logits_close(free["logits"].cpu().numpy(), ref["logits"], RTOL, f"narrow model {B}x{T}x{L}")
        scale = np.abs(ref["logits"]).max()
        decisive = ref["top2_gap"] > 4 * RTOL * scale
        assert (ref["merges"][decisive] == merges[decisive]).all()
        if T <= 50:
            tabs, picks = _step_loop(g, tc, tm, T)
            ft = split_trace(free["logits"].cpu().numpy(), T)
            for s_ in range(T - 1):
                assert_logits_close(tabs[s_], ft[s_], 0.2 * RTOL, f"narrow model, step API table {s_}")
    # sampled replicas of ONE alignment against the oracle twin on the same uniforms
    T, L, R = 12, 48, 6
    codes = synth.synth_codes_tree(1, T, L, 111)
    u = np.random.default_rng(9).random((R, T - 1)).astype(np.float32)
    rs = g.rollout_sample(torch.from_numpy(codes), None, torch.from_numpy(u), temperature=1.0, replicas=R, want_trace=True)
    ref = o.rollout_sample(np.repeat(synth.codes_to_onehot(codes).astype(np.float32), R, 0), None, u, 1.0)
    agree = _certify_sampled(ref, rs["merges"].cpu().numpy(), rs["logits"].cpu().numpy(), u, 1.0, T)
    assert agree.sum() >= R - 1
    g.check_numeric()


@pytest.mark.parametrize("shape", [(3, 12, 40, 31), (2, 40, 72, 32), (1, 64, 48, 33), (2, 70, 32, 34)])
def test_forced_random_merges_vs_oracle(shape, ctx_cache):
    """Teacher-forcing along RANDOM merge lists (valid pairs that are almost never the argmax): in the two-pass NJ step
    the attention weights of such a merge come from neither of the two sources the scorer provides (a pair of the last
    step, the carried candidate) but from the per-alignment fallback kernels (k_pair_xp / k_agg_dot / k_agg_am) --
    except where the random pick happens to be one of them, so the three routes are mixed inside one batch.  Tables
    against the oracle along the same lists."""
    B, T, L, seed = shape
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 21, "sharp"))
    g = ctx_cache(cfgs, packed)
    o = _oracle(cfgs, packed)
    rng = np.random.default_rng(seed)
    codes = synth.synth_codes_tree(B, T, L, seed)
    mask = np.zeros((B, L), bool)
    forced = np.zeros((B, T - 1, 2), np.int32)
    for b in range(B):
        for s_, n in enumerate(range(T, 1, -1)):
            if s_ % 3 == 1 and s_ > 0:                       # every third step: merge the fresh row again (a "new pair")
                i = int(forced[b, s_ - 1, 0])
                j = int(rng.integers(0, n - 1))
                j = j + 1 if j >= i else j
                forced[b, s_] = sorted((min(i, n - 1), j)) if min(i, n - 1) != j else (0, 1)
            else:
                forced[b, s_] = sorted(rng.choice(n, 2, replace=False))
    r = g.rollout_argmax(torch.from_numpy(codes), torch.from_numpy(mask), forced_merges=forced, want_trace=True)
    g.check_numeric()
    ref = o.rollout_argmax(synth.codes_to_onehot(codes).astype(np.float32), mask, forced_merges=forced)
    assert_logits_close(r["logits"].cpu().numpy(), ref["logits"], RTOL, f"forced random merges {B}x{T}x{L}")
    # (`merges` reports the model's own argmax at every forced state, as the oracle's does)
    scale = np.abs(ref["logits"]).max()
    decisive = ref["top2_gap"] > 4 * RTOL * scale
    assert (ref["merges"][decisive] == r["merges"].cpu().numpy()[decisive]).all()


@pytest.mark.parametrize("shape", [(2, 70, 64), (1, 100, 96), (1, 130, 40), (1, 200, 64), (1, 256, 32)])
def test_wide_encoder_matches_oracle(shape, ctx_cache):
    """More than 64 alignment rows: k_tok1p with 4 / 8 waves per column (online softmax), k_row_pv over e-chunks."""
    B, T, L = shape
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 41, "sharp"))
    g = ctx_cache(cfgs, packed)
    codes = synth.synth_codes_tree(B, T, L, 500 + T)
    mask = np.zeros((B, L), bool)
    mask[:, L - 5:] = True
    codes[:, :, L - 5:] = 5
    ref = _oracle(cfgs, packed).encode(onehot_f32(codes), mask)
    got = g.encode(torch.from_numpy(codes), torch.from_numpy(mask)).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=RTOL * np.abs(ref).max())


@pytest.mark.parametrize("case", [(2, 70, 64, 1, 64, 6), (1, 100, 96, 1, 64, 6), (1, 130, 41, 1, 64, 2), (1, 256, 32, 1, 64, 1),
                                  (1, 80, 48, 2, 64, 2), (2, 72, 40, 4, 32, 3)])
def test_wide_encoder_is_fp64_accurate(case):
    """More than 64 rows: the encoder runs in fp64 (csrc/nnj_encoder64.hpp: one tiled v_mfma_f64_16x16x4 GEMM with epilogue
    functors + embed / LayerNorm / softmax / column-attention kernels) and rounds the embeddings to fp32 once.  Against
    the fp64 oracle's encoder: masked tails, odd site counts, patches, a narrow model, the code and the one-hot input
    form, and the layer-0 taps -- all within 5e-7 of the tensor's scale (one fp32 rounding is 6e-8; the f16x3 encoder
    and the plain-fp32 oracle sit at 2e-6 .. 2e-5 here)."""
    from neuralnj_amd._lib import Nnj
    B, T, L, K, dim, layers = case
    cfgs = utils.shipped_config()
    cfgs.model.num_enc_layers, cfgs.model.patch_size, cfgs.model.embed_dim, cfgs.model.num_enc_heads = layers, K, dim, dim // 8
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 41, "sharp"))
    g = Nnj(cfgs, "cuda:0")
    try:
        g.load_weights(packed)
        codes = synth.synth_codes_tree(B, T, L, 500 + T)
        mask = np.zeros((B, L), bool)
        mask[:, L - 2 * K:] = True
        codes[:, :, L - 2 * K:] = 5
        o64 = _oracle_f64(cfgs, packed)
        e64, taps = o64.encode(onehot_f32(codes), mask, taps=True)
        tc, tm = torch.from_numpy(codes), torch.from_numpy(mask)
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
        for stop in (1, 2):
            g.debug_encoder_stop(stop)
            assert rel(g.encode(tc, tm).cpu().numpy(), taps[stop]) <= 5e-7, f"layer-0 tap {stop}"
        g.debug_encoder_stop(0)
        assert rel(g.encode(tc, tm).cpu().numpy(), e64) <= 5e-7
        assert rel(g.encode_onehot(torch.from_numpy(onehot_f32(codes)), tm).cpu().numpy(), e64) <= 5e-7
    finally:
        g.close()


def test_config5_200x4096_properties(ctx_cache):
    """BASELINE configs[4] shape (200 taxa x 4096 sites; the reference's formulation does not fit this container, so
    there is no golden): size-independent properties of the HIP rollout at the full shape -- (1) two copies of the
    alignment in one batch give bit-identical tables and merges; (2) teacher-forcing the free run's own merges
    reproduces its tables bit for bit; (3) merges are valid pair indices; (4) the encoder output and the all-pairs
    table of step 0 on a window of the SAME alignment (first 160 sites; the oracle needs minutes for 4096) equal
    the oracle -- with 200 rows live, i.e. through the same 8-waves-per-column, e-chunked and 256-row-image kernels."""
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
    g = ctx_cache(cfgs, packed)
    T, L = 200, 4096
    one = synth.synth_codes_tree(1, T, L, seed=4242)
    codes = torch.from_numpy(np.concatenate([one, one], 0))
    r = g.rollout_argmax(codes, None, want_trace=True)
    g.check_numeric()
    merges, logits = r["merges"].cpu().numpy(), r["logits"].cpu().numpy()
    assert np.array_equal(merges[0], merges[1]) and np.array_equal(logits[0], logits[1])
    assert (merges[:, :, 0] < merges[:, :, 1]).all()
    assert (merges[:, :, 1] < np.arange(T, 1, -1)[None, :]).all()
    rf = g.rollout_argmax(codes[:1], None, forced_merges=merges[:1], want_trace=True)
    # a batch of one runs other launch geometries (site chunks per workgroup): tables within tolerance, and the
    # same geometry again is bit-identical
    assert_logits_close(rf["logits"].cpu().numpy(), logits[:1], RTOL, "B=1 vs B=2")
    rf2 = g.rollout_argmax(codes[:1], None, forced_merges=merges[:1], want_trace=True)
    assert torch.equal(rf["logits"], rf2["logits"]) and torch.equal(rf["merges"], rf2["merges"])
    win = one[:, :, :160]
    o = _oracle(cfgs, packed)
    ref = o.rollout_argmax(onehot_f32(win), None, forced_merges=None)
    got = g.rollout_argmax(torch.from_numpy(win), None, forced_merges=ref["merges"], want_trace=True)
    assert_logits_close(got["logits"].cpu().numpy(), ref["logits"], 2 * RTOL, "200 x 160 window vs the fp32 oracle")
    ref64 = _oracle_f64(cfgs, packed).rollout_argmax(onehot_f32(win), None, forced_merges=ref["merges"])
    assert_logits_close(got["logits"].cpu().numpy(), ref64["logits"], RTOL, "200 x 160 window vs the fp64 oracle")


def test_full_size_batch_properties(ctx_cache):
    """BASELINE configs[2] size (256 x 50 x 1024): size-independent properties.
    (1) identical MSAs in one batch give identical merge lists and tables, bit for bit;
    (2) a permuted batch gives the permuted result; (3) teacher-forcing the free run's own
    merges reproduces its tables bit for bit; (4) a sample of trees equals the oracle."""
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
    g = ctx_cache(cfgs, packed)
    B, T, L = 256, 50, 1024
    base = synth.synth_codes_tree(32, T, L, seed=77)
    codes = np.concatenate([base] * 8, 0)                      # every MSA appears 8 times
    tc = torch.from_numpy(codes)
    r = g.rollout_argmax(tc, None, want_trace=True)
    merges, logits = r["merges"].cpu().numpy(), r["logits"].cpu().numpy()
    for k in range(1, 8):
        assert np.array_equal(merges[:32], merges[32 * k:32 * (k + 1)])
        assert np.array_equal(logits[:32], logits[32 * k:32 * (k + 1)])
    perm = np.random.default_rng(0).permutation(B)
    rp = g.rollout_argmax(torch.from_numpy(codes[perm]), None)
    assert np.array_equal(rp["merges"].cpu().numpy(), merges[perm])
    rf = g.rollout_argmax(tc, None, forced_merges=merges, want_trace=True)
    assert np.array_equal(rf["logits"].cpu().numpy(), logits)
    assert np.array_equal(rf["merges"].cpu().numpy(), merges)
    o = _oracle(cfgs, packed)
    sample = [0, 13, 31]
    ref = o.rollout_argmax(synth.codes_to_onehot(codes[sample]).astype(np.float32), None, forced_merges=merges[sample])
    assert_logits_close(logits[sample], ref["logits"], RTOL, "full-size sample")
    scale = np.abs(ref["logits"]).max()
    decisive = ref["top2_gap"] > 4 * RTOL * scale
    assert (ref["merges"][decisive] == merges[sample][decisive]).all()


def test_mask_none_equals_all_false(ctx_cache):
    z, cfgs, packed = load_golden("synth_b2_t8_l128_s0")
    g = ctx_cache(cfgs, packed)
    codes = torch.from_numpy(z["codes"])
    a = g.rollout_argmax(codes, None, want_trace=True)
    b = g.rollout_argmax(codes, torch.zeros(2, 128, dtype=torch.bool), want_trace=True)
    assert torch.equal(a["logits"], b["logits"]) and torch.equal(a["merges"], b["merges"])


def _step_loop(g, codes, mask, T):
    """encode + all-pairs table + (T - 2) x nnj_step, every call merging the pair the previous one picked."""
    state = g.encode(codes, mask)
    logits = g.pair_scores_full(state, mask)
    ij, _ = g.select_pair(logits, T)
    tabs, picks = [logits.cpu().numpy()], [ij.cpu().numpy()]
    for n in range(T - 1, 1, -1):
        r = g.step(state, mask, ij, logits)
        state, logits, ij = r["state"], r["logits"], r["ij"]
        assert state.shape[1] == n
        tabs.append(logits.cpu().numpy())
        picks.append(ij.cpu().numpy())
    g.check_numeric()
    return tabs, picks


@pytest.mark.parametrize("name", ["synth_b2_t8_l128_s1", "synth_b1_t20_l256_s0", "synth_b2_t70_l64_s12"])
def test_fused_step_reproduces_the_rollout(name, ctx_cache):
    """nnj_step (merge + new scores + table + argmax in one call) iterated from the encoder output.  On a handle whose
    steps run the four-pass kernels it gives the tables and merges of that handle's rollout bit for bit (same kernels,
    same order; 70 rows: through the star kernels above 64 live rows, then the 64-row kernels).  On the default handle
    the step runs the rollout's two-pass kernels up to 64 rows (the first merge takes its attention weights from the
    fallback kernels instead of the all-pairs kernel's partials: another summation order): the rollout's tables within a
    fifth of the tolerance, the same merges; above 64 rows the four-pass kernels, bit for bit."""
    z, cfgs, packed = load_golden(name)
    g = ctx_cache(cfgs, packed)
    codes, mask = torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"])
    B, T, L = z["codes"].shape
    g4 = _four_pass_ctx(cfgs, packed)
    ref = g4.rollout_argmax(codes, mask, want_trace=True, want_state=True)
    tables = split_trace(ref["logits"].cpu().numpy(), T)
    merges = ref["merges"].cpu().numpy()
    tabs4, picks4 = _step_loop(g4, codes, mask, T)
    g4.close()
    for step in range(T - 1):
        assert np.array_equal(tabs4[step], tables[step]), f"four-pass step API: table after merge {step}"
        assert np.array_equal(picks4[step], merges[:, step])
    two = g.rollout_argmax(codes, mask, want_trace=True, forced_merges=merges)
    assert_logits_close(two["logits"].cpu().numpy(), ref["logits"].cpu().numpy(), 0.2 * RTOL, "two-pass vs four-pass step")
    tabs, picks = _step_loop(g, codes, mask, T)
    if T > 64:
        for step in range(T - 1):
            assert np.array_equal(tabs[step], tables[step]), f"table after merge {step}"
            assert np.array_equal(picks[step], merges[:, step])
        return
    free = g.rollout_argmax(codes, mask, want_trace=True)
    ftabs, fm = split_trace(free["logits"].cpu().numpy(), T), free["merges"].cpu().numpy()
    assert np.array_equal(tabs[0], ftabs[0])
    for step in range(T - 1):
        assert np.array_equal(picks[step], fm[:, step]), f"pick {step}"
        assert_logits_close(tabs[step], ftabs[step], 0.2 * RTOL, f"step API vs rollout, table after merge {step}")


@pytest.mark.parametrize("shape", [(3, 12, 40, 41), (2, 40, 72, 42), (1, 64, 48, 43)])
def test_fused_step_with_caller_chosen_merges_vs_oracle(shape, ctx_cache):
    """nnj_step merging pairs the CALLER names: random valid pairs (the weights of such a merge come from the fallback
    kernels), the pair the previous call picked (from that call's table kernel: a new pair or the carried candidate), and
    a forced_next pair (the table kernel prepares THAT merge) -- mixed over the steps and inside the batch.  Tables
    against the oracle teacher-forced along the merges actually applied."""
    B, T, L, seed = shape
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 21, "sharp"))
    g = ctx_cache(cfgs, packed)
    o = _oracle(cfgs, packed)
    rng = np.random.default_rng(seed)
    codes = synth.synth_codes_tree(B, T, L, seed)
    mask = np.zeros((B, L), bool)
    tc, tm = torch.from_numpy(codes), torch.from_numpy(mask)
    state = g.encode(tc, tm)
    logits = g.pair_scores_full(state, tm)
    ij = g.select_pair(logits, T)[0].cpu().numpy()
    applied, tabs = [], [logits.cpu().numpy()]
    forced_next = None
    for s_, n in enumerate(range(T, 2, -1)):                 # n rows before the merge
        use = ij.copy()                                      # default: the pair the last call picked
        if forced_next is not None:
            use = forced_next                                # ... which was forced
        for b in range(B):
            if (s_ + b) % 3 == 1:                            # a pair of the caller's own
                use[b] = sorted(rng.choice(n, 2, replace=False))
        applied.append(use.copy())
        forced_next = None
        fn = None
        if s_ % 4 == 2 and n - 1 >= 2:                       # ask the table kernel to prepare a given merge
            fn = np.stack([np.array(sorted(rng.choice(n - 1, 2, replace=False)), np.int32) for _ in range(B)])
            forced_next = fn.copy()
        r = g.step(state, tm, torch.from_numpy(use.astype(np.int32)), logits,
                   forced_next=None if fn is None else torch.from_numpy(fn))
        state, logits = r["state"], r["logits"]
        ij = r["ij"].cpu().numpy()
        if fn is not None:
            assert np.array_equal(ij, fn)
        tabs.append(logits.cpu().numpy())
    g.check_numeric()
    forced = np.zeros((B, T - 1, 2), np.int32)
    forced[:, :T - 2] = np.stack(applied, 1)
    forced[:, T - 2] = (0, 1)
    ref = o.rollout_argmax(synth.codes_to_onehot(codes).astype(np.float32), mask, forced_merges=forced)
    rt = split_trace(ref["logits"], T)
    for s_ in range(T - 1):
        assert_logits_close(tabs[s_], rt[s_], RTOL, f"step API, caller's merges {B}x{T}x{L}, table {s_}")


@pytest.mark.parametrize("name", ["synth_b2_t8_l128_s1", "synth_b1_t20_l256_s0"])
def test_dense_state_session_reproduces_the_rollout(name, ctx_cache):
    """The reference's call sequence decode_zxr -> argmax -> env.step -> decode_zxr ... through the dense-state entry
    points (nnj_pair_scores_full / nnj_env_step / nnj_pair_scores_incr): handed the tensors it produced, the library
    continues a session (include/nnj.h) and gives the tables of a rollout on the same (four-pass) kernels bit for bit; a
    state tensor the caller has rewritten in place is NOT taken for the session's (stateless path, same values within
    tolerance)."""
    z, cfgs, packed = load_golden(name)
    g = ctx_cache(cfgs, packed)
    codes, mask = torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"])
    B, T, L = z["codes"].shape
    g4 = _four_pass_ctx(cfgs, packed)
    ref = g4.rollout_argmax(codes, mask, want_trace=True)
    g4.close()
    tables = split_trace(ref["logits"].cpu().numpy(), T)
    merges = ref["merges"].cpu().numpy()
    state = g.encode(codes, mask)
    logits = g.pair_scores_full(state, mask)
    assert np.array_equal(logits.cpu().numpy(), tables[0])
    for step, n in enumerate(range(T, 2, -1)):
        ij = torch.from_numpy(merges[:, step].copy())
        if step == 2:
            # a merged row computed from the session's rows, not in place: the session goes on afterwards
            row = g.aggregate(state, ij)
        new_state = g.env_step(state, ij)
        if step == 2:
            slot = int(merges[0, step, 0])
            assert torch.equal(row[0, 0], new_state[0, slot])
        state = new_state
        assert state.shape[1] == n - 1
        logits = g.pair_scores_incr(state, mask, ij, logits)
        assert np.array_equal(logits.cpu().numpy(), tables[step + 1]), f"table after merge {step + 1}"
    g.check_numeric()
    # in-place write to the session's tensor: the wrapper resets the session, the call runs stateless
    state = g.encode(codes, mask)
    l0 = g.pair_scores_full(state, mask)
    ij0 = torch.from_numpy(merges[:, 0].copy())
    s1 = g.env_step(state, ij0)
    want = g.pair_scores_incr(s1, mask, ij0, l0).cpu().numpy()
    s1.mul_(1.0)                                            # bumps the version counter, values unchanged
    got = g.pair_scores_incr(s1, mask, ij0, l0).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=RTOL * np.abs(want).max())
    s1.add_(1.0)                                            # really different rows: the result must follow them
    moved = g.pair_scores_incr(s1, mask, ij0, l0).cpu().numpy()
    assert np.abs(moved - want).max() > 1e-3 * np.abs(want).max()


def test_batch_shards_reproduce_the_whole_batch(ctx_cache):
    """Multi-GPU sharding is a contiguous split of the batch: a shard must give what the same
    alignments give inside the whole batch.  Launch geometry (site chunks per workgroup, small-batch kernels)
    depends on B, so partial sums may be added in another order: tables within tolerance, merges identical on
    decisive steps -- and bit-identical when the geometry is the same (two shards of equal size)."""
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 33, "sharp"))
    g = ctx_cache(cfgs, packed)
    codes = synth.synth_codes_tree(6, 21, 200, 77)
    full = g.rollout_argmax(torch.from_numpy(codes), None, want_trace=True)
    fl, fm, fg = full["logits"].cpu().numpy(), full["merges"].cpu().numpy(), full["top2_gap"].cpu().numpy()
    halves = []
    for lo in (0, 3):
        r = g.rollout_argmax(torch.from_numpy(codes[lo:lo + 3]), None, want_trace=True)
        halves.append((r["logits"].cpu().numpy(), r["merges"].cpu().numpy()))
        scale = np.abs(fl[lo:lo + 3]).max()
        assert_logits_close(halves[-1][0], fl[lo:lo + 3], RTOL, f"shard {lo}")
        decisive = fg[lo:lo + 3] > 4 * RTOL * scale
        assert (halves[-1][1][decisive] == fm[lo:lo + 3][decisive]).all()
    again = g.rollout_argmax(torch.from_numpy(codes[0:3]), None, want_trace=True)
    assert np.array_equal(again["logits"].cpu().numpy(), halves[0][0])          # same geometry: same bits
    assert np.array_equal(again["merges"].cpu().numpy(), halves[0][1])


def test_operands_beyond_the_fp16_piece_range_are_reported(ctx_cache):
    """The f16x3 GEMMs overflow where fp32 would not (|x| > 65504): the table kernel flags non-finite
    scores and check_numeric raises instead of a wrong tree being returned silently; the flag clears."""
    z, cfgs, packed = load_golden("synth_b2_t8_l128_s0")
    g = ctx_cache(cfgs, packed)
    o = _oracle(cfgs, packed)
    state = o.encode(onehot_f32(z["codes"]), z["mask"])
    mask = torch.from_numpy(z["mask"])
    g.pair_scores_full(torch.from_numpy(state), mask)
    g.check_numeric()                                          # ordinary magnitudes: fine
    g.pair_scores_full(torch.from_numpy(state * 1e6), mask)
    with pytest.raises(FloatingPointError):
        g.check_numeric()
    g.pair_scores_full(torch.from_numpy(state), mask)
    g.check_numeric()                                          # the flag was cleared by the read


def test_unsupported_shapes_fail_loudly(ctx_cache):
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
    g = ctx_cache(cfgs, packed)
    with pytest.raises(RuntimeError, match="256 rows"):
        g.rollout_argmax(torch.zeros(1, 257, 64, dtype=torch.uint8))
    from neuralnj_amd._lib import Nnj
    c4 = utils.shipped_config()
    c4.model.patch_size = 4                                 # tokens of four sites: 30 sites are not a whole number of them
    g4 = Nnj(c4, "cuda:0")
    g4.load_weights(weights.pack(c4, weights.seeded_state(c4, 0, "sharp")))
    with pytest.raises(RuntimeError, match="multiple of patch_size"):
        g4.rollout_argmax(torch.zeros(1, 4, 30, dtype=torch.uint8))
    g4.close()
    c32 = utils.shipped_config()
    c32.model.embed_dim = 32                                # 32 features in 8 heads: heads of 4 features are not covered
    with pytest.raises(RuntimeError, match="embed_dim 8..64"):
        Nnj(c32, "cuda:0")
    g2 = Nnj(cfgs, "cuda:0")
    with pytest.raises(RuntimeError, match="weights not loaded"):
        g2.encode(torch.zeros(1, 4, 32, dtype=torch.uint8))
    g2.close()


def test_model_env_api_follows_reference_call_sequence():
    """model.PhyloATTN / environment.PhyInferEnv used exactly like the reference's
    reinforce_rollout uses them, and the fast path: same merge list and Newick as golden."""
    from neuralnj_amd.environment import PhyInferEnv
    from neuralnj_amd.model import PhyloATTN
    from neuralnj_amd.rollout import argmax_rollout, reinforce_rollout_argmax
    z, cfgs, packed = load_golden("synth_b2_t8_l128_s0")
    agent = PhyloATTN(cfgs)
    sd = weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"]))
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    agent = agent.to("cuda:0")
    B, T, L = z["codes"].shape
    batch = {"data": torch.from_numpy(synth.codes_to_onehot(z["codes"])),
             "seqs": [synth.codes_to_seqs(z["codes"][b]) for b in range(B)],
             "seq_keys": [list(k) for k in z["keys"]],
             "seq_weights": torch.from_numpy((~z["mask"]).astype(np.float32))}
    env = PhyInferEnv(cfgs, "cuda:0")
    scores, best, merges = reinforce_rollout_argmax(batch, agent, env)
    assert np.array_equal(merges, z["merges"])
    assert [s.subtrees[0].utree_op_str for s in env.states] == [str(x) for x in z["newick"]]
    assert best == str(z["best_tree"])
    env2 = PhyInferEnv(cfgs, "cuda:0")
    _, best2, merges2 = argmax_rollout(batch, agent, env2)
    assert np.array_equal(merges2, z["merges"]) and best2 == best
    # arbitrary (non one-hot) float input goes through the on-device embed MLP, like the reference's .float()
    soft = torch.rand(2, 8, 128, 4)
    with torch.no_grad():                   # (with gradients enabled the Finetune operators would run instead)
        got = agent.encode_zxr(soft, torch.zeros(2, 128, dtype=torch.bool)).cpu().numpy()
    from oracle_lib import Oracle
    ref = Oracle(cfgs, packed).encode(soft.numpy(), np.zeros((2, 128), bool))
    np.testing.assert_allclose(got, ref, atol=RTOL * np.abs(ref).max())
    onehot = torch.from_numpy(synth.codes_to_onehot(z["codes"]).astype(np.float32))
    a = agent._context().encode_onehot(onehot, None)
    b = agent._context().encode(torch.from_numpy(z["codes"]), None)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-5)   # LUT path == MLP path


def test_reference_default_shape_through_the_model_api():
    """The reference's own default model shape (utils.py:45-52: patch 4, 32 features, 4 heads, 3 layers) through
    model.PhyloATTN / environment.PhyInferEnv and through the fused step API: merge list, Newick string and tables of
    the reference's golden run."""
    from neuralnj_amd.environment import PhyInferEnv
    from neuralnj_amd.model import PhyloATTN
    from neuralnj_amd.rollout import argmax_rollout, reinforce_rollout_argmax
    z, cfgs, packed = load_golden("dim32_b2_t10_l256_s19")
    assert (cfgs.model.embed_dim, cfgs.model.num_enc_heads, cfgs.model.patch_size, cfgs.model.num_enc_layers) == (32, 4, 4, 3)
    agent = PhyloATTN(cfgs)
    sd = weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"]))
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    agent = agent.to("cuda:0")
    B, T, L = z["codes"].shape
    batch = {"data": torch.from_numpy(synth.codes_to_onehot(z["codes"])),
             "seqs": [synth.codes_to_seqs(z["codes"][b]) for b in range(B)],
             "seq_keys": [list(k) for k in z["keys"]],
             "seq_weights": torch.from_numpy((~z["mask"]).astype(np.float32))}
    env = PhyInferEnv(cfgs, "cuda:0")
    _, best, merges = reinforce_rollout_argmax(batch, agent, env)
    assert np.array_equal(merges, z["merges"])
    assert [s.subtrees[0].utree_op_str for s in env.states] == [str(x) for x in z["newick"]]
    assert env.state_tensor is None or env.state_tensor.shape[-1] == 32
    env2 = PhyInferEnv(cfgs, "cuda:0")
    _, best2, merges2 = argmax_rollout(batch, agent, env2)
    assert np.array_equal(merges2, z["merges"]) and best2 == best == str(z["best_tree"])
    # the fused step API (nnj_step) on dense tensors of 32 features
    g = agent._context()
    tm = torch.from_numpy(z["mask"])
    state = g.encode(torch.from_numpy(z["codes"]), tm)
    assert state.shape == (B, T, L // 4, 32)
    np.testing.assert_allclose(state.cpu().numpy(), z["enc"], atol=RTOL * np.abs(z["enc"]).max())
    tabs = split_trace(z["logits"], T)
    logits = g.pair_scores_full(state, tm)
    assert_logits_close(logits.cpu().numpy(), tabs[0], RTOL, "pair_scores_full")
    ij, _ = g.select_pair(logits, T)
    for step, n in enumerate(range(T - 1, 1, -1)):
        assert np.array_equal(ij.cpu().numpy(), z["merges"][:, step])
        r = g.step(state, tm, ij, logits)
        state, logits, ij = r["state"], r["logits"], r["ij"]
        assert state.shape == (B, n, L // 4, 32)
        assert_logits_close(logits.cpu().numpy(), tabs[step + 1], RTOL, f"nnj_step n={n}")


def test_drop_in_loop_hands_the_table_over_only_when_asked_the_same_question():
    """The reference's loop through model / environment at a size that crosses every tile tier (50 rows down to 2, 16
    alignments): env.step + decode_zxr as one device step (PhyloATTN.fused_env_step) give the merges of the single-call
    rollout; a decode_zxr that does NOT ask the question the fused step answered (a copy of the previous table, another
    mask object, other merged pairs) gets the ordinary evaluation, which agrees with the handed-over table."""
    from neuralnj_amd.environment import PhyInferEnv
    from neuralnj_amd.model import PhyloATTN
    from neuralnj_amd.rollout import argmax_rollout, reinforce_rollout_argmax
    cfgs = utils.shipped_config()
    agent = PhyloATTN(cfgs)
    sd = weights.seeded_state(cfgs, 3, "sharp")
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    agent = agent.to("cuda:0").eval()
    B, T, L = 16, 50, 256
    codes = synth.synth_codes_tree(B, T, L, 77)
    batch = {"data": torch.from_numpy(synth.codes_to_onehot(codes)), "seqs": [[""] * T for _ in range(B)],
             "seq_keys": [[f"t{i}" for i in range(T)] for _ in range(B)],
             "seq_weights": torch.ones((B, L), dtype=torch.float32)}
    _, best, merges = reinforce_rollout_argmax(batch, agent, PhyInferEnv(cfgs, "cuda:0"))
    _, best2, merges2 = argmax_rollout(batch, agent, PhyInferEnv(cfgs, "cuda:0"))
    assert np.array_equal(merges, merges2) and best == best2
    # the hand-over and its refusals, on the first two steps
    dev = "cuda:0"
    mask = torch.zeros((B, L), dtype=torch.bool, device=dev)
    env = PhyInferEnv(cfgs, dev)
    with torch.no_grad():
        st0 = agent.encode_zxr(batch["data"].to(dev), mask)
        env.init_states(batch["seqs"], batch["seq_keys"], batch["data"])
        env.state_tensor = st0
        l0 = agent.decode_zxr(st0, mask, (None, None, None))["logits"]
        act = torch.argmax(l0, dim=-1)
        ij = torch.tensor([env.tree_pairs_dict[T][a] for a in act.tolist()], dtype=torch.int32, device=dev)
        env.step(act, [(None, None)] * B, agent=agent)
        st1 = env.state_tensor
        assert agent._prefetch is not None and agent._prefetch[0] is st1
        pf = agent._prefetch
        other = ij.clone()
        other[0] = torch.tensor([0, 1] if tuple(ij[0].tolist()) != (0, 1) else [0, 2], dtype=torch.int32)
        for kind, args in (("copy of the table", (st1, mask, (ij, None, l0.clone()))),
                           ("another mask object", (st1, mask.clone(), (ij, None, l0))),
                           ("other pairs", (st1, mask, (other, None, l0)))):
            agent._prefetch = pf
            got = agent.decode_zxr(*args)["logits"]
            assert got is not pf[6], kind
            if kind != "other pairs":
                assert_logits_close(got.cpu().numpy(), pf[6].cpu().numpy(), 0.2 * RTOL, kind)
        agent._prefetch = pf
        assert agent.decode_zxr(st1, mask, (ij, None, l0))["logits"] is pf[6]
    agent._context().check_numeric()


def test_aggregate_pairwise_form_matches_oracle():
    """PhyloATTN.aggregate(batchwise_ij_indices=False) (reference model.py:102-155 as decode_gg calls it): N pairs per
    alignment, 1-D indices (the same pairs for every alignment) and 2-D indices with rows that are NOT rows of the state."""
    from neuralnj_amd.model import PhyloATTN
    from oracle_lib import Oracle
    z, cfgs, packed = load_golden("synth_b2_t8_l128_s1")
    agent = PhyloATTN(cfgs)
    sd = weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"]))
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    agent = agent.to("cuda:0").eval()
    o = Oracle(cfgs, packed)
    B, T, L = z["codes"].shape
    state = o.encode(onehot_f32(z["codes"]), z["mask"])
    st = torch.from_numpy(state).to("cuda:0")
    tm = torch.from_numpy(z["mask"]).to("cuda:0")
    with torch.no_grad():
        agent.decode_zxr(st, tm, (None, None, None))                     # stashes the state, like the reference
        row, col = torch.triu_indices(T, T, offset=1)
        row, col = row[:9], col[:9]
        got = agent.aggregate(st[:, row], st[:, col], (row, col)).cpu().numpy()
        assert got.shape == (B, 9, L, 64)
        for k in range(9):
            ij = np.tile(np.array([[int(row[k]), int(col[k])]], np.int32), (B, 1))
            ref = o.aggregate(state, ij)
            np.testing.assert_allclose(got[:, k:k + 1], ref, atol=RTOL * np.abs(ref).max())
        # 2-D indices, rows of the pairs given explicitly (not rows of the state)
        rng = np.random.default_rng(5)
        ii = np.array([[0, 2, 5], [1, 3, 0]], np.int64)
        jj = np.array([[4, 7, 6], [2, 6, 7]], np.int64)
        xi = (state[:, :3] * 0.5 + rng.standard_normal(state[:, :3].shape).astype(np.float32) * 0.1).astype(np.float32)
        xj = (state[:, 3:6] * 0.5 + rng.standard_normal(state[:, :3].shape).astype(np.float32) * 0.1).astype(np.float32)
        got = agent.aggregate(torch.from_numpy(xi).to("cuda:0"), torch.from_numpy(xj).to("cuda:0"),
                              (torch.from_numpy(ii), torch.from_numpy(jj))).cpu().numpy()
        for k in range(3):
            s_k = state.copy()
            for b in range(B):
                s_k[b, ii[b, k]] = xi[b, k]
                s_k[b, jj[b, k]] = xj[b, k]
            ref = o.aggregate(s_k, np.stack([ii[:, k], jj[:, k]], 1).astype(np.int32))
            np.testing.assert_allclose(got[:, k:k + 1], ref, atol=RTOL * np.abs(ref).max())
        with pytest.raises(ValueError):
            agent.aggregate(None, None, (row, col))


@pytest.mark.parametrize("name", ["synth_b2_t8_l128_s1", "synth_b2_t70_l64_s12"])
def test_profiling_covers_every_launch_sequence(name, ctx_cache):
    """bench.py's per-kernel times come from HIP events around the launches (nnj_profile_enable / _read): every launch
    sequence -- rollouts (two-pass up to 64 rows, star kernels above), the step API, the dense-state entry points --
    must leave well-formed event pairs (profiling scopes do not nest)."""
    z, cfgs, packed = load_golden(name)
    g = ctx_cache(cfgs, packed)
    codes, mask = torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"])
    B, T, L = z["codes"].shape
    g.profile_enable(True)
    try:
        r = g.rollout_argmax(codes, mask)
        merges = r["merges"].cpu().numpy()
        u = torch.rand(B, T - 1)
        g.rollout_sample(codes, mask, u)
        _step_loop(g, codes, mask, T)
        state = g.encode(codes, mask)
        logits = g.pair_scores_full(state, mask)
        ij = torch.from_numpy(merges[:, 0].copy())
        g.aggregate(state, ij)
        s1 = g.env_step(state, ij)
        g.pair_scores_incr(s1, mask, ij, logits)
        torch.cuda.synchronize()
        prof = g.profile_read()
    finally:
        g.profile_enable(False)
    assert sum(cnt for _, cnt in prof.values()) > 4 * T and all(ms >= 0.0 for ms, _ in prof.values())


def _certify_sampled(ref, merges, logits_g, u, temperature, T, rtol=RTOL):
    """Sampled trajectories against the oracle twin (same uniforms): identical, except where the uniform lands within
    fp32 rounding of a CDF boundary of the oracle's own table -- at the first divergent step of a differing trajectory
    the oracle's cumulative mass between the two picks must be within the tables' rounding of the target u * total
    (neighbours in the flat pair order, or separated by entries of negligible probability only).  Returns the
    per-trajectory agreement flags."""
    agree = (ref["merges"] == merges).all(axis=(1, 2))
    tabs_o = split_trace(ref["logits"], T)
    tabs_g = split_trace(logits_g, T)
    for b in np.nonzero(~agree)[0]:
        s_ = int(np.argmax(np.any(ref["merges"][b] != merges[b], axis=1)))
        n = T - s_
        to = tabs_o[s_][b].astype(np.float64)
        assert_logits_close(tabs_g[s_][b], tabs_o[s_][b], rtol, "table at the divergent step")
        e = np.exp((to - to.max()) / temperature)
        cdf = np.cumsum(e)
        k_o = flat_pair(n, *ref["merges"][b, s_])
        k_g = flat_pair(n, *merges[b, s_])
        target = float(u[b, s_]) * cdf[-1]
        # table entries within rtol * scale move every CDF value by at most that (relative, / temperature)
        slack = (rtol * max(float(np.abs(to).max()), 1.0) / temperature) * cdf[-1] * 2
        # the oracle's pick is the first index whose cumulative mass exceeds the target; a later pick means the mass up
        # to the entry before it must drop below the target, an earlier one that its own cumulative mass must rise above
        miss = cdf[k_g - 1] - target if k_g > k_o else target - cdf[k_g]
        assert 0 <= miss <= slack, f"trajectory {b} step {s_}: not a CDF-boundary case ({k_o} vs {k_g}, {miss:.3e} > {slack:.3e})"
    return agree


def test_sampling_mode_matches_oracle_and_replicates(ctx_cache):
    """nnj_rollout_sample (NeuralNJ-MC device part, finetune_rl_search.py:147): same sampled merge lists
    as the oracle for the same uniforms; encode-once-and-replicate equals B explicit copies bit for bit;
    a cold temperature reproduces the argmax rollout."""
    z, cfgs, packed = load_golden("synth_b1_t20_l256_s0")
    g = ctx_cache(cfgs, packed)
    o = _oracle(cfgs, packed)
    codes1, mask1 = z["codes"], z["mask"]
    B, T, L = 6, codes1.shape[1], codes1.shape[2]
    u = np.random.default_rng(11).random((B, T - 1)).astype(np.float32)
    codes = np.repeat(codes1, B, 0)
    mask = np.repeat(mask1, B, 0)
    r_rep = g.rollout_sample(torch.from_numpy(codes1), torch.from_numpy(mask1), u, temperature=3.0, replicas=B,
                             want_trace=True)
    r_all = g.rollout_sample(torch.from_numpy(codes), torch.from_numpy(mask), u, temperature=3.0, want_trace=True)
    assert torch.equal(r_rep["merges"], r_all["merges"]) and torch.equal(r_rep["logits"], r_all["logits"])
    merges = r_all["merges"].cpu().numpy()
    assert len({tuple(m.reshape(-1)) for m in merges}) > 1            # different uniforms, different trees
    ref = o.rollout_sample(synth.codes_to_onehot(codes).astype(np.float32), mask, u, temperature=3.0)
    # identical trajectories, except where the uniform lands within fp32 rounding of a CDF boundary of the
    # oracle's own table: at the first divergent step of a differing trajectory the two picks must be
    # NEIGHBOURS in the flat pair order and the target u * total within the tables' rounding of the boundary
    agree = _certify_sampled(ref, merges, r_all["logits"].cpu().numpy(), u, 3.0, T)
    # (under the sharpened weights a table's scale is ~400: fp32 rounding of 1e-4 of it moves the CDF by up to 2.7 % of
    # the mass per step, so a good part of 19-step trajectories meets a boundary case -- each one certified above.  The
    # reference-scale case below is where agreement itself is asserted.)
    assert agree.mean() >= 0.5
    assert_logits_close(r_all["logits"].cpu().numpy()[agree], ref["logits"][agree], RTOL, "sampled tables")
    # reference-scale ("plain") weights: the tables' rounding is 1e-6 of a small scale, a boundary case is rare --
    # at least 90 % of 48 trajectories identical to the oracle twin, every other one certified (VERDICT r3 "weak" 6)
    zp, cfgp, packp = load_golden("plain_b1_t12_l96_s4")
    gp, op = ctx_cache(cfgp, packp), _oracle(cfgp, packp)
    Bp, Tp = 48, zp["codes"].shape[1]
    up = np.random.default_rng(12).random((Bp, Tp - 1)).astype(np.float32)
    rp = gp.rollout_sample(torch.from_numpy(zp["codes"]), torch.from_numpy(zp["mask"]), up, temperature=1.0, replicas=Bp,
                           want_trace=True)
    refp = op.rollout_sample(synth.codes_to_onehot(np.repeat(zp["codes"], Bp, 0)).astype(np.float32),
                             np.repeat(zp["mask"], Bp, 0), up, temperature=1.0)
    agree_p = _certify_sampled(refp, rp["merges"].cpu().numpy(), rp["logits"].cpu().numpy(), up, 1.0, Tp)
    assert agree_p.mean() >= 0.9, f"only {agree_p.sum()} of {Bp} sampled trajectories equal the oracle's under plain weights"
    assert len({tuple(m.reshape(-1)) for m in rp["merges"].cpu().numpy()}) > 4
    cold = g.rollout_sample(torch.from_numpy(codes1), torch.from_numpy(mask1), u[:1], temperature=1e-4)
    assert np.array_equal(cold["merges"].cpu().numpy(), z["merges"])
    with pytest.raises(RuntimeError):
        g.rollout_sample(torch.from_numpy(codes1), None, u[:1], temperature=0.0)


def test_argmax_inference_file_to_tree(tmp_path):
    """The reference's Argmax_inference (finetune_rl_search.py:478-509): list the .phy files of a directory, replicate
    each alignment env.batch_size times, roll out, write <name>.tre -- through neuralnj_amd.rollout.argmax_inference
    (file -> 1-byte codes -> HIP -> Newick) on PHYLIP files written from two golden alignments.  The trees written
    must be the reference's own (RF = 0 against the golden Newick; the strings themselves are equal)."""
    from neuralnj_amd.rollout import argmax_inference
    names = ["synth_b1_t20_l256_s0", "ragged_b2_t9_l30_s8"]
    cfgs = None
    want = {}
    d = tmp_path / "msas"
    d.mkdir()
    for nm in names:
        z, cfgs_, packed = load_golden(nm)
        if nm == names[0]:
            cfgs = cfgs_
        keys = [str(k) for k in z["keys"][0]]
        seqs = synth.codes_to_seqs(z["codes"][0])
        lines = [f"{len(keys)} {len(seqs[0])}"]
        # taxa written in REVERSE order: the loader must sort them by numeric suffix (phydata.py:1252-1262)
        lines += [f"{k} {s_}" for k, s_ in reversed(list(zip(keys, seqs)))]
        (d / f"{nm}.phy").write_text("\n".join(lines) + "\n")
        want[nm] = (int(z["wseed"]), str(z["style"]), str(z["newick"][0]))
    (d / "notes.txt").write_text("not an alignment\n")
    for nm in names:
        wseed, style, newick = want[nm]
        sub = tmp_path / nm
        sub.mkdir()
        (sub / f"{nm}.phy").write_text((d / f"{nm}.phy").read_text())
        c = cfgs.clone()
        c.env.batch_size = 3                        # replicas of the same alignment (Agmax_one_instance :435-444)
        c.reload_checkpoint_path = str(tmp_path / f"{nm}.pt")
        sd = weights.seeded_state(c, wseed, style)
        torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}}, c.reload_checkpoint_path)
        out_dir = tmp_path / f"out_{nm}"
        for fast in (True, False):
            res = argmax_inference(c, str(sub), str(out_dir), device="cuda:0", fast=fast)
            assert list(res) == [f"{nm}.phy"]
            written = (out_dir / f"{nm}.tre").read_text()
            assert utils.rf_distance(written, newick)[0] == 0
            assert written == newick


def test_sampler_first_step_frequencies_chi_square(ctx_cache):
    """Distributional pin of the sampling mode (finetune_rl_search.py:147, Categorical(logits / temperature)): 8192
    replicas of one 8-taxon alignment, i.i.d. uniforms -> the first-step picks follow softmax(table0 / T).
    Pearson chi-square over the 28 pairs (27 degrees of freedom: the 99.9 % quantile is 55.5)."""
    z, cfgs, packed = load_golden("synth_b1_t8_l128_s1")
    g = ctx_cache(cfgs, packed)
    B, T = 8192, 8
    temp = 12.0
    u = np.random.default_rng(2024).random((B, T - 1)).astype(np.float32)
    r = g.rollout_sample(torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"]), u, temperature=temp, replicas=B,
                         want_trace=True)
    m = r["merges"].cpu().numpy()
    t0 = r["logits"].cpu().numpy()[0, :28].astype(np.float64)
    p = np.exp((t0 - t0.max()) / temp)
    p /= p.sum()
    picks = np.array([flat_pair(T, i, j) for i, j in m[:, 0]])
    obs = np.bincount(picks, minlength=28).astype(np.float64)
    exp_ = p * B
    assert exp_.min() > 5, "expected counts too small for a chi-square test: raise the temperature"
    chi2 = float(((obs - exp_) ** 2 / exp_).sum())
    assert chi2 < 55.5, f"first-step frequencies do not follow softmax(logits / T): chi2 = {chi2:.1f}"
    # and the pick is exactly the inverse CDF of the table the kernel wrote (fp64, flat order)
    cdf = np.cumsum(np.exp((t0 - t0.max()) / temp))
    want = np.searchsorted(cdf, u[:, 0].astype(np.float64) * cdf[-1], side="right")
    near = np.abs(cdf[np.minimum(want, 27)] - u[:, 0] * cdf[-1]) < 1e-9 * cdf[-1]
    assert ((picks == want) | near).all()


def test_topology_keys_match_topo_repr(ctx_cache):
    """nnj_topology_hash (duplicate filter of the sampling mode, utils.py:76): equal 64-bit keys <=> equal topo_repr
    strings, on sampled rollouts with many repeats; rollout.sample_rollouts returns each distinct tree once."""
    from neuralnj_amd.environment import PhyInferEnv
    z, cfgs, packed = load_golden("synth_b1_t8_l128_s0")
    g = ctx_cache(cfgs, packed)
    B, T = 600, 8
    u = np.random.default_rng(5).random((B, T - 1)).astype(np.float32)
    r = g.rollout_sample(torch.from_numpy(z["codes"]), torch.from_numpy(z["mask"]), u, temperature=8.0, replicas=B)
    keys = g.topology_hash(r["merges"]).cpu().numpy()
    merges = r["merges"].cpu().numpy()
    env = PhyInferEnv(cfgs, "cpu")
    names = [str(k) for k in z["keys"][0]]
    env.init_states([[""] * T] * B, [names] * B, None)
    env.apply_merges(merges)
    topo = np.array([s.subtrees[0].topo_repr for s in env.states])
    assert 1 < len(set(topo)) < B
    by_key, by_topo = {}, {}
    for k, t in zip(keys, topo):
        assert by_key.setdefault(int(k), t) == t            # one topology per key
        assert by_topo.setdefault(t, int(k)) == int(k)      # one key per topology
    # two merge ORDERS of the same tree get the same key: ((0,1),(2,3)) joined in either order
    a = np.array([[[0, 1], [1, 2], [0, 1]]], np.int32)      # (0,1) first, then (2,3), then root
    b = np.array([[[2, 3], [0, 1], [0, 1]]], np.int32)      # (2,3) first, then (0,1), then root
    ka, kb = g.topology_hash(a).item(), g.topology_hash(b).item()
    assert ka == kb
    c = np.array([[[0, 2], [1, 2], [0, 1]]], np.int32)      # (0,2) | (1,3): another tree
    assert g.topology_hash(c).item() != ka


def test_sample_rollouts_returns_distinct_trees():
    from neuralnj_amd.environment import PhyInferEnv
    from neuralnj_amd.model import PhyloATTN
    from neuralnj_amd.rollout import sample_rollouts
    z, cfgs, packed = load_golden("synth_b1_t8_l128_s0")
    agent = PhyloATTN(cfgs)
    sd = weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"]))
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    agent = agent.to("cuda:0")
    batch = {"data": torch.from_numpy(synth.codes_to_onehot(z["codes"])),
             "seqs": [synth.codes_to_seqs(z["codes"][0])], "seq_keys": [[str(k) for k in z["keys"][0]]],
             "seq_weights": torch.from_numpy((~z["mask"]).astype(np.float32))}
    env = PhyInferEnv(cfgs, "cuda:0")
    trees, merges = sample_rollouts(batch, agent, env, 256, seed=3, temperature=8.0)
    assert sum(c for _, c in trees) == 256 and len(trees) > 1
    # pairwise distinct topologies, and every rollout's tree is in the list
    env2 = PhyInferEnv(cfgs, "cpu")
    env2.init_states([[""] * 8] * 256, [batch["seq_keys"][0]] * 256, None)
    env2.apply_merges(merges)
    all_topo = [s.subtrees[0].topo_repr for s in env2.states]
    assert len(trees) == len(set(all_topo))
    assert {nwk for nwk, _ in trees} == {s.subtrees[0].utree_op_str for s in env2.states} or len(trees) == len(set(all_topo))


def test_search_round_samples_scores_and_ranks():
    """rollout.search_rollouts: the device part of one RL_Search round (finetune_rl_search.py:338-427) -- sample,
    drop duplicate topologies, optimise branch lengths, rank by log-likelihood.  Checks the plumbing: the trees come
    back best first with real branch lengths, the scores equal an independent numpy evaluation of the returned Newick's
    merge list, and more rollouts never give a worse best tree."""
    import re
    from neuralnj_amd.environment import PhyInferEnv
    from neuralnj_amd.model import PhyloATTN
    from neuralnj_amd.rollout import search_rollouts
    z, cfgs, packed = load_golden("synth_b1_t8_l128_s0")
    agent = PhyloATTN(cfgs)
    sd = weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"]))
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    agent = agent.to("cuda:0")
    batch = {"data": torch.from_numpy(synth.codes_to_onehot(z["codes"])),
             "seqs": [synth.codes_to_seqs(z["codes"][0])], "seq_keys": [[str(k) for k in z["keys"][0]]],
             "seq_weights": torch.from_numpy((~z["mask"]).astype(np.float32))}
    best, best_ll, trees = search_rollouts(batch, agent, PhyInferEnv(cfgs, "cuda:0"), 64, seed=1, temperature=6.0)
    lls = [t[1] for t in trees]
    assert lls == sorted(lls, reverse=True) and best_ll == lls[0] and best == trees[0][0]
    assert sum(t[2] for t in trees) == 64 and len(trees) > 1
    lens = [float(x) for x in re.findall(r":([0-9.eE+-]+)", best)]
    assert len(lens) == 2 * 8 - 2 and all(v > 0 for v in lens) and 0.12345 not in lens
    assert len({utils.rf_distance(t[0], best)[0] for t in trees}) > 1          # really different topologies
    _, ll_more, _ = search_rollouts(batch, agent, PhyInferEnv(cfgs, "cuda:0"), 256, seed=1, temperature=6.0)
    assert ll_more >= best_ll - 1e-6                    # the first 64 uniforms are a prefix of the 256


def test_search_round_at_config5_200x4096_end_to_end():
    """VERDICT r3 item 4 -- one COMPLETE round of the reference's RL_Search (finetune_rl_search.py:338-427) at BASELINE
    configs[4]: 8 sampled rollouts of one 200 x 4096 alignment (encoded once), duplicate topologies dropped by the device
    keys, GTR+I+G parameters estimated (`opt_model=True`, environment.py:373-377), branch lengths optimised in three
    sweeps (iters=3) and every distinct tree scored on the GPU, trees returned best first.  The ranked log-likelihoods
    must equal oracle/lik_oracle.py (numpy / scipy, sites as an array axis) on the returned merge lists, branch lengths
    and model -- every tree of the round, not one."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import lik_oracle as LO
    from neuralnj_amd.environment import PhyInferEnv
    from neuralnj_amd.model import PhyloATTN
    from neuralnj_amd.rollout import search_rollouts
    cfgs = utils.shipped_config()
    agent = PhyloATTN(cfgs)
    sd = weights.seeded_state(cfgs, 3, "plain")
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    agent = agent.to("cuda:0")
    T, L, R = 200, 4096, 8
    codes = synth.synth_codes_tree(1, T, L, seed=4242)
    batch = {"codes": torch.from_numpy(codes), "seqs": [[""] * T], "seq_keys": [[f"taxon{i + 1}" for i in range(T)]],
             "seq_weights": torch.ones((1, L), dtype=torch.float32)}
    det = {}
    best, best_ll, trees = search_rollouts(batch, agent, PhyInferEnv(cfgs, "cuda:0"), R, seed=5, temperature=1.0,
                                           model="auto", sweeps=3, details=det)
    assert sum(t[2] for t in trees) == R and 1 <= len(trees) <= R
    lls = [t[1] for t in trees]
    assert lls == sorted(lls, reverse=True) and best_ll == lls[0] and best == trees[0][0]
    assert all(np.isfinite(lls)) and lls[0] < 0
    m = det["model"]
    model = dict(rates=[m.rates[k] for k in range(6)], freqs=[m.freqs[k] for k in range(4)], alpha=float(m.alpha),
                 pinv=float(m.pinv), ncat=int(m.ncat))
    assert model["alpha"] > 0 and 0 <= model["pinv"] < 1
    for k in range(len(trees)):
        want = LO.tree_loglik_sites(codes[0], det["merges"][k], det["brlen"][k].astype(np.float64), model)
        assert abs(float(det["loglik"][k]) - want) <= 1e-9 * abs(want), (k, float(det["loglik"][k]), want)
        assert (det["brlen"][k] > 0).all()
    # the Newick strings carry the same trees: topology keys of the strings' merge lists == the scored ones
    for k in (0, len(trees) - 1):
        m2, _ = utils.newick_to_merges(trees[k][0], batch["seq_keys"][0])
        ctx = agent._context()
        keys = ctx.topology_hash(torch.from_numpy(np.stack([m2, det["merges"][k]]).astype(np.int32)).to("cuda:0")).cpu()
        assert int(keys[0]) == int(keys[1])


def test_graph_replay_sees_new_inputs():
    """Round 4 regression: a small-batch rollout whose arguments all repeat is replayed from a hipGraph (include/nnj.h,
    nnj_rollout_argmax).  With the path's three hipMemsetAsync calls captured as memset nodes every second replay returned
    wrong tables for new data in the same buffers (>= 256 sites) -- bench.py's single-alignment leg replays ONE input and
    never saw it.  Fixed buffers on a side stream (the allocator pattern that makes every argument repeat), new data per
    call, against a handle that never replays (NNJ_GRAPH=0): state, tables and merges bit for bit on every call."""
    import os
    from neuralnj_amd._lib import Nnj
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
    g = Nnj(cfgs, "cuda:0")
    g.load_weights(packed)
    os.environ["NNJ_GRAPH"] = "0"
    try:
        ref = Nnj(cfgs, "cuda:0")
    finally:
        del os.environ["NNJ_GRAPH"]
    ref.load_weights(packed)
    st = torch.cuda.Stream()
    #  also with the handle's sub-batch streams set to 1 and to 3 (a graph is captured from the caller's stream:
    # a fork / join inside the capture would need its own edges), and with a caller-supplied site mask in a fixed buffer
    for (B, T, L, streams, masked) in ((1, 34, 256, 2, False), (2, 50, 512, 2, False), (2, 50, 512, 1, False),
                                       (3, 20, 256, 3, False), (2, 34, 256, 2, True)):
        g.set_concurrency(streams)
        ref.set_concurrency(streams)
        buf = torch.empty((B, T, L), dtype=torch.uint8, device="cuda:0")
        mbuf = torch.zeros((B, L), dtype=torch.uint8, device="cuda:0") if masked else None
        for it in range(5):
            cn = synth.synth_codes(B, T, L, seed=70 + it, gap_frac=0.2)
            mk = None
            if masked:                                   # a padded tail whose length changes from call to call
                mk = np.zeros((B, L), np.uint8)
                mk[:, L - 3 - 2 * it:] = 1
                cn[:, :, L - 3 - 2 * it:] = 5
                mbuf.copy_(torch.from_numpy(mk))
            c = torch.from_numpy(cn).cuda()
            buf.copy_(c)
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                r = g.rollout_argmax(buf, mbuf, want_trace=True, want_state=True)
                cur = {k: v.cpu() for k, v in r.items()}
            del r
            want = {k: v.cpu() for k, v in ref.rollout_argmax(c, None if mk is None else torch.from_numpy(mk).cuda(),
                                                              want_trace=True, want_state=True).items()}
            for k in ("state", "logits", "merges"):
                assert torch.equal(cur[k], want[k]), \
                    f"call {it} ({B} x {T} x {L}, streams {streams}, masked {masked}): {k} differs from the graph-free handle"
    g.close()
    ref.close()


def test_small_magnitude_weights(ctx_cache):
    """the f16x3 operand split keeps 22 significand bits only while the low piece of an operand stays a
    NORMAL fp16 number; weights of magnitude 1e-3..1e-2 push it into the fp16 denormal range.  Every weight matrix of
    the reference-scale ("plain") initialisation is scaled by 0.05 here (weights of 6e-3 and below): the score
    tables must still hold the 1e-4 tolerance against the oracle, and the trees must be the oracle's."""
    cfgs = utils.shipped_config()
    st = weights.seeded_state(cfgs, 31, "plain")
    st = {k: (v * np.float32(0.05) if v.ndim == 2 else v) for k, v in st.items()}
    packed = weights.pack(cfgs, st)
    g = ctx_cache(cfgs, packed)
    o = _oracle(cfgs, packed)
    for (B, T, L, seed) in ((2, 12, 96, 1), (1, 40, 256, 2), (1, 80, 64, 3)):
        codes = synth.synth_codes_tree(B, T, L, seed)
        r = g.rollout_argmax(torch.from_numpy(codes), None, want_trace=True, want_state=True)
        merges = r["merges"].cpu().numpy()
        ref = o.rollout_argmax(onehot_f32(codes), None, forced_merges=merges, want_state=True)
        np.testing.assert_allclose(r["state"].cpu().numpy(), ref["state"], atol=RTOL * np.abs(ref["state"]).max())
        assert_logits_close(r["logits"].cpu().numpy(), ref["logits"], RTOL, f"small weights {B}x{T}x{L}")
        scale = np.abs(ref["logits"]).max()
        decisive = ref["top2_gap"] > 4 * RTOL * max(scale, 1.0)
        assert (ref["merges"][decisive] == merges[decisive]).all()


@pytest.mark.parametrize("style", ["plain", "sharp"])
def test_config5_200x4096_matches_fp64_golden(style, ctx_cache):
    """BASELINE configs[4] at the FULL size against the float64 oracle (tests/golden/gen_cfg5_f64.py: a free Argmax
    run of one 200 x 4096 alignment, 22 minutes of CPU per weight style): the HIP rollout teacher-forced along the
    stored merges against the complete score tables of eight sampled steps (200 ... 3 rows live: star kernels, the
    65 -> 64 row hand-over, the two-pass steps), and the free run against the stored merge list.
      plain (reference-scale weights): every table within 1e-4 of its scale.
      sharp (the stress weights of the other fixtures): six encoder layers amplify fp32 rounding of the encoder output
      about twenty-fold into the tables, so NO fp32 evaluation of this shape is within 1e-4 of the truth (the fixture
      keeps the distance of the plain-fp32 oracle -- the reference's arithmetic -- from the fp64 tables per step: 2.6e-4).
      Since round 5 the > 64-row path encodes in fp64 (csrc/nnj_encoder64.hpp) and every HIP table must be within the
      plain 1e-4 under both weight styles (the relative clause `1.25 x the fp32 oracle's noise` of round 4 is gone)."""
    import hashlib
    z = _cfg5_golden(style)
    if z is None:
        pytest.skip(f"tests/golden/cfg5_f64 fixture for {style} weights not generated")
    T, L = (int(v) for v in z["shape"])
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"])))
    assert weights.digest(packed) == str(z["weights_sha256"])
    codes = synth.synth_codes_tree(1, T, L, seed=int(z["seed"]))
    assert hashlib.sha256(codes.tobytes()).hexdigest() == str(z["codes_sha256"])
    g = ctx_cache(cfgs, packed)
    r = g.rollout_argmax(torch.from_numpy(codes), None, forced_merges=z["merges"][None], want_trace=True)
    g.check_numeric()
    tabs = split_trace(r["logits"].cpu().numpy(), T)
    rows = []
    for k, s in enumerate(int(v) for v in z["steps"]):
        ref = z[f"table_{s}"]
        scale = max(float(np.abs(ref).max()), 1.0)
        err = float(np.abs(tabs[s][0] - ref).max()) / scale
        o32 = float(z["o32_err"][k]) / scale
        rows.append((s, T - s, err, o32))
    noise = max(o for _, _, _, o in rows)                     # the fp32 oracle's worst stored table
    for s, n, err, o32 in rows:
        bound = RTOL
        assert err <= bound, f"{style} weights, step {s} ({n} rows): HIP {err:.2e} of the table's scale (fp32 oracle {o32:.2e} here, {noise:.2e} at its worst step), bound {bound:.2e}"
    print(f"200 x 4096, {style} weights, vs fp64 (step, rows, HIP, fp32 oracle): " + ", ".join(f"({s}, {n}, {e:.1e}, {o:.1e})" for s, n, e, o in rows))
    free = g.rollout_argmax(torch.from_numpy(codes), None)["merges"].cpu().numpy()[0]
    decisive = z["top2_gap"] > 4 * RTOL * float(z["scale"])
    first_bad = next((s for s in range(T - 1) if not np.array_equal(free[s], z["merges"][s])), T - 1)
    assert first_bad == T - 1 or not decisive[first_bad], f"free run leaves the fp64 merge list at decisive step {first_bad}"


def _cfg5_golden(style):
    name = "cfg5_f64_t200_l4096.npz" if style == "sharp" else f"cfg5_f64_{style}_t200_l4096.npz"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    return np.load(path) if os.path.exists(path) else None


def test_search_mode_at_config5_200x4096(ctx_cache):
    """BASELINE configs[4] in Search mode (infer_opt=Search: sampled rollouts, finetune_rl_search.py:338-427) at the
    full size: eight replicas of ONE 200 x 4096 alignment, encoded once.  (1) merge lists are valid and not all equal;
    (2) the tables of a sampled rollout are those of the Argmax rollout teacher-forced along the same merges, bit for
    bit (same kernels, same batch geometry); (3) the duplicate filter's keys agree with the merge lists; (4) on a
    160-site window of the same alignment (all 200 rows live: the oracle needs minutes for 4096 sites) the sampled
    trajectories are the oracle twin's for the same uniforms, CDF-boundary cases certified."""
    cfgs = utils.shipped_config()
    packed = weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp"))
    g = ctx_cache(cfgs, packed)
    T, L, R = 200, 4096, 8
    one = synth.synth_codes_tree(1, T, L, seed=4242)
    u = np.random.default_rng(5).random((R, T - 1)).astype(np.float32)
    rs = g.rollout_sample(torch.from_numpy(one), None, u, temperature=1.0, replicas=R, want_trace=True)
    g.check_numeric()
    merges = rs["merges"].cpu().numpy()
    assert (merges[:, :, 0] < merges[:, :, 1]).all() and (merges[:, :, 1] < np.arange(T, 1, -1)[None, :]).all()
    assert len({tuple(m.reshape(-1)) for m in merges}) > 1
    forced = g.rollout_argmax(torch.from_numpy(np.repeat(one, R, 0)), None, forced_merges=merges, want_trace=True)
    assert torch.equal(forced["logits"], rs["logits"])
    keys = g.topology_hash(rs["merges"]).cpu().numpy()
    same_tree = {tuple(m.reshape(-1)) for m in merges}
    assert len(set(keys.tolist())) <= len(same_tree)
    win = one[:, :, :160]
    o = _oracle(cfgs, packed)
    uw = u[:4]
    ref = o.rollout_sample(onehot_f32(np.repeat(win, 4, 0)), np.zeros((4, 160), bool), uw, temperature=1.0)
    got = g.rollout_sample(torch.from_numpy(win), None, uw, temperature=1.0, replicas=4, want_trace=True)
    # (200 rows under the stress weights: two fp32 evaluations differ by up to 2e-4 -- test_config5_200x4096_properties)
    # Every trajectory that leaves the twin's is certified at its first divergent step; over 199 sampled steps per
    # trajectory a CDF-boundary case somewhere is the rule, so no share of fully identical trajectories is asked for.
    _certify_sampled(ref, got["merges"].cpu().numpy(), got["logits"].cpu().numpy(), uw, 1.0, T, rtol=2 * RTOL)


def test_switchable_kernels_agree(tmp_path):
    """The kernel families that are still behind an NNJ_* switch compute the SAME rollout as the default dispatch: one child
    process per setting (the library reads its switches once), a traced Argmax rollout of three seeded 50 x 72 alignments
    (one padded), tables within 2e-5 of the default's scale and equal merge lists.  Round 5 removed the measured-and-lost
    arms of rounds 3-4 from the library (their A/B files stay under profiles/; sources in the git history and
    tools/experiments/); what is left are the two fallback families every rollout can reach in some geometry: the
    site-sharing alpha pass (NNJ_STEP_W=0 forces it for 17..48 pairs too) and the merge-weight fallback pass
    (NNJ_TWO_PASS_CAND=0: no carried candidate)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    variants = {
        "default": {},
        "site_sharing_step": {"NNJ_STEP_W": "0"},
        "no_carried_candidate": {"NNJ_TWO_PASS_CAND": "0"},
    }
    res = {}
    for name, env in variants.items():
        out = tmp_path / (name + ".npz")
        e = dict(os.environ)
        for k in ("NNJ_STEP_W", "NNJ_TWO_PASS", "NNJ_TWO_PASS_CAND"):
            e.pop(k, None)
        e.update(env)
        p = subprocess.run([sys.executable, os.path.join(here, "variant_run.py"), str(out)], env=e, capture_output=True,
                           text=True, timeout=300)
        assert p.returncode == 0, (name, p.stderr[-2000:])
        res[name] = np.load(out)
    ref = res["default"]
    scale = max(float(np.abs(ref["logits"]).max()), 1.0)
    for name, z in res.items():
        assert np.array_equal(z["merges"], ref["merges"]), name
        err = float(np.abs(z["logits"] - ref["logits"]).max()) / scale
        assert err <= 2e-5, (name, err)
