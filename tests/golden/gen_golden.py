#!/usr/bin/env python3
"""Generate golden vectors for the NeuralNJ Argmax hot path from the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference, read-only).  It imports the
reference's own model.py / environment.py / utils.py / finetune_rl_search.py and runs
`reinforce_rollout(eval=True, argmax=True, branch_optimize=False)` unmodified
(reference finetune_rl_search.py:78-189), recording what that code computes.  Nothing
of the reference is copied: the outputs below are numeric vectors and result strings.

Third-party packages the reference imports at module import time but never calls on
this path (raxmlpy's native binding, fvcore, ete3, dendropy, Bio, tensorboard) are
absent from this image; inert placeholder modules are registered in `sys.modules` so
the imports succeed.  None of their functions is executed on the Argmax path (a call
raises).  `fvcore`'s CfgNode is only an attribute container here.

Weights are NOT stored: they come from neuralnj_amd.weights.seeded_state (numpy
Philox) and are pushed into the reference model with load_state_dict(strict=True);
the fixture stores their sha256 so drift is detected.

Usage:  python tests/golden/gen_golden.py [--only NAME]
Output: tests/golden/<case>.npz
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


def _install_placeholders():
    def _never(*a, **k):
        raise RuntimeError("placeholder for an absent third-party function was called")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class CfgNode(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    mod("raxmlpy.cpp_binding", optimize_brlen=_never, compute_llh=_never, test_func=_never)
    mod("fvcore")
    mod("fvcore.common")
    mod("fvcore.common.config", CfgNode=CfgNode)
    mod("ete3", Tree=_never)
    mod("dendropy")
    bio = mod("Bio")
    phylo = mod("Bio.Phylo", read=_never)
    bt = mod("Bio.Phylo.BaseTree", Clade=type("Clade", (), {}))
    bio.Phylo = phylo
    phylo.BaseTree = bt
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        mod("torch.utils.tensorboard", SummaryWriter=_never)
    sys.path.insert(0, os.path.join(REF, "RAxMLpy"))
    sys.path.insert(0, REF)


def make_cfg(utils_mod, layers=6):
    cfgs = utils_mod.empty_config()
    cfgs.model.vocab_size = 4
    cfgs.model.patch_size = 1
    cfgs.model.embed_dim = 64
    cfgs.model.num_enc_heads = 8
    cfgs.model.num_enc_layers = layers
    cfgs.env.batch_size = 1
    return cfgs


def run_case(name, codes, mask, wseed, style, ref, keys=None, layers=6, capture_layers=False, patch=1, dim=64, heads=8):
    """codes uint8 [B,T,L]; mask bool [B,L] (True = padded site)."""
    import torch
    from neuralnj_amd import synth, weights

    frs, utils_mod, PGPI, PhyInferEnv = ref
    torch.manual_seed(0)
    cfgs = make_cfg(utils_mod, layers)
    cfgs.model.patch_size = patch
    cfgs.model.embed_dim = dim
    cfgs.model.num_enc_heads = heads
    agent = PGPI(cfgs)
    st = weights.seeded_state(cfgs, wseed, style)
    agent.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    agent.eval()
    packed = weights.pack(cfgs, st)

    B, T, L = codes.shape
    onehot = synth.codes_to_onehot(codes)  # int8 [B,T,L,4]
    seqs = [synth.codes_to_seqs(codes[b]) for b in range(B)]
    if keys is None:
        keys = [[f"taxon{i + 1}" for i in range(T)] for _ in range(B)]
    batch = {
        "seqs": seqs,
        "seq_keys": keys,
        "data": torch.from_numpy(onehot),
        "seq_weights": torch.from_numpy((~mask).astype(np.float32)),
    }
    env = PhyInferEnv(cfgs, torch.device("cpu"))

    # record every decode_zxr result and the encoder output by wrapping bound methods
    trace = {"logits": [], "enc": None, "sub": {}}
    orig_decode, orig_encode = agent.decode_zxr, agent.encode_zxr

    def decode_spy(*a, **k):
        r = orig_decode(*a, **k)
        trace["logits"].append(r["logits"].detach().cpu().numpy().copy())
        return r

    def encode_spy(*a, **k):
        r = orig_encode(*a, **k)
        trace["enc"] = r.detach().cpu().numpy().copy()
        return r

    agent.decode_zxr, agent.encode_zxr = decode_spy, encode_spy
    hooks = []
    if capture_layers:
        def mk(tag):
            def hook(_m, _i, o):
                t = o[0] if isinstance(o, tuple) else o
                trace["sub"][tag] = t.detach().cpu().numpy().copy()
            return hook
        hooks.append(agent.embed.register_forward_hook(mk("embed")))
        l0 = agent.seq_emb_layers[0]
        hooks.append(l0.row_self_attention.register_forward_hook(mk("l0_row")))
        hooks.append(l0.column_self_attention.register_forward_hook(mk("l0_col")))
        hooks.append(l0.feed_forward_layer.register_forward_hook(mk("l0_ffn")))

    # record the merges by wrapping env.step
    merges = []
    orig_step = env.step

    def step_spy(actions, *a, **k):
        n = env.states[0].num_trees
        merges.append([env.tree_pairs_dict[n][int(x)] for x in actions])
        return orig_step(actions, *a, **k)

    env.step = step_spy
    frs.device = torch.device("cpu")
    with torch.no_grad():
        _sel, _lps, scores, best_tree = frs.reinforce_rollout(
            batch, agent, env, cfgs, eval=True, argmax=True, branch_optimize=False)
    for h in hooks:
        h.remove()

    enc = trace["enc"]  # [B,T,C,D]
    merges = np.array(merges, dtype=np.int32).transpose(1, 0, 2)  # [B,T-1,2]
    # per-step logits: ragged -> one flat vector per batch element + offsets
    offs = np.cumsum([0] + [l.shape[1] for l in trace["logits"]]).astype(np.int64)
    logits = np.concatenate(trace["logits"], axis=1).astype(np.float32)  # [B, sum P(n)]
    gaps = []
    for l in trace["logits"]:
        if l.shape[1] >= 2:
            s = np.sort(l, axis=1)
            gaps.append(s[:, -1] - s[:, -2])
        else:
            gaps.append(np.zeros(l.shape[0], dtype=l.dtype))
    gaps = np.stack(gaps, 1).astype(np.float32)

    newick = [st_.subtrees[0].utree_op_str for st_ in env.states]
    topo = [st_.subtrees[0].topo_repr for st_ in env.states]

    out = dict(
        codes=codes, mask=mask, wseed=np.int64(wseed), style=np.array(style),
        layers=np.int64(layers), patch=np.int64(patch), dim=np.int64(dim), heads=np.int64(heads), weights_sha256=np.array(weights.digest(packed)),
        merges=merges, logits=logits, logits_offsets=offs, top2_gap=gaps,
        newick=np.array(newick), topo=np.array(topo), best_tree=np.array(best_tree),
        keys=np.array(keys),
        enc_checksum=np.float64(enc.astype(np.float64).sum()),
        enc_abs_checksum=np.float64(np.abs(enc.astype(np.float64)).sum()),
    )
    if enc.size <= 2 * 8 * 128 * 64:
        out["enc"] = enc.astype(np.float32)
    else:
        # strided slice: every 7th row, every 61st column, all features
        out["enc_rows"] = np.arange(0, T, 7, dtype=np.int64)
        out["enc_cols"] = np.arange(0, enc.shape[2], 61, dtype=np.int64)
        out["enc_slice"] = enc[:, ::7, ::61, :].astype(np.float32)
    for k, v in trace["sub"].items():
        out["sub_" + k] = v.astype(np.float32)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: B={B} T={T} L={L} min top2 gap={gaps[:, :-1].min() if gaps.shape[1] > 1 else 0:.3e} "
          f"max|logit|={np.abs(logits).max():.3e} -> {os.path.relpath(path, REPO)}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()

    _install_placeholders()
    import torch
    torch.set_num_threads(8)
    import finetune_rl_search as frs  # noqa: E402  (reference module)
    import utils as utils_mod  # noqa: E402  (reference module)
    from environment import PhyInferEnv  # noqa: E402
    from model import PhyloATTN as PGPI  # noqa: E402
    from neuralnj_amd import synth

    ref = (frs, utils_mod, PGPI, PhyInferEnv)
    cases = []

    def add(name, fn):
        if args.only is None or args.only == name:
            cases.append((name, fn))

    for seed in (0, 1, 2):
        for (B, T, L) in ((1, 8, 128), (2, 8, 128), (1, 20, 256), (1, 50, 1024)):
            nm = f"synth_b{B}_t{T}_l{L}_s{seed}"
            def fn(nm=nm, B=B, T=T, L=L, seed=seed):
                if seed == 2:   # i.i.d. columns: no phylogenetic signal, decisions are near-ties
                    codes = synth.synth_codes(B, T, L, 1000 + seed, gap_frac=0.2)
                else:           # evolved down a random tree, 10 % gaps
                    codes = synth.synth_codes_tree(B, T, L, 1000 + seed)
                mask = np.zeros((B, L), dtype=bool)
                run_case(nm, codes, mask, seed, "sharp", ref, capture_layers=(T == 8 and B == 1))
            add(nm, fn)

    # padded-site case: last 24 columns are '*' padding with mask=True
    def padded():
        codes = synth.synth_codes_tree(2, 8, 128, 77)
        codes[:, :, 104:] = 5
        mask = np.zeros((2, 128), dtype=bool)
        mask[:, 104:] = True
        run_case("padded_b2_t8_l128_s3", codes, mask, 3, "sharp", ref, capture_layers=False)
    add("padded_b2_t8_l128_s3", padded)

    # plain-style weights (reference-like init scale), small
    def plain():
        codes = synth.synth_codes_tree(1, 12, 96, 5)
        run_case("plain_b1_t12_l96_s4", codes, np.zeros((1, 96), dtype=bool), 4, "plain", ref)
    add("plain_b1_t12_l96_s4", plain)

    # 3 taxa (single decision, context of one row) and 2-layer model
    def tiny():
        codes = synth.synth_codes_tree(2, 3, 64, 9)
        run_case("tiny_b2_t3_l64_s5", codes, np.zeros((2, 64), dtype=bool), 5, "sharp", ref, layers=2)
    add("tiny_b2_t3_l64_s5", tiny)

    # the two example MSAs shipped with the reference, through its own loader
    ex_dir = os.path.join(REF, "examples", "len1024taxa50")
    for fname in sorted(os.listdir(ex_dir)):
        if not fname.endswith(".phy"):
            continue
        nm = "example_" + fname[:-4].replace(".", "p")
        def fn(nm=nm, fname=fname):
            from phydata import load_pi_instance  # reference loader
            batch = load_pi_instance(os.path.join(ex_dir, fname))
            onehot = batch["data"].numpy()
            codes = synth.onehot_to_codes(onehot)
            mask = (batch["seq_weights"].numpy() == 0)
            run_case(nm, codes, mask, 0, "sharp", ref, keys=batch["seq_keys"])
        add(nm, fn)

    # ---- round 2: BASELINE-size cases the round-1 set lacked
    # reference-scale ("plain") weights at 50 x 1024, on tree-evolved and on i.i.d. (bench-style) data
    def plain50_tree():
        codes = synth.synth_codes_tree(1, 50, 1024, 2006)
        run_case("plain_b1_t50_l1024_s6", codes, np.zeros((1, 1024), dtype=bool), 6, "plain", ref)
    add("plain_b1_t50_l1024_s6", plain50_tree)

    def plain50_iid():
        codes = synth.synth_codes(1, 50, 1024, 2007, gap_frac=0.2)
        run_case("plainiid_b1_t50_l1024_s7", codes, np.zeros((1, 1024), dtype=bool), 7, "plain", ref)
    add("plainiid_b1_t50_l1024_s7", plain50_iid)

    # the bench's own data law (synth_codes, gap 0.2) under the bench's weights (seed 0, sharp)
    def bench_like():
        codes = synth.synth_codes(1, 50, 1024, 1000, gap_frac=0.2)
        run_case("benchlike_b1_t50_l1024_s0", codes, np.zeros((1, 1024), dtype=bool), 0, "sharp", ref)
    add("benchlike_b1_t50_l1024_s0", bench_like)

    # site counts that are not multiples of 4 / 16 / 32 (real alignments: 1,023 or 1,501 sites)
    def ragged_small():
        codes = synth.synth_codes_tree(2, 9, 30, 2008)
        run_case("ragged_b2_t9_l30_s8", codes, np.zeros((2, 30), dtype=bool), 8, "sharp", ref)
    add("ragged_b2_t9_l30_s8", ragged_small)

    def ragged_mid():
        codes = synth.synth_codes_tree(1, 20, 251, 2009)
        codes[:, :, 240:] = 5
        mask = np.zeros((1, 251), dtype=bool)
        mask[:, 240:] = True
        run_case("ragged_b1_t20_l251_s9", codes, mask, 9, "sharp", ref)
    add("ragged_b1_t20_l251_s9", ragged_mid)

    def ragged_big():
        codes = synth.synth_codes_tree(1, 50, 1023, 2010)
        run_case("ragged_b1_t50_l1023_s10", codes, np.zeros((1, 1023), dtype=bool), 10, "sharp", ref)
    add("ragged_b1_t50_l1023_s10", ragged_big)

    # more than 64 taxa: the reference's bundled evaluation set has 100-taxon alignments
    def wide_synth():
        codes = synth.synth_codes_tree(1, 100, 256, 2011)
        run_case("synth_b1_t100_l256_s11", codes, np.zeros((1, 256), dtype=bool), 11, "sharp", ref)
    add("synth_b1_t100_l256_s11", wide_synth)

    # round 3: 200 taxa (the kernels above 128 rows were pinned through the oracle only), and two more 100-taxon
    # alignments under the stress weights (how far do fp32-level evaluations of this shape scatter?)
    def wide_200():
        codes = synth.synth_codes_tree(1, 200, 256, 2013)
        run_case("synth_b1_t200_l256_s13", codes, np.zeros((1, 256), dtype=bool), 13, "sharp", ref)
    add("synth_b1_t200_l256_s13", wide_200)

    for sd in (14, 15):
        def wide_100(sd=sd):
            codes = synth.synth_codes_tree(1, 100, 256, 2000 + sd)
            run_case(f"synth_b1_t100_l256_s{sd}", codes, np.zeros((1, 256), dtype=bool), sd, "sharp", ref)
        add(f"synth_b1_t100_l256_s{sd}", wide_100)

    # patch_size > 1 (a token = several consecutive sites; the reference's utils.py default is 4): a small case with
    # padded sites, and 20 taxa x 256 sites
    def patch4_small():
        codes = synth.synth_codes_tree(2, 8, 128, 2016)
        codes[:, :, 112:] = 5
        mask = np.zeros((2, 128), dtype=bool)
        mask[:, 112:] = True
        run_case("patch4_b2_t8_l128_s16", codes, mask, 16, "sharp", ref, patch=4, capture_layers=False)
    add("patch4_b2_t8_l128_s16", patch4_small)

    def patch4_mid():
        codes = synth.synth_codes_tree(1, 20, 256, 2017)
        run_case("patch4_b1_t20_l256_s17", codes, np.zeros((1, 256), dtype=bool), 17, "sharp", ref, patch=4)
    add("patch4_b1_t20_l256_s17", patch4_mid)

    def patch2_mid():
        codes = synth.synth_codes(1, 12, 96, 2018, gap_frac=0.2)
        run_case("patch2_b1_t12_l96_s18", codes, np.zeros((1, 96), dtype=bool), 18, "plain", ref, patch=2)
    add("patch2_b1_t12_l96_s18", patch2_mid)

    # embed_dim < 64: the reference's own default shape (utils.py:45-52: patch 4, 32 features, 4 heads, 3 layers) and the
    # same width at patch 1 with a padded tail; the HIP kernels run such a model zero-padded to 64 features
    def dim32_default():
        codes = synth.synth_codes_tree(2, 10, 256, 1019)
        mask = np.zeros((2, 256), dtype=bool)
        run_case("dim32_b2_t10_l256_s19", codes, mask, 19, "sharp", ref, layers=3, patch=4, dim=32, heads=4)
    add("dim32_b2_t10_l256_s19", dim32_default)

    def dim32_patch1():
        codes = synth.synth_codes_tree(1, 20, 160, 1020)
        codes[:, :, 148:] = 5
        mask = np.zeros((1, 160), dtype=bool)
        mask[:, 148:] = True
        run_case("dim32_b1_t20_l160_s20", codes, mask, 20, "plain", ref, layers=3, patch=1, dim=32, heads=4)
    add("dim32_b1_t20_l160_s20", dim32_patch1)

    def dim32_t50():
        # the reference's default model at the size of its bundled examples (50 taxa x 1024 sites = 256 tokens of 4 sites)
        codes = synth.synth_codes_tree(1, 50, 1024, 1022)
        run_case("dim32_b1_t50_l1024_s22", codes, np.zeros((1, 1024), dtype=bool), 22, "plain", ref, layers=3, patch=4,
                 dim=32, heads=4)
    add("dim32_b1_t50_l1024_s22", dim32_t50)

    def dim16_small():
        codes = synth.synth_codes_tree(1, 9, 96, 1021)
        run_case("dim16_b1_t9_l96_s21", codes, np.zeros((1, 96), dtype=bool), 21, "sharp", ref, layers=2, patch=2, dim=16, heads=2)
    add("dim16_b1_t9_l96_s21", dim16_small)

    def wide_70():
        codes = synth.synth_codes_tree(2, 70, 64, 2012)
        run_case("synth_b2_t70_l64_s12", codes, np.zeros((2, 64), dtype=bool), 12, "plain", ref)
    add("synth_b2_t70_l64_s12", wide_70)

    for length, fname in ((256, "G_l_256_n_100_0_0.01_101.phy"), (1024, None)):
        d = os.path.join(REF, "data_gen", "data", "test", f"len{length}", "taxa100")
        if fname is None:
            fname = sorted(f for f in os.listdir(d) if f.endswith(".phy"))[0]
        nm = "data_" + fname[:-4].replace(".", "p")
        def fn(nm=nm, d=d, fname=fname):
            from phydata import load_pi_instance  # reference loader
            batch = load_pi_instance(os.path.join(d, fname))
            onehot = batch["data"].numpy()
            codes = synth.onehot_to_codes(onehot)
            mask = (batch["seq_weights"].numpy() == 0)
            run_case(nm, codes, mask, 0, "sharp", ref, keys=batch["seq_keys"])
        add(nm, fn)

    for nm, fn in cases:
        fn()


if __name__ == "__main__":
    main()
