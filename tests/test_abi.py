"""The C-ABI library loads without a GPU and exports every symbol include/nnj.h declares
(no compute call here); argument errors are reported through the ABI, not exceptions."""
import ctypes as C
import os
import re

import pytest


def _declared(repo_root):
    src = open(os.path.join(repo_root, "include", "nnj.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nnj_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header(repo_root):
    from neuralnj_amd import _lib, build
    build.build_hip()
    lib = _lib.load_library()
    names = _declared(repo_root)
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nnj.h but not exported"
    assert set(_lib.exported_symbols()) == set(names)
    assert lib.nnj_abi_version() == 1


def test_train_library_exports_its_header(repo_root):
    """libnnj_train_hip.so (the Finetune operators) exports exactly what include/nnj_train.h declares."""
    from neuralnj_amd import build, train_ops
    build.build_train()
    lib = train_ops.load_library()
    src = open(os.path.join(repo_root, "include", "nnj_train.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(nnjt_[a-z0-9_]+)\s*\(", src)))
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nnj_train.h but not exported"
    assert set(train_ops.exported_symbols()) == set(names)
    assert lib.nnjt_abi_version() == train_ops.ABI_VERSION == 2


def test_param_count_and_argument_errors():
    from neuralnj_amd import _lib
    lib = _lib.load_library()
    cfg = _lib.NnjConfig(4, 1, 64, 8, 6, 0)
    n = C.c_size_t()
    assert lib.nnj_num_params(C.byref(cfg), C.byref(n)) == 0 and n.value == 425857
    assert lib.nnj_num_params(None, C.byref(n)) == -1
    h = C.c_void_p()
    ref_default = _lib.NnjConfig(4, 4, 32, 4, 3, 0)         # the reference's default model (utils.py:45-52): covered
    assert lib.nnj_num_params(C.byref(ref_default), C.byref(n)) == 0 and n.value == 57889
    assert lib.nnj_create(C.byref(ref_default), C.byref(h)) in (0, -6)   # created, or NNJ_ERR_NO_DEVICE on a CPU box
    if h.value:
        lib.nnj_destroy(h)
    for bad in (_lib.NnjConfig(4, 1, 64, 4, 6, 0),          # heads of 16 features
                _lib.NnjConfig(4, 1, 128, 16, 6, 0),        # wider than the kernels' 64 features
                _lib.NnjConfig(4, 1, 36, 4, 6, 0),          # not a multiple of 8
                _lib.NnjConfig(5, 1, 64, 8, 6, 0)):         # another alphabet
        assert lib.nnj_create(C.byref(bad), C.byref(h)) == -2   # NNJ_ERR_UNSUPPORTED, stated loudly
        assert b"embed_dim 8..64" in lib.nnj_last_error(None)
    assert lib.nnj_profile_kinds() >= 12
    assert lib.nnj_profile_kind_name(1) == b"k_tok1"


def test_product_never_imports_the_oracle(repo_root):
    """The product package must not reference oracle/ (parity rule)."""
    for root, _, files in os.walk(os.path.join(repo_root, "neuralnj_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle_lib" not in txt and "libnnj_oracle" not in txt and "nnjo_" not in txt, f


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from neuralnj_amd import utils
    from neuralnj_amd._lib import Nnj
    with pytest.raises(RuntimeError):
        Nnj(utils.shipped_config())
    from neuralnj_amd.model import PhyloATTN
    m = PhyloATTN(utils.shipped_config())
    assert len(m.state_dict()) == 172
    with pytest.raises(RuntimeError):
        m.encode_zxr(torch.zeros(1, 4, 8, 4, dtype=torch.int8), torch.zeros(1, 8, dtype=torch.bool))


def test_state_dict_layout_matches_spec():
    """Same keys, order and shapes as the reference's PhyloATTN.state_dict()."""
    from neuralnj_amd import utils, weights
    from neuralnj_amd.model import PhyloATTN
    cfgs = utils.shipped_config()
    sd = PhyloATTN(cfgs).state_dict()
    spec = weights.param_spec(cfgs)
    assert list(sd.keys()) == [n for n, _ in spec]
    assert [tuple(v.shape) for v in sd.values()] == [s for _, s in spec]
    assert list(sd.keys())[0] == "seq_emb_layers.0.row_self_attention.layer.k_proj.weight"
    assert list(sd.keys())[-1] == "s_out.2.bias"


def test_workspace_regions_hold_every_launch_geometry():
    """the first table of sampled replicas of ONE alignment is launched with the geometry of a batch
    of one, whose star blocks can exceed the regions sized for the batch (T > 128, C >= 6000, B not a power of two).
    Host-only check of every scorer launch of a rollout against the regions loop_ws reserves -- no device needed."""
    from neuralnj_amd import _lib
    lib = _lib.load_library()
    assert lib.nnj_workspace_selfcheck(0, 8, 128) == -1
    bad = []
    for T in (2, 3, 17, 50, 64, 65, 100, 128, 129, 200, 256):
        for Cn in (128, 1024, 4096, 6000, 8192):
            for B in (1, 2, 3, 5, 6, 7, 8, 9, 11, 16, 64, 256):
                if T > 64 and B > 16:
                    continue
                rc = lib.nnj_workspace_selfcheck(B, T, Cn)
                if rc != 0:
                    bad.append((B, T, Cn, rc))
    assert not bad, bad
