"""Differentiable operators of the Finetune mode over libnnj_train_hip.so.

One torch.autograd.Function per operator of the reference's forward pass (model.py, msa_modules.py,
axial_attention.py).  torch keeps the graph and owns the device memory; every arithmetic step -- forward and
backward -- is a hand-written gfx950 kernel behind include/nnj_train.h, called on raw device pointers.  There is no
torch arithmetic in here (views / reshapes / empty allocations only) and no CPU fallback: without the library or a
GPU these raise.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NNJ_TRAIN_LIB_PATH") or os.path.join(_HERE, "libnnj_train_hip.so")
_vp = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32


class Gemm(C.Structure):
    _fields_ = [("A", _vp), ("B", _vp), ("C", _vp),
                ("M", _i32), ("N", _i32), ("K", _i32), ("nb1", _i32), ("nb2", _i32),
                ("sAm", _i64), ("sAk", _i64), ("sAb1", _i64), ("sAb2", _i64),
                ("sBk", _i64), ("sBn", _i64), ("sBb1", _i64), ("sBb2", _i64),
                ("sCm", _i64), ("sCn", _i64), ("sCb1", _i64), ("sCb2", _i64),
                ("alpha", C.c_float), ("beta", C.c_float), ("bias", _vp)]


_SIGS = {
    "nnjt_abi_version": ([], C.c_int),
    "nnjt_last_error": ([], C.c_char_p),
    "nnjt_gemm_run": ([C.POINTER(Gemm), _vp], C.c_int),
    "nnjt_skinny_gemm": ([_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i64, C.c_float, _vp], C.c_int),
    "nnjt_wgrad64": ([_vp, _vp, _vp, _i64, _i64, _vp], C.c_int),
    "nnjt_add_bias": ([_vp, _vp, _i64, _i32, _vp], C.c_int),
    "nnjt_colsum": ([_vp, _vp, _i64, _i32, _vp], C.c_int),
    "nnjt_sum_rows": ([_vp, _vp, _i64, _i64, _vp], C.c_int),
    "nnjt_layernorm_fwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp], C.c_int),
    "nnjt_layernorm_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp], C.c_int),
    "nnjt_gelu_fwd": ([_vp, _vp, _i64, _vp], C.c_int),
    "nnjt_gelu_bwd": ([_vp, _vp, _vp, _i64, _vp], C.c_int),
    "nnjt_gate_fwd": ([_vp, _vp, _vp, _vp, _i64, _vp], C.c_int),
    "nnjt_gate_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp], C.c_int),
    "nnjt_softmax_fwd": ([_vp, _vp, _vp, _i64, _i32, _vp], C.c_int),
    "nnjt_softmax_bwd": ([_vp, _vp, _vp, _i64, _i32, _vp], C.c_int),
    "nnjt_axpby": ([C.c_float, _vp, C.c_float, _vp, _vp, _i64, _vp], C.c_int),
    "nnjt_rowscale": ([_vp, _vp, _vp, _i64, _i32, _vp], C.c_int),
    "nnjt_dropout_fwd": ([_vp, _vp, _vp, _i64, C.c_float, C.c_uint64, C.c_uint64, _vp], C.c_int),
    "nnjt_dropout_bwd": ([_vp, _vp, _vp, _i64, C.c_float, _vp], C.c_int),
    "nnjt_fill_where": ([_vp, _vp, C.c_float, _i64, _i32, _i32, _i32, _vp], C.c_int),
    "nnjt_gather_rows": ([_vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp], C.c_int),
    "nnjt_scatter_rows_add": ([_vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp], C.c_int),
    "nnjt_permute5": ([_vp, _vp, C.POINTER(_i64), C.POINTER(_i64), _i32, _vp], C.c_int),
}
_lib = None
ABI_VERSION = 2                                          # include/nnj_train.h as of this file (nnjt_abi_version)


def exported_symbols():
    return sorted(_SIGS)


def load_library(path: str = LIB_PATH):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m neuralnj_amd.build` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (argt, rest) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = argt
        fn.restype = rest
    if lib.nnjt_abi_version() != ABI_VERSION:           # e.g. nnjt_gemm grew a field: a stale build must not be called
        raise RuntimeError(f"{path} has ABI version {lib.nnjt_abi_version()}, this package binds version {ABI_VERSION}: "
                           "rebuild it with `python -m neuralnj_amd.build`")
    _lib = lib
    return lib


def _chk(rc):
    if rc != 0:
        raise RuntimeError(f"libnnj_train_hip error {rc}: {load_library().nnjt_last_error().decode()}")


def _p(t):
    return None if t is None else _vp(t.data_ptr())


def _st(t):
    # the raw handle of torch's current stream (torch.cuda.current_stream builds a Stream object per call: 11 ms of an
    # episode's 7000 operator calls)
    idx = t.device.index
    cur = torch.cuda.current_device()
    if idx is not None and idx != cur:
        # the nnjt_* kernels launch on the current device: a tensor elsewhere would be computed on the wrong GPU, on a
        # stream of another device.  rollout.reinforce_loss selects the device for the whole episode.
        raise RuntimeError(f"libnnj_train_hip: tensor on cuda:{idx} but the current device is cuda:{cur}; "
                           f"wrap the call in torch.cuda.device({idx})")
    return _vp(torch._C._cuda_getCurrentRawStream(cur if idx is None else idx))


def _need(t):
    if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("the Finetune operators take contiguous fp32 tensors on a HIP device (no CPU path)")
    return t


# ---------------------------------------------------------------------------------------------------- GEMM plumbing
def gemm(A, B, Cout, M, N, K, sA, sB, sC, nb=(1, 1), bA=(0, 0), bB=(0, 0), bC=(0, 0), alpha=1.0, beta=0.0, bias=None):
    """Cout[b1,b2][m,n] = alpha * sum_k A[..][m,k] B[..][k,n] + bias[n] + beta * Cout; sX = (row stride, column stride)."""
    g = Gemm(_p(A), _p(B), _p(Cout), M, N, K, nb[0], nb[1], sA[0], sA[1], bA[0], bA[1], sB[0], sB[1], bB[0], bB[1],
             sC[0], sC[1], bC[0], bC[1], alpha, beta, _p(bias))
    _chk(load_library().nnjt_gemm_run(C.byref(g), _st(Cout)))


def _piece(M, N, K, nb=1):
    """length of the pieces a contraction of length K is cut into: enough pieces that the launch has about a thousand
    workgroups (one 128 x 64 output tile per piece and batch entry), none shorter than 64; 0 = do not cut"""
    tiles = ((M + 127) // 128) * ((N + 63) // 64) * nb
    want = max(1, 1024 // tiles)                             # four workgroups per CU: each is a chain of load -> LDS -> barrier
    if want == 1 or K < 128:
        return 0
    piece = max(64, -(-K // want))
    piece = -(-piece // 16) * 16
    return piece if piece < K else 0


def gemm_longk(A, B, Cout, M, N, K, sA, sB, sC, alpha=1.0):
    """Cout[m,n] = alpha * sum_k A[m,k] B[k,n] for a contraction that is long next to its output (weight gradients: k =
    tokens; the pair scorer's attention logits: k = sites x features): a single workgroup per output tile would walk k
    alone, so k is cut into pieces computed as a batch into a scratch, which nnjt_sum_rows adds up in order."""
    piece = _piece(M, N, K)
    if not piece:
        return gemm(A, B, Cout, M, N, K, sA, sB, sC, alpha=alpha)
    nfull = K // piece
    tail = K - nfull * piece
    parts = torch.empty((nfull + (1 if tail else 0), M, N), dtype=torch.float32, device=Cout.device)
    gemm(A, B, parts, M, N, piece, sA, sB, (N, 1), nb=(nfull, 1), bA=(piece * sA[1], 0), bB=(piece * sB[0], 0),
         bC=(M * N, 0), alpha=alpha)
    if tail:
        At = A.view(-1)[nfull * piece * sA[1]:]
        Bt = B.view(-1)[nfull * piece * sB[0]:]
        gemm(At, Bt, parts[nfull], M, N, tail, sA, sB, (N, 1), alpha=alpha)
    assert sC == (N, 1) and Cout.is_contiguous(), "cut contractions write dense outputs"
    _chk(load_library().nnjt_sum_rows(_p(parts), _p(Cout), parts.shape[0], M * N, _st(Cout)))


# ---------------------------------------------------------------------------------------------------- operators
class Linear(torch.autograd.Function):
    """y = x W^T + b over the last dim (nn.Linear; reference msa_modules.py / model.py / axial_attention.py)."""

    @staticmethod
    def forward(ctx, x, W, b):
        _need(x), _need(W)
        K = x.shape[-1]
        Mo = W.shape[0]
        rows = x.numel() // K
        y = torch.empty(x.shape[:-1] + (Mo,), dtype=torch.float32, device=x.device)
        if b is not None:
            _need(b)
        gemm(x, W, y, rows, Mo, K, (K, 1), (1, K), (Mo, 1), bias=b)            # bias added in the GEMM's epilogue
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _need(dy.contiguous())
        K = x.shape[-1]
        Mo = W.shape[0]
        rows = x.numel() // K
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(dy, W, dx, rows, K, Mo, (Mo, 1), (K, 1), (K, 1))
        if ctx.needs_input_grad[1]:
            dW = torch.empty_like(W)
            if Mo == 64 and K == 64 and rows >= 4096:
                # the 64 -> 64 layers over many tokens (most of the model): operands straight from global memory into
                # the MFMA lanes, ~512 workgroups, parts added in order
                per = max(256, -(-rows // 512))
                per = -(-per // 8) * 8
                parts = torch.empty((-(-rows // per), 4096), dtype=torch.float32, device=x.device)
                lib = load_library()
                _chk(lib.nnjt_wgrad64(_p(dy), _p(x), _p(parts), rows, per, _st(x)))
                _chk(lib.nnjt_sum_rows(_p(parts), _p(dW), parts.shape[0], 4096, _st(x)))
            else:
                gemm_longk(dy, x, dW, Mo, K, rows, (1, Mo), (K, 1), (K, 1))      # dW = dy^T x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.zeros(Mo, dtype=torch.float32, device=x.device)
            _chk(load_library().nnjt_colsum(_p(dy), _p(db), rows, Mo, _st(dy)))
        return dx, dW, db


class LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        _need(x)
        cols = x.shape[-1]
        rows = x.numel() // cols
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        _chk(load_library().nnjt_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, cols, _st(x)))
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = _need(dy.contiguous())
        cols = x.shape[-1]
        rows = x.numel() // cols
        dx = torch.empty_like(x)
        dg = torch.zeros_like(gamma)
        db = torch.zeros_like(gamma)
        _chk(load_library().nnjt_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), rows,
                                               cols, _st(x)))
        return dx, dg, db


class Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need(x)
        y = torch.empty_like(x)
        _chk(load_library().nnjt_gelu_fwd(_p(x), _p(y), x.numel(), _st(x)))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _need(dy.contiguous())
        dx = torch.empty_like(x)
        _chk(load_library().nnjt_gelu_bwd(_p(dy), _p(x), _p(dx), x.numel(), _st(x)))
        return dx


class _DropoutState:
    """Counter of the dropout generator: the seed follows torch.initial_seed() (torch.manual_seed sets it), the offset
    restarts when the seed changes and advances by the element count of every call.  `tape`, when a list, receives the
    keep mask of every call in order; `replay`, when a list, supplies them instead of the generator (uint8 device
    tensors, consumed front to back) -- a recorded run, or the masks of a reference run, reproduced exactly."""
    seed = None
    offset = 0
    tape = None
    replay = None

    @classmethod
    def take(cls, n):
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        if seed != cls.seed:
            cls.seed, cls.offset = seed, 0
        off = cls.offset
        cls.offset += n
        return seed, off


class Dropout(torch.autograd.Function):
    """nn.Dropout(p) in training mode (reference msa_modules.py:119,149; axial_attention.py:56,136,233)."""

    @staticmethod
    def forward(ctx, x, p):
        _need(x)
        y = torch.empty_like(x)
        if _DropoutState.replay is not None:
            keep = _DropoutState.replay.pop(0)
            if keep.dtype != torch.uint8 or keep.numel() != x.numel() or keep.device != x.device or not keep.is_contiguous():
                raise RuntimeError("dropout replay: the next recorded mask does not fit this call")
            _chk(load_library().nnjt_dropout_bwd(_p(x), _p(keep), _p(y), x.numel(), float(p), _st(x)))   # x * keep / (1 - p)
        else:
            keep = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
            seed, off = _DropoutState.take(x.numel())
            _chk(load_library().nnjt_dropout_fwd(_p(x), _p(y), _p(keep), x.numel(), float(p), seed, off, _st(x)))
        if _DropoutState.tape is not None:
            _DropoutState.tape.append(keep)
        ctx.save_for_backward(keep)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        dy = _need(dy.contiguous())
        dx = torch.empty_like(dy)
        _chk(load_library().nnjt_dropout_bwd(_p(dy), _p(keep), _p(dx), dy.numel(), ctx.p, _st(dy)))
        return dx, None


def dropout(x, p):
    return Dropout.apply(x, p) if p > 0.0 else x


class Gate(torch.autograd.Function):
    """out = sigmoid(h) * a + (1 - sigmoid(h)) * b   (reference model.py:105-108, 150-153)."""

    @staticmethod
    def forward(ctx, h, a, b):
        _need(h), _need(a), _need(b)
        out = torch.empty_like(h)
        _chk(load_library().nnjt_gate_fwd(_p(h), _p(a), _p(b), _p(out), h.numel(), _st(h)))
        ctx.save_for_backward(h, a, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, a, b = ctx.saved_tensors
        dout = _need(dout.contiguous())
        dh, da, db = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        _chk(load_library().nnjt_gate_bwd(_p(dout), _p(h), _p(a), _p(b), _p(dh), _p(da), _p(db), h.numel(), _st(h)))
        return dh, da, db


class Softmax(torch.autograd.Function):
    """softmax over the last dim; keep (uint8, same shape, or None): entries with keep == 0 get probability 0."""

    @staticmethod
    def forward(ctx, x, keep):
        _need(x)
        cols = x.shape[-1]
        rows = x.numel() // cols
        y = torch.empty_like(x)
        _chk(load_library().nnjt_softmax_fwd(_p(x), _p(keep), _p(y), rows, cols, _st(x)))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _need(dy.contiguous())
        cols = y.shape[-1]
        rows = y.numel() // cols
        dx = torch.empty_like(y)
        _chk(load_library().nnjt_softmax_bwd(_p(dy), _p(y), _p(dx), rows, cols, _st(y)))
        return dx, None


class Axpby(torch.autograd.Function):
    """a * x + b * y (y may be None)."""

    @staticmethod
    def forward(ctx, x, y, a, b):
        _need(x)
        if y is not None:
            _need(y)
        out = torch.empty_like(x)
        _chk(load_library().nnjt_axpby(a, _p(x), b, _p(y), _p(out), x.numel(), _st(x)))
        ctx.a, ctx.b, ctx.has_y = a, b, y is not None
        return out

    @staticmethod
    def backward(ctx, d):
        d = _need(d.contiguous())
        dx = dy = None
        if ctx.needs_input_grad[0]:
            dx = d if ctx.a == 1.0 else _scaled(d, ctx.a)
        if ctx.has_y and ctx.needs_input_grad[1]:
            dy = d if ctx.b == 1.0 else _scaled(d, ctx.b)
        return dx, dy, None, None


def _scaled(t, a):
    out = torch.empty_like(t)
    _chk(load_library().nnjt_axpby(a, _p(t), 0.0, None, _p(out), t.numel(), _st(t)))
    return out


def add(x, y):
    return Axpby.apply(x, y, 1.0, 1.0)


def sub(x, y):
    return Axpby.apply(x, y, 1.0, -1.0)


class RowScale(torch.autograd.Function):
    """out[r, :] = x[r, :] * s[r]; s carries no gradient (scalings and 0/1 masks)."""

    @staticmethod
    def forward(ctx, x, s):
        _need(x), _need(s)
        cols = x.shape[-1]
        rows = x.numel() // cols
        assert s.numel() == rows
        out = torch.empty_like(x)
        _chk(load_library().nnjt_rowscale(_p(x), _p(s), _p(out), rows, cols, _st(x)))
        ctx.save_for_backward(s)
        return out

    @staticmethod
    def backward(ctx, d):
        (s,) = ctx.saved_tensors
        d = _need(d.contiguous())
        cols = d.shape[-1]
        out = torch.empty_like(d)
        _chk(load_library().nnjt_rowscale(_p(d), _p(s), _p(out), d.numel() // cols, cols, _st(d)))
        return out, None


class FillWhere(torch.autograd.Function):
    """masked_fill on a tensor viewed as [outer (sel_rows), inner, cols]: x[o, i, c] = value where sel[o, c]."""

    @staticmethod
    def forward(ctx, x, sel, value, inner, sel_rows):
        _need(x)
        cols = x.shape[-1]
        rows = x.numel() // cols
        out = x.clone()
        _chk(load_library().nnjt_fill_where(_p(out), _p(sel), value, rows, inner, sel_rows, cols, _st(x)))
        ctx.save_for_backward(sel)
        ctx.geom = (rows, inner, sel_rows, cols)
        return out

    @staticmethod
    def backward(ctx, d):
        (sel,) = ctx.saved_tensors
        rows, inner, sel_rows, cols = ctx.geom
        out = _need(d.contiguous()).clone()
        _chk(load_library().nnjt_fill_where(_p(out), _p(sel), 0.0, rows, inner, sel_rows, cols, _st(out)))
        return out, None, None, None, None


class GatherRows(torch.autograd.Function):
    """out[b, p] = src[b, idx[b, p]] over rows of trailing size `width` (src [B, n, ...], idx int64 [B, p])."""

    @staticmethod
    def forward(ctx, src, idx):
        _need(src)
        B, n = src.shape[:2]
        width = src.numel() // (B * n)
        idx = idx.to(torch.int64).contiguous()
        p = idx.shape[1]
        out = torch.empty((B, p) + tuple(src.shape[2:]), dtype=torch.float32, device=src.device)
        _chk(load_library().nnjt_gather_rows(_p(src), _p(idx), _p(out), B, n, p, width, _st(src)))
        ctx.save_for_backward(idx)
        ctx.shape = tuple(src.shape)
        return out

    @staticmethod
    def backward(ctx, d):
        (idx,) = ctx.saved_tensors
        d = _need(d.contiguous())
        B, n = ctx.shape[:2]
        width = d.numel() // (B * idx.shape[1])
        ds = torch.zeros(ctx.shape, dtype=torch.float32, device=d.device)
        _chk(load_library().nnjt_scatter_rows_add(_p(d), _p(idx), _p(ds), B, n, idx.shape[1], width, _st(d)))
        return ds, None


class Permute(torch.autograd.Function):
    """x.permute(order).contiguous() for tensors of up to five dims (einops.rearrange of the reference)."""

    @staticmethod
    def forward(ctx, x, order):
        _need(x)
        nd = x.dim()
        assert nd <= 5 and len(order) == nd
        pad = 5 - nd
        shape = (1,) * pad + tuple(x.shape)
        strides = (0,) * pad + tuple(x.stride())
        order5 = tuple(range(pad)) + tuple(o + pad for o in order)
        dims = (_i64 * 5)(*[shape[o] for o in order5])
        strd = (_i64 * 5)(*[strides[o] for o in order5])
        out = torch.empty([x.shape[o] for o in order], dtype=torch.float32, device=x.device)
        _chk(load_library().nnjt_permute5(_p(x), _p(out), dims, strd, 0, _st(x)))
        ctx.geom = (tuple(x.shape), [shape[o] for o in order5], [strides[o] for o in order5])
        return out

    @staticmethod
    def backward(ctx, d):
        xshape, dims, strd = ctx.geom
        d = _need(d.contiguous())
        dx = torch.empty(xshape, dtype=torch.float32, device=d.device)
        _chk(load_library().nnjt_permute5(_p(dx), _p(d), (_i64 * 5)(*dims), (_i64 * 5)(*strd), 1, _st(d)))
        return dx, None


class Bmm(torch.autograd.Function):
    """C[nb] = alpha * A[nb] @ B[nb] (B given as [nb,K,N], or as [nb,N,K] with trans_b): the attention einsums
    (reference axial_attention.py:97,114,216,234; model.py:118,148)."""

    @staticmethod
    def forward(ctx, A, B, trans_b, alpha):
        _need(A), _need(B)
        nb, M, K = A.shape
        N = B.shape[1] if trans_b else B.shape[2]
        Cout = torch.empty((nb, M, N), dtype=torch.float32, device=A.device)
        sB = (1, K) if trans_b else (N, 1)
        _bgemm(A, B, Cout, nb, M, N, K, (K, 1), sB, (N, 1), M * K, B.shape[1] * B.shape[2], M * N, alpha)
        ctx.save_for_backward(A, B)
        ctx.trans_b, ctx.alpha = trans_b, alpha
        return Cout

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dC = _need(dC.contiguous())
        nb, M, K = A.shape
        tb, alpha = ctx.trans_b, ctx.alpha
        N = B.shape[1] if tb else B.shape[2]
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dA = torch.empty_like(A)                 # dA = dC @ B^T  (tb: dC @ B)
            sB = (K, 1) if tb else (1, N)
            _bgemm(dC, B, dA, nb, M, K, N, (N, 1), sB, (K, 1), M * N, B.shape[1] * B.shape[2], M * K, alpha)
        if ctx.needs_input_grad[1]:
            dB = torch.empty_like(B)
            if tb:                                   # dB[n,k] = sum_m dC[m,n] A[m,k]
                _bgemm(dC, A, dB, nb, N, K, M, (1, N), (K, 1), (K, 1), M * N, M * K, N * K, alpha)
            else:                                    # dB[k,n] = sum_m A[m,k] dC[m,n]
                _bgemm(A, dC, dB, nb, K, N, M, (1, K), (N, 1), (N, 1), M * K, M * N, K * N, alpha)
        return dA, dB, None, None


def _bgemm(A, B, Cout, nb, M, N, K, sA, sB, sC, bsA, bsB, bsC, alpha):
    """batched product with a long-contraction path and the 65535-entry grid limit handled here"""
    if M <= 64 and K <= 64 and N >= 4096 and N % 64 == 0 and nb <= 65535 and sB == (N, 1) and sC == (N, 1) \
            and bsB == K * N and bsC == M * N:
        # few rows times sites x features (x_g of the pair scorer, and the state's gradient through it)
        _chk(load_library().nnjt_skinny_gemm(_p(A), sA[0], sA[1], bsA, _p(B), bsB, _p(Cout), bsC, nb, M, K, N, alpha,
                                             _st(Cout)))
        return
    piece = _piece(M, N, K, nb) if nb <= 64 else 0
    if piece and sC == (N, 1) and bsC == M * N and Cout.is_contiguous():
        # long contractions of a small batch (the pair scorer's logits over sites x features, one entry per alignment
        # or replica): ALL entries' pieces in one launch -- parts [piece, entry, M, N] through the kernel's two batch
        # dimensions -- and one nnjt_sum_rows over the pieces
        nfull = K // piece
        tail = K - nfull * piece
        parts = torch.empty((nfull + (1 if tail else 0), nb, M, N), dtype=torch.float32, device=Cout.device)
        gemm(A, B, parts, M, N, piece, sA, sB, (N, 1), nb=(nfull, nb), bA=(piece * sA[1], bsA), bB=(piece * sB[0], bsB),
             bC=(nb * M * N, M * N), alpha=alpha)
        if tail:
            gemm(A.reshape(-1)[nfull * piece * sA[1]:], B.reshape(-1)[nfull * piece * sB[0]:], parts[nfull], M, N, tail,
                 sA, sB, (N, 1), nb=(nb, 1), bA=(bsA, 0), bB=(bsB, 0), bC=(M * N, 0), alpha=alpha)
        _chk(load_library().nnjt_sum_rows(_p(parts), _p(Cout), parts.shape[0], nb * M * N, _st(Cout)))
        return
    if piece:
        for b in range(nb):
            gemm_longk(A[b], B[b], Cout[b], M, N, K, sA, sB, sC, alpha=alpha)
        return
    for b0 in range(0, nb, 32768):
        cnt = min(32768, nb - b0)
        gemm(A[b0:], B[b0:], Cout[b0:], M, N, K, sA, sB, sC, nb=(cnt, 1), bA=(bsA, 0), bB=(bsB, 0), bC=(bsC, 0),
             alpha=alpha)
