"""ctypes binding of libnnj_hip.so (include/nnj.h).  PyTorch is used only for device
memory and streams.  There is NO fallback: if the HIP library is missing or no gfx950
device is usable, construction raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import weights as _weights

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NNJ_LIB_PATH") or os.path.join(_HERE, "libnnj_hip.so")   # (override: A/B timing of two builds)

_vp = C.c_void_p


class NnjConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("vocab_size", "patch_size", "embed_dim", "num_heads", "num_layers", "device")]


class NnjSubstModel(C.Structure):
    """include/nnj.h nnj_subst_model (GTR+I+G)."""
    _fields_ = [("rates", C.c_double * 6), ("freqs", C.c_double * 4), ("alpha", C.c_double), ("pinv", C.c_double),
                ("ncat", C.c_int32)]


_SIGS = {
    "nnj_abi_version": ([], C.c_int),
    "nnj_create": ([C.POINTER(NnjConfig), C.POINTER(_vp)], C.c_int),
    "nnj_destroy": ([_vp], C.c_int),
    "nnj_last_error": ([_vp], C.c_char_p),
    "nnj_num_params": ([C.POINTER(NnjConfig), C.POINTER(C.c_size_t)], C.c_int),
    "nnj_load_weights": ([_vp, _vp, C.c_size_t], C.c_int),
    "nnj_workspace_bytes": ([_vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)], C.c_int),
    "nnj_workspace_selfcheck": ([C.c_int32, C.c_int32, C.c_int32], C.c_int),
    "nnj_encode": ([_vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_size_t, _vp], C.c_int),
    "nnj_pair_scores_full": ([_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_size_t, _vp], C.c_int),
    "nnj_pair_scores_incr": ([_vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_size_t, _vp], C.c_int),
    "nnj_score_index_map": ([_vp, _vp, _vp, C.c_int32, C.c_int32, _vp], C.c_int),
    "nnj_aggregate": ([_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_size_t, _vp], C.c_int),
    "nnj_env_step": ([_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_size_t, _vp], C.c_int),
    "nnj_session_reset": ([_vp], C.c_int),
    "nnj_select_pair": ([_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp], C.c_int),
    "nnj_rollout_argmax": ([_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp,
                            C.c_size_t, _vp], C.c_int),
    "nnj_rollout_sample": ([_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_float, _vp, _vp, _vp,
                            C.c_size_t, _vp], C.c_int),
    "nnj_profile_enable": ([_vp, C.c_int32], C.c_int),
    "nnj_profile_kinds": ([], C.c_int),
    "nnj_profile_kind_name": ([C.c_int32], C.c_char_p),
    "nnj_profile_read": ([_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32], C.c_int),
    "nnj_profile_dropped": ([_vp, C.POINTER(C.c_int64)], C.c_int),
    "nnj_set_concurrency": ([_vp, C.c_int32], C.c_int),
    "nnj_topology_hash": ([_vp, _vp, C.c_int32, C.c_int32, _vp, _vp], C.c_int),
    "nnj_lik_workspace_bytes": ([C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)], C.c_int),
    "nnj_tree_loglik": ([_vp, _vp, C.c_int32, _vp, _vp, _vp, C.POINTER(NnjSubstModel), C.c_int32, C.c_int32, C.c_int32, _vp,
                         _vp, C.c_size_t, _vp], C.c_int),
    "nnj_lik_model_probe": ([_vp, C.POINTER(NnjSubstModel), C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                             C.POINTER(C.c_double)], C.c_int),
    "nnj_tree_optimize": ([_vp, _vp, C.c_int32, _vp, _vp, _vp, C.POINTER(NnjSubstModel), C.c_int32, C.c_int32, C.c_int32,
                           C.c_int32, _vp, _vp, _vp, C.c_size_t, _vp], C.c_int),
    "nnj_debug_encoder_stop": ([_vp, C.c_int32], C.c_int),
    "nnj_numeric_status": ([_vp, C.POINTER(C.c_int32), _vp], C.c_int),
    "nnj_step": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_size_t,
                  _vp], C.c_int),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


ABI_VERSION = 1                                          # include/nnj.h as of this file (nnj_abi_version)


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library and bind every symbol of include/nnj.h (no compute call)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m neuralnj_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (argt, rest) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = argt
        fn.restype = rest
    if lib.nnj_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path} has ABI version {lib.nnj_abi_version()}, this package binds version {ABI_VERSION}: "
                           "rebuild it with `python -m neuralnj_amd.build`")
    _lib = lib
    return lib


def _p(t):
    if t is None:
        return None
    return _vp(t.data_ptr())


class Nnj:
    """One device context: handle + weights + a growable torch workspace."""

    def __init__(self, cfgs, device=None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("neuralnj_amd needs a ROCm GPU (gfx950); none is visible and there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("neuralnj_amd runs on HIP devices only")
        if self.device.index is None:         # "cuda" = torch's current device, not device 0
            self.device = torch.device("cuda", torch.cuda.current_device())
        m = cfgs.model
        self.cfg = NnjConfig(int(m.vocab_size), int(m.patch_size), int(m.embed_dim), int(m.num_enc_heads),
                             int(m.num_enc_layers), int(self.device.index))
        self.cfgs = cfgs
        self.D = int(m.embed_dim)
        self.K = int(m.patch_size)            # a token of the state is K consecutive sites (reference model.py:72-76)
        self.h = _vp()
        rc = self.lib.nnj_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"nnj_create failed ({rc}): {self.lib.nnj_last_error(None).decode()}")
        self._ws = None
        self._has_weights = False

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(f"libnnj_hip error {rc}: {self.lib.nnj_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None) and self.h:
            self.lib.nnj_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        self._last_stream = torch.cuda.current_stream(self.device)      # the stream the workspace was last used on
        return _vp(self._last_stream.cuda_stream)

    # The library continues a session when it is handed the very tensor it last saw / returned (include/nnj.h,
    # "Sessions").  The wrapper keeps that tensor alive (its address cannot be reused) and remembers its version
    # counter: a tensor the caller has written to in place, or any other tensor, resets the session first.
    def _session_in(self, state):
        ref = getattr(self, "_sess_tensor", None)
        if ref is None:
            return
        if state is not ref or state._version != self._sess_version:
            if state.data_ptr() == ref.data_ptr():
                self._chk(self.lib.nnj_session_reset(self.h))
            self._sess_tensor = None

    def _session_out(self, dense):
        self._sess_tensor = dense
        self._sess_version = dense._version

    def workspace(self, B, T, L):
        need = C.c_size_t()
        self._chk(self.lib.nnj_workspace_bytes(self.h, B, T, L, C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value:
            if self._ws is not None and getattr(self, "_last_stream", None) is not None:
                # kernels of the last call may still read the old buffer on a stream other than the allocator's
                self._ws.record_stream(self._last_stream)
            self._ws = None
            self._sess_tensor = None                # a session lives in the workspace it was started in
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self._ws

    def load_weights(self, packed):
        """packed: flat fp32 numpy array / tensor in state_dict order (weights.pack)."""
        if hasattr(packed, "detach"):
            packed = packed.detach().cpu().numpy()
        packed = np.ascontiguousarray(packed, dtype=np.float32)
        self._chk(self.lib.nnj_load_weights(self.h, packed.ctypes.data_as(_vp), packed.size))
        self._has_weights = True

    def load_state_dict(self, state_dict):
        self.load_weights(_weights.pack(self.cfgs, state_dict))

    def _u8(self, t):
        if t is None:
            return None
        t = torch.as_tensor(t)
        if t.dtype == torch.bool:
            t = t.to(torch.uint8)
        return t.to(self.device, torch.uint8).contiguous()

    def _i32(self, t):
        return torch.as_tensor(t).to(self.device, torch.int32).contiguous()

    def _f32(self, t):
        return torch.as_tensor(t).to(self.device, torch.float32).contiguous()

    # ------------------------------------------------------------------ entry points
    def encode(self, codes, mask=None):
        codes = self._u8(codes)
        B, T, L = codes.shape
        mask = self._u8(mask)
        out = torch.empty((B, T, L // self.K, self.D), dtype=torch.float32, device=self.device)
        ws = self.workspace(B, T, L)
        self._chk(self.lib.nnj_encode(self.h, _p(codes), None, _p(mask), _p(out), B, T, L, _p(ws), ws.numel(),
                                      self._stream()))
        return out

    def encode_onehot(self, onehot, mask=None):
        """General float input [B,T,L,4] (any values, not only the six site vectors)."""
        onehot = self._f32(onehot)
        B, T, L, V = onehot.shape
        assert V == 4
        mask = self._u8(mask)
        out = torch.empty((B, T, L // self.K, self.D), dtype=torch.float32, device=self.device)
        ws = self.workspace(B, T, L)
        self._chk(self.lib.nnj_encode(self.h, None, _p(onehot), _p(mask), _p(out), B, T, L, _p(ws), ws.numel(),
                                      self._stream()))
        return out

    def pair_scores_full(self, state, mask=None):
        state = self._f32(state)
        B, n, Ct, _ = state.shape
        L = Ct * self.K                                          # sites
        mask = self._u8(mask)
        out = torch.empty((B, n * (n - 1) // 2), dtype=torch.float32, device=self.device)
        ws = self.workspace(B, n, L)
        self._chk(self.lib.nnj_pair_scores_full(self.h, _p(state), _p(mask), _p(out), B, n, L, _p(ws), ws.numel(),
                                                self._stream()))
        self._session_out(state)           # the rows of `state` now live in a session
        return out

    def pair_scores_incr(self, state, mask, ij_prev, logits_prev):
        state = self._f32(state)
        B, n, Ct, _ = state.shape
        L = Ct * self.K                                          # sites
        mask = self._u8(mask)
        ij_prev = self._i32(ij_prev)
        logits_prev = self._f32(logits_prev)
        assert tuple(logits_prev.shape) == (B, (n + 1) * n // 2)
        out = torch.empty((B, n * (n - 1) // 2), dtype=torch.float32, device=self.device)
        self._session_in(state)
        ws = self.workspace(B, n + 1, L)
        self._chk(self.lib.nnj_pair_scores_incr(self.h, _p(state), _p(mask), _p(ij_prev), _p(logits_prev), _p(out),
                                                B, n, L, _p(ws), ws.numel(), self._stream()))
        return out

    def score_index_map(self, ij_prev, n):
        ij_prev = self._i32(ij_prev)
        B = ij_prev.shape[0]
        out = torch.empty((B, n * (n - 1) // 2), dtype=torch.int64, device=self.device)
        self._chk(self.lib.nnj_score_index_map(self.h, _p(ij_prev), _p(out), B, n, self._stream()))
        return out

    def aggregate(self, state, ij):
        state = self._f32(state)
        B, n, Ct, _ = state.shape
        L = Ct * self.K                                          # sites
        ij = self._i32(ij)
        out = torch.empty((B, 1, Ct, self.D), dtype=torch.float32, device=self.device)
        self._session_in(state)
        ws = self.workspace(B, n, L)
        self._chk(self.lib.nnj_aggregate(self.h, _p(state), _p(ij), _p(out), B, n, L, _p(ws), ws.numel(),
                                         self._stream()))
        return out

    def env_step(self, state, ij):
        state = self._f32(state)
        B, n, Ct, _ = state.shape
        L = Ct * self.K                                          # sites
        ij = self._i32(ij)
        out = torch.empty((B, n - 1, Ct, self.D), dtype=torch.float32, device=self.device)
        self._session_in(state)
        in_session = getattr(self, "_sess_tensor", None) is state
        ws = self.workspace(B, n, L)
        self._chk(self.lib.nnj_env_step(self.h, _p(state), _p(ij), _p(out), B, n, L, _p(ws), ws.numel(),
                                        self._stream()))
        if in_session:
            self._session_out(out)
        return out

    def select_pair(self, logits, n):
        logits = self._f32(logits)
        B = logits.shape[0]
        ij = torch.empty((B, 2), dtype=torch.int32, device=self.device)
        gap = torch.empty((B,), dtype=torch.float32, device=self.device)
        self._chk(self.lib.nnj_select_pair(self.h, _p(logits), _p(ij), _p(gap), B, n, self._stream()))
        return ij, gap

    def rollout_argmax(self, codes, mask=None, forced_merges=None, want_trace=False, want_state=False, out=None):
        """Device-resident Argmax rollout.  Returns dict of device tensors (no host sync).
        `out`: the dict returned by an earlier call with the same shapes -- its tensors are written again instead
        of fresh ones being allocated (a small-batch call whose arguments all repeat is replayed as a hipGraph)."""
        codes = self._u8(codes)
        B, T, L = codes.shape
        mask = self._u8(mask)
        fm = None if forced_merges is None else self._i32(forced_merges)
        if out is not None and not want_trace and not want_state and tuple(out["merges"].shape) == (B, T - 1, 2):
            merges = out["merges"]
        else:
            merges = torch.empty((B, T - 1, 2), dtype=torch.int32, device=self.device)
        total = sum(n * (n - 1) // 2 for n in range(2, T + 1))
        trace = torch.empty((B, total), dtype=torch.float32, device=self.device) if want_trace else None
        gap = torch.empty((B, T - 1), dtype=torch.float32, device=self.device) if want_trace else None
        st = torch.empty((B, T, L // self.K, self.D), dtype=torch.float32, device=self.device) if want_state else None
        ws = self.workspace(B, T, L)
        self._chk(self.lib.nnj_rollout_argmax(self.h, _p(codes), _p(mask), B, T, L, _p(fm), _p(merges), _p(trace),
                                              _p(gap), _p(st), _p(ws), ws.numel(), self._stream()))
        out = dict(merges=merges)
        if want_trace:
            out["logits"] = trace
            out["top2_gap"] = gap
        if want_state:
            out["state"] = st
        return out

    def step(self, state, mask, ij, logits_prev, forced_next=None):
        """One loop iteration on the device: merge `ij` (env.step), score the new pairs and assemble the table
        (decode_zxr), argmax.  state [B,n+1,L,D] -> dict(state [B,n,L,D], logits [B,P(n)], ij [B,2], top2_gap [B])."""
        state = self._f32(state)
        B, n1, Ct, _ = state.shape
        L = Ct * self.K                                          # sites
        n = n1 - 1
        mask = self._u8(mask)
        ij = self._i32(ij)
        lp = self._f32(logits_prev)
        fn = None if forced_next is None else self._i32(forced_next)
        st = torch.empty((B, n, Ct, self.D), dtype=torch.float32, device=self.device)
        lo = torch.empty((B, n * (n - 1) // 2), dtype=torch.float32, device=self.device)
        cij = torch.empty((B, 2), dtype=torch.int32, device=self.device)
        gap = torch.empty((B,), dtype=torch.float32, device=self.device)
        self._session_in(state)
        ws = self.workspace(B, n1, L)
        self._chk(self.lib.nnj_step(self.h, _p(state), _p(mask), _p(ij), _p(lp), _p(fn), _p(st), _p(lo), _p(cij),
                                    _p(gap), B, n, L, _p(ws), ws.numel(), self._stream()))
        self._session_out(st)
        return dict(state=st, logits=lo, ij=cij, top2_gap=gap)

    def topology_hash(self, merges):
        """64-bit topology keys [B] (int64 view of the uint64 keys) of merge lists [B,T-1,2]: equal <=> same topo_repr."""
        merges = self._i32(merges)
        B, tm1, _ = merges.shape
        keys = torch.empty((B,), dtype=torch.int64, device=self.device)
        self._chk(self.lib.nnj_topology_hash(self.h, _p(merges), B, tm1 + 1, _p(keys), self._stream()))
        return keys

    def set_concurrency(self, streams: int):
        """Sub-batches of a rollout that run on streams of the library's own (include/nnj.h nnj_set_concurrency)."""
        self._chk(self.lib.nnj_set_concurrency(self.h, int(streams)))

    def check_numeric(self):
        """Synchronises the stream and raises if any pair-score table written since the last call held a
        non-finite score (an operand left the range of the fp16 pieces, include/nnj.h): call once per batch
        before trusting merge lists fetched to the host."""
        flag = C.c_int32(0)
        self._chk(self.lib.nnj_numeric_status(self.h, C.byref(flag), self._stream()))
        if flag.value & 8:
            raise ValueError(
                "tree likelihood: a merge list held a pair with i >= j or a position outside the live list "
                "(NNJ_STATUS_BAD_MERGE): the likelihoods of this call are invalid")
        if flag.value & 4:
            raise RuntimeError(
                "two-pass NJ step: a merge had no source for its attention weights and no fallback was launched "
                "(NNJ_FLAG_MERGE_WEIGHTS, an internal error): the results of this batch are invalid")
        if flag.value & 2:
            raise RuntimeError(
                "a partner-wave barrier of a HIP kernel timed out (NNJ_STATUS_BARRIER_TIMEOUT): the results of "
                "this batch were computed on incomplete LDS images and are invalid")
        if flag.value & 1:
            raise FloatingPointError(
                "non-finite pair scores: an operand magnitude exceeded the fp16 piece range (65504) of the "
                "f16x3 GEMMs; the results of this batch are invalid")

    def rollout_sample(self, codes, mask, uniforms, temperature=1.0, replicas=None, want_trace=False):
        """Sampled rollouts (NeuralNJ-MC device part).  `uniforms` float [B,T-1] in [0,1).
        replicas=None: codes [B,T,L] are B alignments.  replicas=B: codes [1,T,L] is ONE alignment,
        encoded once and rolled out B times."""
        codes = self._u8(codes)
        n_enc, T, L = codes.shape
        B = n_enc if replicas is None else int(replicas)
        if replicas is not None and n_enc != 1:
            raise ValueError("replicas needs a single alignment (codes [1,T,L])")
        mask = self._u8(mask)
        u = self._f32(uniforms)
        assert tuple(u.shape) == (B, T - 1)
        merges = torch.empty((B, T - 1, 2), dtype=torch.int32, device=self.device)
        total = sum(n * (n - 1) // 2 for n in range(2, T + 1))
        trace = torch.empty((B, total), dtype=torch.float32, device=self.device) if want_trace else None
        ws = self.workspace(B, T, L)
        self._chk(self.lib.nnj_rollout_sample(self.h, _p(codes), _p(mask), B, T, L, 1 if replicas is not None else B,
                                              _p(u), C.c_float(float(temperature)), _p(merges), _p(trace), _p(ws),
                                              ws.numel(), self._stream()))
        out = dict(merges=merges)
        if want_trace:
            out["logits"] = trace
        return out

    # ------------------------------------------------------------------ profiling
    def profile_enable(self, on=True):
        self._chk(self.lib.nnj_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        k = self.lib.nnj_profile_kinds()
        ms = (C.c_double * k)()
        cnt = (C.c_int64 * k)()
        dropped = C.c_int64(0)
        self._chk(self.lib.nnj_profile_dropped(self.h, C.byref(dropped)))
        self._chk(self.lib.nnj_profile_read(self.h, ms, cnt, k))
        if dropped.value:
            raise RuntimeError(f"{dropped.value} kernel launches could not be timed (event creation failed)")
        return {self.lib.nnj_profile_kind_name(i).decode(): (ms[i], cnt[i]) for i in range(k)}

    def debug_encoder_stop(self, stage):
        self._chk(self.lib.nnj_debug_encoder_stop(self.h, stage))
