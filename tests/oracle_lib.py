"""ctypes binding of the CPU oracle (oracle/nnj_oracle.c).  TEST INFRASTRUCTURE:
imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")


class _Cfg(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("vocab_size", "patch_size", "embed_dim", "num_heads", "num_layers", "device")]


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _load(name):
    path = os.path.join(ORACLE_DIR, "_build", name)
    if not os.path.exists(path):
        build_oracle()
    return C.CDLL(path)


def _ptr(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


class Oracle:
    """One handle of the CPU restatement. precision: 'f32' (reference precision) or 'f64'."""

    def __init__(self, cfgs, packed: np.ndarray, precision: str = "f32"):
        self.lib = _load("libnnj_oracle.so" if precision == "f32" else "libnnj_oracle64.so")
        m = cfgs.model
        self.cfg = _Cfg(int(m.vocab_size), int(m.patch_size), int(m.embed_dim),
                        int(m.num_enc_heads), int(m.num_enc_layers), 0)
        self.D = int(m.embed_dim)
        self.K = int(m.patch_size)
        self.V = int(m.vocab_size)
        self.h = C.c_void_p()
        self._chk(self.lib.nnjo_create(C.byref(self.cfg), C.byref(self.h)))
        packed = np.ascontiguousarray(packed, dtype=np.float32)
        self._chk(self.lib.nnjo_load_weights(self.h, _ptr(packed, C.c_float), C.c_size_t(packed.size)))

    def _chk(self, rc):
        if rc != 0:
            self.lib.nnjo_last_error.restype = C.c_char_p
            msg = self.lib.nnjo_last_error(self.h)
            raise RuntimeError(f"oracle error {rc}: {msg.decode() if msg else ''}")

    def __del__(self):
        try:
            if self.h:
                self.lib.nnjo_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def set_threads(self, n: int) -> int:
        return int(self.lib.nnjo_set_threads(C.c_int32(n)))

    @staticmethod
    def _mask(mask):
        return None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)

    def encode(self, onehot, mask=None, taps=False):
        onehot = np.ascontiguousarray(onehot, dtype=np.float32)
        B, T, L, _ = onehot.shape
        Cc = L // self.K
        out = np.empty((B, T, Cc, self.D), np.float32)
        m = self._mask(mask)
        tap_arrs = [np.empty_like(out) for _ in range(4)] if taps else None
        tp = None
        if taps:
            tp = (C.POINTER(C.c_float) * 4)(*[_ptr(a, C.c_float) for a in tap_arrs])
        self._chk(self.lib.nnjo_encode(self.h, _ptr(onehot, C.c_float), _ptr(m, C.c_uint8),
                                       _ptr(out, C.c_float), B, T, L, tp))
        return (out, tap_arrs) if taps else out

    def pair_scores_full(self, state, mask=None):
        state = np.ascontiguousarray(state, dtype=np.float32)
        B, n, Cc, D = state.shape
        L = Cc * self.K
        out = np.empty((B, n * (n - 1) // 2), np.float32)
        self._chk(self.lib.nnjo_pair_scores_full(self.h, _ptr(state, C.c_float), _ptr(self._mask(mask), C.c_uint8),
                                                 _ptr(out, C.c_float), B, n, L))
        return out

    def pair_scores_incr(self, state, mask, ij_prev, logits_prev, want_new=False):
        state = np.ascontiguousarray(state, dtype=np.float32)
        B, n, Cc, D = state.shape
        L = Cc * self.K
        ij_prev = np.ascontiguousarray(ij_prev, dtype=np.int32)
        logits_prev = np.ascontiguousarray(logits_prev, dtype=np.float32)
        assert logits_prev.shape == (B, (n + 1) * n // 2)
        out = np.empty((B, n * (n - 1) // 2), np.float32)
        new = np.empty((B, n), np.float32) if want_new else None
        self._chk(self.lib.nnjo_pair_scores_incr(self.h, _ptr(state, C.c_float), _ptr(self._mask(mask), C.c_uint8),
                                                 _ptr(ij_prev, C.c_int32), _ptr(logits_prev, C.c_float),
                                                 _ptr(out, C.c_float), B, n, L, _ptr(new, C.c_float)))
        return (out, new) if want_new else out

    def score_index_map(self, ij_prev, n):
        ij_prev = np.ascontiguousarray(ij_prev, dtype=np.int32)
        B = ij_prev.shape[0]
        out = np.empty((B, n * (n - 1) // 2), np.int64)
        self._chk(self.lib.nnjo_score_index_map(_ptr(ij_prev, C.c_int32), _ptr(out, C.c_int64), B, n))
        return out

    def aggregate(self, state, ij):
        state = np.ascontiguousarray(state, dtype=np.float32)
        B, n, Cc, D = state.shape
        ij = np.ascontiguousarray(ij, dtype=np.int32)
        out = np.empty((B, 1, Cc, D), np.float32)
        self._chk(self.lib.nnjo_aggregate(self.h, _ptr(state, C.c_float), _ptr(ij, C.c_int32),
                                          _ptr(out, C.c_float), B, n, Cc * self.K))
        return out

    def env_step(self, state, ij):
        state = np.ascontiguousarray(state, dtype=np.float32)
        B, n, Cc, D = state.shape
        ij = np.ascontiguousarray(ij, dtype=np.int32)
        out = np.empty((B, n - 1, Cc, D), np.float32)
        self._chk(self.lib.nnjo_env_step(self.h, _ptr(state, C.c_float), _ptr(ij, C.c_int32),
                                         _ptr(out, C.c_float), B, n, Cc * self.K))
        return out

    def select_pair(self, logits, n):
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        B = logits.shape[0]
        ij = np.empty((B, 2), np.int32)
        gap = np.empty((B,), np.float32)
        self._chk(self.lib.nnjo_select_pair(_ptr(logits, C.c_float), _ptr(ij, C.c_int32), _ptr(gap, C.c_float), B, n))
        return ij, gap

    def rollout_argmax(self, onehot, mask=None, forced_merges=None, want_state=False):
        onehot = np.ascontiguousarray(onehot, dtype=np.float32)
        B, T, L, _ = onehot.shape
        Cc = L // self.K
        total = sum(n * (n - 1) // 2 for n in range(2, T + 1))
        merges = np.empty((B, T - 1, 2), np.int32)
        trace = np.empty((B, total), np.float32)
        gap = np.empty((B, T - 1), np.float32)
        st = np.empty((B, T, Cc, self.D), np.float32) if want_state else None
        fm = None if forced_merges is None else np.ascontiguousarray(forced_merges, dtype=np.int32)
        self._chk(self.lib.nnjo_rollout_argmax(self.h, _ptr(onehot, C.c_float), _ptr(self._mask(mask), C.c_uint8),
                                               B, T, L, _ptr(fm, C.c_int32), _ptr(merges, C.c_int32),
                                               _ptr(trace, C.c_float), _ptr(gap, C.c_float), _ptr(st, C.c_float)))
        res = dict(merges=merges, logits=trace, top2_gap=gap)
        if want_state:
            res["state"] = st
        return res

    def rollout_sample(self, onehot, mask, uniforms, temperature=1.0):
        onehot = np.ascontiguousarray(onehot, dtype=np.float32)
        B, T, L, _ = onehot.shape
        total = sum(n * (n - 1) // 2 for n in range(2, T + 1))
        merges = np.empty((B, T - 1, 2), np.int32)
        trace = np.empty((B, total), np.float32)
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        assert u.shape == (B, T - 1)
        self._chk(self.lib.nnjo_rollout_sample(self.h, _ptr(onehot, C.c_float), _ptr(self._mask(mask), C.c_uint8),
                                               B, T, L, _ptr(u, C.c_float), C.c_float(float(temperature)),
                                               _ptr(merges, C.c_int32), _ptr(trace, C.c_float)))
        return dict(merges=merges, logits=trace)
