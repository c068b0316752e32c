"""PhyloATTN: host mirror of the reference's model.py surface over libnnj_hip.so.

Same constructor argument (`cfgs.model.*`), same `state_dict()` keys and shapes as the
reference's PhyloATTN (reference model.py:11-60), so a reference checkpoint loads with
`load_state_dict(ckpt['model_state_dict'], strict=True)`.  The parameters are plain
containers: every forward computation -- encode_zxr (model.py:67-88), decode_zxr
(model.py:158-209), aggregate (model.py:102-155) -- runs in the HIP library on the GPU.
There is no torch implementation of the maths here and no CPU fallback.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import weights as _weights
from ._lib import Nnj


class _AttnParams(nn.Module):          # holds {k,v,q,out}_proj like the reference's attention modules
    def __init__(self, d, heads=8):
        super().__init__()
        self.num_heads = heads             # (axial_attention.py:18,153)
        self.k_proj = nn.Linear(d, d)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)


class _FfnParams(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)


class _Residual(nn.Module):            # `.layer` + `.layer_norm`, as NormalizedResidualBlock names them
    def __init__(self, layer, d):
        super().__init__()
        self.layer = layer
        self.layer_norm = nn.LayerNorm(d)


class _AxialLayerParams(nn.Module):
    def __init__(self, d, f, heads=8):
        super().__init__()
        self.row_self_attention = _Residual(_AttnParams(d, heads), d)
        self.column_self_attention = _Residual(_AttnParams(d, heads), d)
        self.feed_forward_layer = _Residual(_FfnParams(d, f), d)


class PhyloATTN(nn.Module):
    def __init__(self, cfgs):
        super().__init__()
        m = cfgs.model
        self.cfgs = cfgs
        self.vocab_size = m.vocab_size
        self.patch_size = m.patch_size
        self.patch_num = m.fixed_length // m.patch_size if "fixed_length" in m else 0
        self.embed_dim = d = m.embed_dim
        self.num_enc_heads = m.num_enc_heads
        self.num_enc_layers = m.num_enc_layers
        self.dropout = 0.4                      # model.py:23; applied in train() mode with gradients (train_model.encode)
        self.seq_emb_layers = nn.ModuleList([_AxialLayerParams(d, 4 * d, int(m.num_enc_heads)) for _ in range(m.num_enc_layers)])
        self.embed = nn.Sequential(nn.Linear(m.vocab_size * m.patch_size, d), nn.GELU(), nn.Linear(d, d))
        self.h_linear_last = nn.Linear(d, d)
        self.g_linear_last = nn.Linear(d, d)
        self.g_attn_q = nn.Linear(d, d)
        self.g_attn_k = nn.Linear(d, d)
        self.s_out = nn.Sequential(nn.Linear(d, d), nn.GELU(), nn.Linear(d, 1))
        self._ctx = None
        self._packed_version = None
        self.batch_input = None
        self.seq_mask = None

    def model_params(self):
        return list(self.parameters())

    # ------------------------------------------------------------------ device context
    def _device(self):
        return next(self.parameters()).device

    def _context(self) -> Nnj:
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("PhyloATTN (neuralnj_amd) computes on the GPU only: call .to('cuda') first; "
                               "there is no CPU path")
        if self._ctx is None or self._ctx.device != dev:
            self._ctx = Nnj(self.cfgs, dev)
            self._packed_version = None
        plist = self.__dict__.get("_plist")              # (walking the module tree per call cost 0.5 ms)
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())
        version = sum(p._version for p in plist) + 7919 * id(self._ctx)
        if version != self._packed_version:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            self._ctx.load_weights(_weights.pack(self.cfgs, sd))
            self._packed_version = version
        return self._ctx

    # ------------------------------------------------------------------ reference API
    @staticmethod
    def onehot_to_codes(batch_input: torch.Tensor) -> torch.Tensor:
        """[B,T,L,4] one-hot (int8/float) -> [B,T,L] uint8 site codes (0..3, 4 = gap, 5 = pad)."""
        x = batch_input
        s = x.sum(-1)
        codes = torch.where(s == 4, torch.full_like(s, 4), torch.where(s == 0, torch.full_like(s, 5),
                                                                      x.argmax(-1).to(s.dtype)))
        lut = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [1, 1, 1, 1], [0, 0, 0, 0]],
                           dtype=x.dtype, device=x.device)
        if not torch.equal(lut[codes.long()], x):
            raise ValueError("encode_zxr: input is not made of the six site vectors of the reference's CHARS_DICT")
        return codes.to(torch.uint8)

    # ------------------------------------------------------------------ Finetune mode (gradients)
    def _wants_grad(self):
        """The reference fine-tunes by calling these same methods outside torch.no_grad() (finetune_rl_search.py:113,
        129, 164 with eval=False) and back-propagating through them: with gradients enabled the differentiable
        operators of train_model.py run (forward and backward kernels of libnnj_train_hip.so) -- with dropout when the
        module is in train() mode (train.py:435), without in eval() (the Finetune loop); under no_grad the fused
        inference kernels, which are the eval-mode forward (every no_grad call site of the reference calls
        agent.eval() first: finetune_rl_search.py:110, train.py:91,100)."""
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def encode_zxr(self, batch_input, batch_seq_mask=None):
        if self._wants_grad():
            from . import train_model
            if batch_input.dim() != 4 or batch_input.shape[-1] != self.vocab_size:
                raise ValueError("encode_zxr expects [B, rows, cols, vocab] one-hot input")
            return train_model.encode(self, batch_input, batch_seq_mask)
        ctx = self._context()
        if batch_input.dim() != 4 or batch_input.shape[-1] != self.vocab_size:
            raise ValueError("encode_zxr expects [B, rows, cols, vocab] one-hot input")
        num_cols = batch_input.shape[2]
        self.patch_num = math.ceil(num_cols / self.patch_size)
        x = batch_input.to(ctx.device)
        try:
            codes = self.onehot_to_codes(x)          # the six site vectors: 1 byte per site, LUT embed
        except ValueError:
            return ctx.encode_onehot(x.float(), batch_seq_mask)   # arbitrary float input: embed MLP on the device
        return ctx.encode(codes, batch_seq_mask)

    def decode_zxr(self, batch_input, batch_seq_mask=None, indices_to_prev_info=None):
        if self._wants_grad():
            from . import train_model
            self.batch_input = batch_input
            self.seq_mask = None if batch_seq_mask is None else ~batch_seq_mask[:, None, :: self.patch_size]
            if not self.patch_num:
                self.patch_num = math.ceil(batch_input.shape[2] / self.patch_size)
            scores = train_model.decode(self, batch_input, batch_seq_mask, indices_to_prev_info)
            return {"logits": scores, "distance": scores}
        ctx = self._context()
        actions_ij_prev, _score_indices_to_prev, logits_prev = indices_to_prev_info
        self.batch_input = batch_input
        self.seq_mask = None if batch_seq_mask is None else ~batch_seq_mask[:, None, :: self.patch_size]
        if logits_prev is None:
            scores = ctx.pair_scores_full(batch_input, batch_seq_mask)
        else:
            scores = self._prefetched_table(batch_input, batch_seq_mask, actions_ij_prev, logits_prev)
            if scores is None:
                # the old->new index map (utils.get_score_indices_to_prev) is recomputed on the device
                scores = ctx.pair_scores_incr(batch_input, batch_seq_mask, actions_ij_prev, logits_prev)
        # what env.step's fused device call needs (fused_env_step): the tensors of this decode, with their versions
        self._decoded = (batch_input, batch_input._version, batch_seq_mask, scores, scores._version)
        self._prefetch = None
        return {"logits": scores, "distance": scores}

    # The reference's loop alternates env.step (aggregate + compaction of the state) and decode_zxr (scores of the new
    # pairs + table).  On the device the two are ONE step (nnj_step: the merged row is produced inside the alpha pass
    # of the new pairs, include/nnj.h), so env.step hands the merge to `fused_env_step`, which runs that step and keeps
    # its table; the decode_zxr that follows returns the table if it is asked exactly the question the step answered
    # (the state env.step returned, unmodified; the same mask and previous table objects; the same merged pairs) and
    # computes it the ordinary way otherwise.
    def fused_env_step(self, state, ij_t):
        d = getattr(self, "_decoded", None)
        if d is None or self._wants_grad() or d[0] is not state or state._version != d[1] or d[3]._version != d[4]:
            return None
        ij32 = ij_t.to(torch.int32)
        r = self._context().step(state, d[2], ij32, d[3])
        self._prefetch = (r["state"], r["state"]._version, d[2], d[3], d[4], ij32, r["logits"])
        return r["state"]

    def _prefetched_table(self, batch_input, batch_seq_mask, ij_prev, logits_prev):
        p = getattr(self, "_prefetch", None)
        if p is None or batch_input is not p[0] or batch_input._version != p[1] or batch_seq_mask is not p[2] \
                or logits_prev is not p[3] or logits_prev._version != p[4]:
            return None
        ij = torch.as_tensor(ij_prev).to(p[5].device, torch.int32)
        if ij.shape != p[5].shape or not torch.equal(ij, p[5]):
            return None
        return p[6]

    def aggregate(self, x_i, x_j, ij_indices, batchwise_ij_indices=False):
        """reference model.py:102-155.  batchwise_ij_indices=True: one pair per alignment (env.step's form; x_i / x_j
        may be None = rows ij of the stashed state).  False: N pairs per alignment, x_i / x_j [B,N,C,D] and
        ij_indices 1-D [N] (the same pairs for every alignment, the first step's form) or 2-D [B,N] -- the form
        decode_gg uses; the hot path scores pairs through decode_zxr instead, this form is here for callers of the
        reference's method and runs one nnj_aggregate per pair."""
        if self.batch_input is None:
            raise RuntimeError("aggregate() needs the state stashed by a preceding decode_zxr()")
        st = self.batch_input
        dev = st.device
        ii, jj = ij_indices
        ii = torch.as_tensor(ii).to(dev).to(torch.int64)
        jj = torch.as_tensor(jj).to(dev).to(torch.int64)
        B = st.shape[0]
        if not batchwise_ij_indices:
            if x_i is None or x_j is None:
                raise ValueError("aggregate(batchwise_ij_indices=False) takes the pairs' rows x_i, x_j [B,N,C,D]")
            if ii.dim() == 1:                                  # 'fst step': the same N pairs for every alignment
                ii, jj = ii[None, :].expand(B, -1), jj[None, :].expand(B, -1)
            if ii.dim() != 2 or ii.shape != jj.shape or ii.shape[0] != B or x_i.shape[1] != ii.shape[1]:
                raise ValueError("ij_indices must be 1-D [N] or 2-D [B,N], matching x_i / x_j [B,N,C,D]")
            if self._wants_grad():
                from . import train_model
                return train_model.aggregate(self, st.contiguous(), x_i, x_j, ii.contiguous(), jj.contiguous())
            # A pair's rows are excluded from its own context (model.py:118-141), so a state whose rows i and j ARE
            # x_i and x_j gives nnj_aggregate exactly this pair: no assumption that x_i / x_j were gathered from the state
            ctx = self._context()
            bi = torch.arange(B, device=dev)
            out = []
            for k in range(ii.shape[1]):
                s_k = st.clone()
                s_k[bi, ii[:, k]] = x_i[:, k].to(s_k.dtype)
                s_k[bi, jj[:, k]] = x_j[:, k].to(s_k.dtype)
                out.append(ctx.aggregate(s_k, torch.stack([ii[:, k], jj[:, k]], dim=1)))
            return torch.cat(out, dim=1)
        if self._wants_grad():
            from . import train_model
            ii, jj = ii.view(-1, 1), jj.view(-1, 1)
            st = st.contiguous()
            from .train_ops import GatherRows
            xi = GatherRows.apply(st, ii) if x_i is None else x_i
            xj = GatherRows.apply(st, jj) if x_j is None else x_j
            return train_model.aggregate(self, st, xi, xj, ii, jj)
        ctx = self._context()
        return ctx.aggregate(st, torch.stack([ii.view(-1), jj.view(-1)], dim=1))

    # fast path used by neuralnj_amd.rollout
    def rollout_argmax(self, codes, mask=None, **kw):
        return self._context().rollout_argmax(codes, mask, **kw)
