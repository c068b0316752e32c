"""Parameter inventory, seeded initialiser and packing for the NeuralNJ hot path.

The order of `param_spec` is the `state_dict()` order of the reference's
`PhyloATTN` (reference model.py:25-60, msa_modules.py:51-53,107,144-145,
axial_attention.py:24-28,159-163): per encoder layer
row-attention {k,v,q,out}_proj + layer_norm, column-attention {k,v,q,out}_proj +
layer_norm, FFN fc1/fc2 + layer_norm; then embed.0, embed.2, h_linear_last,
g_linear_last, g_attn_q, g_attn_k, s_out.0, s_out.2.  The packed fp32 vector
handed to `nnj_load_weights` is the concatenation of these tensors (row-major,
`weight[out, in]`) in exactly this order.

The initialiser is torch-independent (numpy Philox) so that the golden-fixture
generator, the CPU oracle and the HIP library all see bit-identical weights.
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, Tuple

import numpy as np


def model_dims(cfgs) -> dict:
    m = cfgs.model
    D = int(m.embed_dim)
    return dict(
        vocab=int(m.vocab_size),
        patch=int(m.patch_size),
        D=D,
        H=int(m.num_enc_heads),
        F=4 * D,  # reference model.py:28 (ffn_embedding_dim = embed_dim * 4)
        layers=int(m.num_enc_layers),
    )


def param_spec(cfgs) -> List[Tuple[str, Tuple[int, ...]]]:
    d = model_dims(cfgs)
    D, F = d["D"], d["F"]
    spec: List[Tuple[str, Tuple[int, ...]]] = []
    for l in range(d["layers"]):
        p = f"seq_emb_layers.{l}."
        for attn in ("row_self_attention", "column_self_attention"):
            for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
                spec.append((f"{p}{attn}.layer.{proj}.weight", (D, D)))
                spec.append((f"{p}{attn}.layer.{proj}.bias", (D,)))
            spec.append((f"{p}{attn}.layer_norm.weight", (D,)))
            spec.append((f"{p}{attn}.layer_norm.bias", (D,)))
        spec.append((f"{p}feed_forward_layer.layer.fc1.weight", (F, D)))
        spec.append((f"{p}feed_forward_layer.layer.fc1.bias", (F,)))
        spec.append((f"{p}feed_forward_layer.layer.fc2.weight", (D, F)))
        spec.append((f"{p}feed_forward_layer.layer.fc2.bias", (D,)))
        spec.append((f"{p}feed_forward_layer.layer_norm.weight", (D,)))
        spec.append((f"{p}feed_forward_layer.layer_norm.bias", (D,)))
    spec.append(("embed.0.weight", (D, d["vocab"] * d["patch"])))
    spec.append(("embed.0.bias", (D,)))
    spec.append(("embed.2.weight", (D, D)))
    spec.append(("embed.2.bias", (D,)))
    for name in ("h_linear_last", "g_linear_last", "g_attn_q", "g_attn_k"):
        spec.append((f"{name}.weight", (D, D)))
        spec.append((f"{name}.bias", (D,)))
    spec.append(("s_out.0.weight", (D, D)))
    spec.append(("s_out.0.bias", (D,)))
    spec.append(("s_out.2.weight", (1, D)))
    spec.append(("s_out.2.bias", (1,)))
    return spec


def num_params(cfgs) -> int:
    return int(sum(int(np.prod(s)) for _, s in param_spec(cfgs)))


# Gains applied on top of U(-1/sqrt(fan_in), 1/sqrt(fan_in)).  Plain 1/sqrt(fan_in)
# weights give near-uniform softmaxes, which would hide scaling / masking mistakes
# in a parity test, so the "sharp" style widens the attention and gate projections.
_STYLES = {
    "plain": dict(attn_qk=1.0, glob_qk=1.0, attn_v=1.0, lin=1.0, gate=1.0, ln_jitter=0.0, bias=1.0),
    "sharp": dict(attn_qk=4.0, glob_qk=1.4, attn_v=1.5, lin=1.5, gate=3.0, ln_jitter=0.3, bias=1.0),
}


def seeded_state(cfgs, seed: int, style: str = "sharp") -> Dict[str, np.ndarray]:
    """Deterministic fp32 parameters keyed like the reference's state_dict."""
    g = _STYLES[style]
    rng = np.random.Generator(np.random.Philox(key=[int(seed), 0x6E6E6A]))
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_spec(cfgs):
        if "layer_norm" in name:
            if name.endswith("weight"):
                a = 1.0 + g["ln_jitter"] * (2.0 * rng.random(shape) - 1.0)
            else:
                a = g["ln_jitter"] * (2.0 * rng.random(shape) - 1.0)
        else:
            fan_in = shape[-1] if name.endswith("weight") else None
            if fan_in is None:
                # bias: bound by the fan_in of its weight, which precedes it in the spec
                wname = name[: -len("bias")] + "weight"
                fan_in = out[wname].shape[-1]
            bound = 1.0 / np.sqrt(float(fan_in))
            if ".q_proj." in name or ".k_proj." in name:
                gain = g["attn_qk"]
            elif name.startswith(("g_attn_q", "g_attn_k")):
                gain = g["glob_qk"]
            elif ".v_proj." in name or ".out_proj." in name:
                gain = g["attn_v"]
            elif name.startswith(("h_linear_last", "g_linear_last")):
                gain = g["gate"]
            else:
                gain = g["lin"]
            if name.endswith("bias"):
                gain = g["bias"]
            a = gain * bound * (2.0 * rng.random(shape) - 1.0)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def pack(cfgs, state: Dict[str, np.ndarray]) -> np.ndarray:
    """Concatenate a state_dict-like mapping into the flat fp32 vector of the C ABI."""
    parts = []
    for name, shape in param_spec(cfgs):
        t = state[name]
        if hasattr(t, "detach"):
            t = t.detach().cpu().numpy()
        t = np.asarray(t, dtype=np.float32)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        parts.append(t.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def unpack(cfgs, flat: np.ndarray) -> Dict[str, np.ndarray]:
    flat = np.asarray(flat, dtype=np.float32).reshape(-1)
    out, off = {}, 0
    for name, shape in param_spec(cfgs):
        n = int(np.prod(shape))
        out[name] = flat[off : off + n].reshape(shape).copy()
        off += n
    if off != flat.size:
        raise ValueError(f"packed vector has {flat.size} floats, spec needs {off}")
    return out


def digest(flat: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(flat, dtype=np.float32).tobytes()).hexdigest()
