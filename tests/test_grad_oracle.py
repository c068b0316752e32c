"""CPU: the gradient oracle (oracle/grad_oracle.py: the reference's forward restated in torch, float32 / float64) against
the golden gradients captured from the reference itself (tests/golden/grad_*.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))


@pytest.mark.parametrize("name,dtype,tol", [("b2_t6_l48_pad", torch.float32, 2e-3), ("b2_t6_l48_pad", torch.float64, 4e-3),
                                            ("b2_t8_l128_s0", torch.float64, 2e-3),
                                            # train() mode, dropout 0.4 with the masks the fixture recorded
                                            ("train_b2_t6_l48_pad", torch.float32, 2e-3), ("train_b2_t6_l48_pad", torch.float64, 4e-3),
                                            # the reference's default model: 32 features, 4 heads, 3 layers, patch 4
                                            ("dim32_b2_t6_l48_pad", torch.float32, 2e-3), ("dim32_b2_t6_l48_pad", torch.float64, 4e-3)])
def test_gradient_oracle_reproduces_the_reference(name, dtype, tol):
    import grad_oracle
    from neuralnj_amd import synth, utils, weights
    z = np.load(os.path.join(HERE, "golden", f"grad_{name}.npz"), allow_pickle=True)
    cfgs = utils.shipped_config()
    cfgs.model.num_enc_layers = int(z["layers"])
    heads, patch = (int(z["heads"]), int(z["patch"])) if "heads" in z.files else (8, 1)
    if "dim" in z.files:
        cfgs.model.embed_dim, cfgs.model.num_enc_heads, cfgs.model.patch_size = int(z["dim"]), heads, patch
    st = weights.seeded_state(cfgs, int(z["wseed"]), str(z["style"]))
    sd = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in st.items()}
    import helpers
    drop = helpers.recorded_dropout(z)
    loss, tables = grad_oracle.reinforce_loss(sd, synth.codes_to_onehot(z["codes"]), z["mask"], z["merges"], z["tree_scores"],
                                              float(z["baseline"]), float(z["temperature"]), float(z["strength"]),
                                              int(z["layers"]), dtype, drop=drop, heads=heads, patch=patch)
    assert drop is None or drop.calls() == drop.expected
    assert abs(float(loss.detach()) - float(z["loss"])) <= 2e-4 * max(1.0, abs(float(z["loss"])))
    got_t = torch.cat([t.detach().reshape(t.shape[0], -1) for t in tables], 1).numpy()
    assert np.abs(got_t - z["tables"]).max() <= 1e-4 * np.abs(z["tables"]).max()
    loss.backward()
    ref = z["grads"]
    gmax = float(np.abs(ref).max())
    zero = ("row_self_attention.layer.k_proj.bias", "column_self_attention.layer.k_proj.bias", "g_attn_k.bias", "s_out.2.bias")
    off, worst = 0, 0.0
    for k, p in sd.items():
        n = p.numel()
        want = ref[off:off + n].reshape(tuple(p.shape))
        off += n
        got = p.grad.numpy()
        if k.endswith(zero):
            assert np.abs(got).max() <= 1e-3 * gmax and np.abs(want).max() <= 1e-3 * gmax, k
            continue
        err = float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-7 * gmax)
        worst = max(worst, err)
        assert err <= tol, f"{k}: {err:.2e}"
    assert off == ref.size
    print(f"{name} {dtype}: worst per-tensor difference from the reference's gradient {worst:.2e}")
