"""Per-kind kernel times of the bench workload for a timing-only (knock-out) build: rollouts of B x 50 x 1024 on one stream with
the library's per-launch HIP events, no numeric check (a knock-out's tables are wrong by construction).
Usage: NNJ_LIB_PATH=ab_build/libX.so python tools/ko_time.py [B] [rollouts]"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from neuralnj_amd import synth, utils, weights  # noqa: E402
from neuralnj_amd._lib import Nnj  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfgs = utils.shipped_config()
g = Nnj(cfgs, "cuda:0")
g.load_weights(weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp")))
codes = torch.from_numpy(synth.synth_codes(B, 50, 1024, seed=1000, gap_frac=0.2)).cuda()
g.set_concurrency(1)
g.rollout_argmax(codes, None)
torch.cuda.synchronize()
g.profile_enable(True)
for _ in range(K):
    g.rollout_argmax(codes, None)
torch.cuda.synchronize()
prof = g.profile_read()
g.profile_enable(False)
ms = {k: v[0] / K for k, v in prof.items()}
tag = os.path.basename(os.environ.get("NNJ_LIB_PATH", "shipped"))
print(tag.ljust(12), " ".join(f"{n[2:]}={ms.get(n, 0):.1f}" for n in ("k_row_s", "k_row_pv", "k_tok1", "k_ffn", "k_qkv6", "k_pair_alpha", "k_pair_score",
                                                              "k_pair_alpha_incr", "k_pair_score_incr")), f"total={sum(ms.values()):.1f}")
