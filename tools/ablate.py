#!/usr/bin/env python3
"""Timing ablations of k_tok1's column attention (results are WRONG on purpose; timing only).
debug_stop = 16*bits: bit0 skip the whole column-attention block of k_tok1 (the row part still runs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnj_amd import synth, utils, weights
from neuralnj_amd._lib import Nnj
B = 128
cfgs = utils.shipped_config()
g = Nnj(cfgs, "cuda:0"); g.load_weights(weights.pack(cfgs, weights.seeded_state(cfgs, 0, "sharp")))
codes = torch.from_numpy(synth.synth_codes(B, 50, 1024, seed=1, gap_frac=0.2)).cuda()
for bits in (0, 1):
    g.debug_encoder_stop(16 * bits)
    g.encode(codes); torch.cuda.synchronize()
    g.profile_enable(True)
    for _ in range(2): g.encode(codes)
    torch.cuda.synchronize()
    p = g.profile_read(); g.profile_enable(False)
    print(f"bits={bits:2d} tok1 ms/launch={p['k_tok1'][0]/p['k_tok1'][1]:.3f}  tok2={p['k_tok2'][0]/p['k_tok2'][1]:.3f} row={p['k_row_attn'][0]/p['k_row_attn'][1]:.3f}", flush=True)
