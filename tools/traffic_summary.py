#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass).
usage: python3 tools/traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [traffic.json]
Corrections of MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide streaming reads at 64 bytes -> doubled; WRITE_SIZE is taken as is.
The optional fourth argument also writes profiles/traffic.json's per-kind view used by bench.py."""
import collections
import csv
import json
import sys

KIND = [("k_embed", "k_embed"), ("k_key_classes", "k_embed"), ("k_qkv6", "k_qkv6"), ("k_row_s", "k_row_s"),
        ("k_row_pv", "k_row_pv"), ("k_tok1", "k_tok1"), ("k_ffn", "k_ffn"), ("k_row_xf", "k_row_xf"),
        ("k_pair_alpha", "k_pair_alpha"), ("k_alpha_softmax", "k_alpha_softmax"), ("k_pair_score", "k_pair_score"),
        ("k_assemble_argmax", "k_assemble_argmax"), ("k_agg_alpha", "k_agg_alpha"), ("k_agg_finish", "k_agg_finish"),
        ("k_inc_alpha", "k_pair_alpha_incr"), ("k_inc_score", "k_pair_score_incr"),
        ("k_step_alpha", "k_pair_alpha_incr"), ("k_step_softmax", "k_alpha_softmax"), ("k_pair_xp", "k_step_small"),
        ("k_agg_dot", "k_step_small"), ("k_agg_am", "k_step_small"), ("k_beta_sum", "k_step_small")]


def read(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k] += float(r["Counter_Value"]) * 1024.0
        n[k] += 1
    return tot, n


fetch, nf = read(sys.argv[1], "FETCH_SIZE")
write, nw = read(sys.argv[2], "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    launches = max(nf.get(k, 0), nw.get(k, 0)) or 1
    f, w = 2.0 * fetch.get(k, 0) / launches, write.get(k, 0) / launches
    kernels[k] = dict(launches=launches, fetch_bytes_per_launch_corrected=f, write_bytes_per_launch=w,
                      hbm_bytes_per_launch=f + w)
note = ("FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes of tools/prof_run.py; KiB units; FETCH_SIZE "
        "doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE taken as is.")
json.dump(dict(note=note, kernels=kernels), open(sys.argv[3], "w"), indent=1)
if len(sys.argv) > 4:
    per = {}
    for k, v in kernels.items():
        kind = next((kd for pre, kd in KIND if k.startswith(pre)), None)
        if kind is None:
            continue
        e = per.setdefault(kind, dict(hbm=0.0, launches=0))
        e["hbm"] += v["hbm_bytes_per_launch"] * v["launches"]
        e["launches"] += v["launches"]
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from neuralnj_amd import build as nbuild
    out = dict(source=f"{sys.argv[3]} ({note})", source_hash=nbuild.source_hash(),
               per_kind={k: dict(hbm_bytes_per_launch=e["hbm"] / e["launches"], launches=e["launches"])
                         for k, e in per.items()})
    json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, v in list(kernels.items())[:24]:
    print(f"{k[:40]:40s} x{v['launches']:3d}  fetch {v['fetch_bytes_per_launch_corrected'] / 1e9:8.3f} GB  write {v['write_bytes_per_launch'] / 1e9:8.3f} GB")
