#!/bin/bash
# rocprofv3 kernel trace of one complete Search round at 200 x 4096 (tools/search_profile.py): the likelihood kernels'
# time per launch next to their algorithmic bytes.  Output: gpurun_out/search_profile/{kernel_stats.csv, summary.txt}
export TMPDIR=/tmp
O=$(pwd)/gpurun_out/search_profile
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $(pwd)/tools/search_profile.py 2 > $O/run.txt 2> $O/err.txt
S=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp $S $O/kernel_stats.csv
rm -rf $O/stats
python3 - > $O/summary.txt <<PY
import csv
T, L, NC = 200, 4096, 4
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
# algorithmic bytes PER TREE of one launch: partial likelihoods are [join][cat x state][site] fp64
part = (T - 1) * NC * 4 * 8 * L            # all partials of a tree
model = {"k_lik_down": 2 * part + T * L,   # every join writes its partial once, every internal child is read once
         "k_lik_outer": 3 * part,          # reads D of the sibling and O of the parent, writes O
         "k_lik_newton": 2 * NC * 4 * 8 * L}  # per (tree, edge) and Newton iteration: O_v and D_v of the edge
print(open("$O/run.txt").read())
print(f"{'kernel':40s} {'calls':>6s} {'avg_us':>9s} {'total_ms':>9s} {'share':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"{r['Name'][:40]:40s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
print()
for r in rows:
    for k, bytes_tree in model.items():
        if r["Name"].startswith(k):
            avg = float(r["AverageNs"]) * 1e-9
            print(f"{k}: algorithmic bytes per tree and launch {bytes_tree/1e6:.1f} MB"
                  + (" (per edge and iteration)" if k == "k_lik_newton" else "")
                  + f"; average launch {avg*1e6:.1f} us")
PY
cat $O/summary.txt
