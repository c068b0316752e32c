#!/usr/bin/env python3
"""Loop skeleton of one kernel in a `hipcc -S --cuda-device-only` listing: per basic block the instruction mix
(MFMA, other vector ALU, transcendental, LDS, vector memory, scalar, waits, nops), or with --dump the memory
operations, waits and branches in program order.  Usage: isa_skel.py nnj.s 'k_inc_score_w<3, true>' [--dump]"""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.split("\n")


def main():
    path, want = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    funcs = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l))]
    dem = demangle([f for _, f in funcs])
    hit = [(i, f, d) for (i, f), d in zip(funcs, dem) if want in d]
    if not hit:
        print("no kernel matches; candidates:")
        for d in sorted(set(dem)):
            print("  ", d[:120])
        return
    start, fn, dn = hit[0]
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    print(dn[:160])
    block, stats, order = "entry", {}, []
    trans = ("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt", "v_sin", "v_cos")
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                block = m.group(1)
            if not m:
                continue
        if re.match(r"^\.LBB", s):
            continue
        op = s.split()[0]
        st = stats.setdefault(block, dict(mfma=0, valu=0, trans=0, lds=0, vmem=0, salu=0, wait=0, nop=0, bar=0, br=0, n=0))
        if block not in order:
            order.append(block)
        st["n"] += 1
        if op.startswith("v_mfma"):
            st["mfma"] += 1
        elif op.startswith(trans):
            st["trans"] += 1
        elif op.startswith("v_"):
            st["valu"] += 1
        elif op.startswith("ds_"):
            st["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            st["vmem"] += 1
        elif op == "s_waitcnt":
            st["wait"] += 1
        elif op == "s_nop":
            st["nop"] += 1
        elif op == "s_barrier":
            st["bar"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            st["br"] += 1
        elif op.startswith("s_"):
            st["salu"] += 1
        if dump and (op.startswith(("global_", "buffer_", "scratch_", "s_waitcnt", "s_cbranch", "s_branch", "s_barrier", "ds_")) or op.startswith("v_mfma")):
            print(f"  {block:12s} {s[:110]}")
    print(f"{'block':12s} {'n':>5s} {'mfma':>5s} {'valu':>5s} {'trans':>5s} {'lds':>4s} {'vmem':>4s} {'salu':>4s} {'wait':>4s} {'nop':>4s} {'bar':>3s} {'br':>3s}")
    for b in order:
        st = stats[b]
        if st["n"] < 8:
            continue
        print(f"{b:12s} {st['n']:5d} {st['mfma']:5d} {st['valu']:5d} {st['trans']:5d} {st['lds']:4d} {st['vmem']:4d} {st['salu']:4d} {st['wait']:4d} {st['nop']:4d} {st['bar']:3d} {st['br']:3d}")


main()
