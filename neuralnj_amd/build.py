"""Builds libnnj_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.

Staleness is decided by CONTENT, not by mtime: the sha256 of every source the library is compiled from
(csrc/*.hip, csrc/*.hpp, include/nnj.h) and of the compile command is stored next to the library; a snapshot
copied to another machine (fresh mtimes everywhere) is rebuilt only if a source really differs."""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnnj_hip.so")
STAMP = LIB + ".srchash"
SOURCES = ["nnj_api.hip", "nnj_step0_tu.hip"]
# No -ffast-math (nor -fassociative-math): the f16x3 operand split relies on exact IEEE subtractions
# (x - fp16(x)); with reassociation the pieces no longer add up and the parity tests fail at 4e-4.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]
# Per-translation-unit flags: the backend's scheduling strategy is a per-kernel choice that the toolchain offers per
# translation unit only (profiles/r05/ab_sched_strategies.txt: max-ilp orders the MFMA blocks of the two step-0 kernels better, 14.85 -> 13.8
# and 33.1 -> 32.7 ms per rollout, and makes k_tok1p spill).  Same instructions, bit-identical results.
EXTRA_FLAGS = {"nnj_step0_tu.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Wno-unused-function"]}


def source_files():
    """Every file the library is compiled from (all headers of csrc/ are included by nnj_api.hip)."""
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))
    files.append(os.path.join(HERE, "..", "include", "nnj.h"))
    return files


def source_hash() -> str:
    h = hashlib.sha256(" ".join(FLAGS + SOURCES + [f"{k}:{' '.join(v)}" for k, v in sorted(EXTRA_FLAGS.items())]).encode())
    for f in source_files():
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    compile_flags = [f for f in FLAGS if f != "-shared"]
    objs, procs = [], []
    for src in SOURCES:                      # one object per translation unit (its own flags), compiled side by side
        obj = os.path.join(CSRC, f"{os.path.splitext(src)[0]}.{os.getpid()}.o")     # (two builders at once must not share objects)
        cmd = [hipcc] + compile_flags + EXTRA_FLAGS.get(src, []) + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    tmp_lib = f"{LIB}.{os.getpid()}.tmp"
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + objs
    if verbose:
        print(" ".join(link))
    subprocess.run(link, cwd=CSRC, check=True)
    os.replace(tmp_lib, LIB)                 # atomic: a reader never maps a half-written library
    for obj in objs:
        os.remove(obj)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


# ---- libnnj_train_hip.so: the differentiable operators of the Finetune mode (include/nnj_train.h), a library of its
# own (the inference library and its source hash -- which profiles/traffic.json is keyed on -- do not depend on it)
TRAIN_CSRC = os.path.join(HERE, "csrc_train")
TRAIN_LIB = os.path.join(HERE, "libnnj_train_hip.so")
TRAIN_STAMP = TRAIN_LIB + ".srchash"


def train_source_hash() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(glob.glob(os.path.join(TRAIN_CSRC, "*"))) + [os.path.join(HERE, "..", "include", "nnj_train.h")]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build_train(force: bool = False, verbose: bool = False) -> str:
    stale = not os.path.exists(TRAIN_LIB) or not os.path.exists(TRAIN_STAMP)
    if not stale:
        with open(TRAIN_STAMP) as f:
            stale = f.read().strip() != train_source_hash()
    if not force and not stale:
        return TRAIN_LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", TRAIN_LIB, "nnj_train.hip"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=TRAIN_CSRC, check=True)
    with open(TRAIN_STAMP, "w") as f:
        f.write(train_source_hash() + "\n")
    return TRAIN_LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_train(force=True, verbose=True))
