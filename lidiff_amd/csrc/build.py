"""Builds liblidiff_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m lidiff_amd.csrc.build [--force]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["coords.hip", "spconv.hip", "spconv_rows.hip", "spconv_bf16.hip", "spconv_split3.hip", "norm.hip", "step.hip", "head.hip"]
# step.hip restates a sequence of separately rounded torch launches: no fused multiply-adds there
EXTRA_FLAGS = {"step.hip": ["-ffp-contract=off"]}
HEADERS = ["common.h", "spconv.h", os.path.join("..", "..", "include", "lidiff_amd.h")]
LIB = os.path.join(HERE, "liblidiff_amd.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the lidiff_amd HIP extension cannot be built")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    objs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
               "-c", os.path.join(HERE, src), "-o", obj] + EXTRA_FLAGS.get(src, [])
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.run(cmd, check=True)
        objs.append(obj)
    subprocess.run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
