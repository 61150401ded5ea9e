"""Builds the C++ / OpenMP restatement of MinkowskiEngine's CPU algorithm (oracle/cpp/me_cpu_ref.cpp) into
oracle/_build/libme_cpu_ref.so with g++; SGEMM comes from the MKL inside torch's libtorch_cpu.so (the BLAS the Python
oracle's GEMMs use as well).  TEST INFRASTRUCTURE: called by __graft_entry__.build() and, lazily, by oracle/me_cpp.py.

    python -m oracle.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpp", "me_cpu_ref.cpp")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libme_cpu_ref.so")


def torch_lib_dir() -> str:
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(SRC), os.path.getmtime(__file__)):
        return LIB
    gxx = shutil.which("g++")
    if gxx is None:
        raise RuntimeError("g++ not found: the C++ oracle cannot be built")
    os.makedirs(OUT_DIR, exist_ok=True)
    tl = torch_lib_dir()
    cmd = [gxx, "-O3", "-mavx2", "-mfma", "-std=c++17", "-fopenmp", "-fPIC", "-shared", SRC, "-o", LIB,
           f"-L{tl}", "-ltorch_cpu", f"-Wl,-rpath,{tl}"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
