"""Builds alphazero_amd/lib/libaz_engine.so with hipcc for gfx950 (in-tree, so the .so travels)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(_HERE, "csrc", "az_engine.hip"), os.path.join(_HERE, "csrc", "az_tower.hip"),
        os.path.join(_HERE, "csrc", "az_tower_f32.hip"), os.path.join(_HERE, "csrc", "az_smallnet.hip")]
DEPS = [*SRCS, os.path.join(_HERE, "csrc", "az_games.hip.h"),
        os.path.join(os.path.dirname(_HERE), "include", "az_engine.h")]
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "libaz_engine.so")

# -ffp-contract=off + correctly rounded f32 divide/sqrt: the PUCT / prior arithmetic must round exactly
# like the reference's torch-CPU chain (bit-exactness notes).
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-int-to-void-pointer-cast"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *FLAGS, "-o", LIB, *SRCS]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
