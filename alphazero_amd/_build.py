"""Builds alphazero_amd/lib/libaz_engine.so with hipcc for gfx950 (in-tree, so the .so travels)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(_HERE, "csrc", "az_engine.hip"), os.path.join(_HERE, "csrc", "az_tower.hip"),
        os.path.join(_HERE, "csrc", "az_tower_f32.hip"), os.path.join(_HERE, "csrc", "az_smallnet.hip"),
        os.path.join(_HERE, "csrc", "az_tower_h2.hip"), os.path.join(_HERE, "csrc", "az_tower_h1.hip"),
        os.path.join(_HERE, "csrc", "az_tower_w.hip")]
DEPS = [*SRCS, os.path.join(_HERE, "csrc", "az_games.hip.h"),
        os.path.join(os.path.dirname(_HERE), "include", "az_engine.h")]
LIB_DIR = os.path.join(_HERE, "lib")
# AZ_ENGINE_LIB: load another build of the library instead (A/B measurements of kernel variants: scripts/dev/build_variant.sh)
LIB = os.environ.get("AZ_ENGINE_LIB") or os.path.join(LIB_DIR, "libaz_engine.so")

# -ffp-contract=off + correctly rounded f32 divide/sqrt: the PUCT / prior arithmetic must round exactly
# like the reference's torch-CPU chain (bit-exactness notes).
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-int-to-void-pointer-cast"]
OBJ_DIR = os.path.join(LIB_DIR, "obj")  # per-source objects (git-ignored): only what changed is recompiled


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [d for d in DEPS if d not in SRCS]
    jobs, objs = [], []
    for src in SRCS:  # one object per source, the stale ones compiled side by side (no device code crosses a file)
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        newest = max(os.path.getmtime(f) for f in [src, *headers] if os.path.exists(f))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            cmd = [hipcc, *FLAGS, "-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
