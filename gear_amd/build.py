"""Builds gear_amd/libgear_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgear_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in
           ("gear_hip.hip", "gear_kernels.hpp", "gear_device.hpp", "gear_scene.hpp", "gear_types.hpp")]
HEADER = os.path.join(_HERE, "..", "include", "gear_hip.h")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in SOURCES + [HEADER])


def build(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
               "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-o", LIB_PATH, SOURCES[0]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
