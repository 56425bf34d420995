"""Builds gear_amd/libgear_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgear_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in
           ("gear_hip.hip", "gear_kernels.hpp", "gear_pipeline.hpp", "gear_small.hpp", "gear_edges.hpp", "gear_shard.hpp", "gear_device.hpp", "gear_scene.hpp", "gear_types.hpp")]
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


def build_ablate(defines=("GEAR_HIP_ABLATE",), name="libgear_hip_ablate.so", verbose: bool = False) -> str:
    """The profiling build tools/ablate_*.py and tools/attrib_sweep.sh load through GEAR_HIP_LIB: the same
    library with the ablation bits of `what` compiled in (they return WRONG flags on purpose; the product
    build masks `what` to the declared bits and contains none of that code)."""
    out = os.path.join(_HERE, "..", "build", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize",
           "-Wno-unused-function"] + ["-D" + d for d in defines] + ["-o", out, SOURCES[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


HOST_LIB_PATH = os.path.join(_HERE, "libgear_host.so")
HOST_SOURCES = [os.path.join(_HERE, "host", f) for f in ("gear_host_capi.cpp", "gear_host.hpp")]
HOST_HEADER = os.path.join(_HERE, "..", "include", "gear_host.h")


def host_is_stale() -> bool:
    if not os.path.exists(HOST_LIB_PATH):
        return True
    t = os.path.getmtime(HOST_LIB_PATH)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in HOST_SOURCES + [HOST_HEADER, HEADER])


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The C++ host layer (planner, RRT driver, URDF flattening) -> gear_amd/libgear_host.so, linked
    against libgear_hip.so next to it.  Plain g++: no device code in it."""
    build(False, verbose)
    if force or host_is_stale():
        cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra", "-o", HOST_LIB_PATH, HOST_SOURCES[0],
               "-L" + _HERE, "-lgear_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
