"""Builds manta_amd/libmanta_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmanta_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def sources():
    return [os.path.join(CSRC, "api.cpp")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "manta_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_profile_variant(verbose=True):
    """developer tool: same library with the per-phase shader-clock counters of assemble_kernel compiled in
    (tools/profile_phases.py); never used by tests, bench.py or the product path"""
    out = os.path.join(HERE, "libmanta_amd_prof.so")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-DMANTA_ASM_PROFILE",
           "-I", CSRC, "-o", out] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-I", CSRC, "-o", OUT] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--profile" in sys.argv:
        build_profile_variant()
    else:
        build(force="--force" in sys.argv)
