"""Builds manta_amd/libmanta_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

The library is one translation unit per kernel family (csrc/kernels_tu.cpp compiled with -DMANTA_TU=<id>, csrc/wave.hpp) plus the
host sources (-DMANTA_TU=MANTA_TU_HOST: every kernel only declared), compiled in parallel into manta_amd/build/*.o and linked; a
translation unit is compiled again only when one of the files it includes (its .d file) changed."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libmanta_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", CSRC]

# csrc/wave.hpp: MANTA_TU_*
HOST_TU = 1
KERNEL_TUS = {
    2: "asm", 3: "asm_generic", 4: "graph", 5: "graph_big", 6: "contig", 7: "repeat", 8: "align0", 9: "align1", 10: "align2",
    11: "align_pair", 12: "jump_pair", 13: "glue",
}
HOST_SOURCES = ["api.cpp", "api_batch.cpp", "api_reads.cpp"]


def units():
    """(object path, source path, MANTA_TU id), the slowest to compile first"""
    u = [(os.path.join(OBJ, "k_%s.o" % name), os.path.join(CSRC, "kernels_tu.cpp"), tu) for tu, name in KERNEL_TUS.items()]
    u.sort(key=lambda x: {"asm_generic": 0, "asm": 1, "graph_big": 2, "repeat": 3, "contig": 4, "graph": 5}.get(os.path.basename(x[0])[2:-2], 9))
    u += [(os.path.join(OBJ, "h_%s.o" % os.path.splitext(s)[0]), os.path.join(CSRC, s), HOST_TU) for s in HOST_SOURCES]
    return u


def sources():
    return [os.path.join(CSRC, s) for s in HOST_SOURCES] + [os.path.join(CSRC, "kernels_tu.cpp")]


def _deps(obj):
    """files the object was compiled from (its make-style .d file); None if unknown"""
    d = obj[:-2] + ".d"
    if not os.path.exists(d):
        return None
    text = open(d).read().replace("\\\n", " ")
    return [p for p in text.split(":", 1)[1].split() if p]


def _stale(obj, extra_flags):
    if not os.path.exists(obj):
        return True
    flags_file = obj[:-2] + ".flags"
    if not os.path.exists(flags_file) or open(flags_file).read() != " ".join(extra_flags):
        return True
    deps = _deps(obj)
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    return any((not os.path.exists(p)) or os.path.getmtime(p) > t for p in deps + [os.path.abspath(__file__)])


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(_stale(o, []) or os.path.getmtime(o) > t for o, _, _ in units())


def _compile(unit, extra_flags, verbose):
    obj, src, tu = unit
    cmd = [HIPCC] + FLAGS + extra_flags + ["-DMANTA_TU=%d" % tu, "-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(obj[:-2] + ".flags", "w").write(" ".join(extra_flags))
    return obj


def _jobs():
    if os.environ.get("MANTA_AMD_BUILD_JOBS"):
        return max(1, int(os.environ["MANTA_AMD_BUILD_JOBS"]))
    return max(1, min(8, os.cpu_count() or 1))


def build_profile_variant(verbose=True):
    """developer tool: same library with the per-phase shader-clock counters of assemble_kernel compiled in
    (tools/profile_phases.py); one translation unit; never used by tests, bench.py or the product path"""
    out = os.path.join(HERE, "libmanta_amd_prof.so")
    cmd = [HIPCC] + FLAGS + ["-shared", "-DMANTA_ASM_PROFILE", "-o", out, os.path.join(CSRC, "api_unity.cpp")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=True, extra_flags=(), out=OUT, obj_dir=None):
    """obj_dir: developer variants (extra_flags) keep their objects apart from the product's"""
    extra_flags = list(extra_flags)
    if not force and not extra_flags and out == OUT and not needs_build():
        return OUT
    us = units()
    if obj_dir:
        us = [(os.path.join(obj_dir, os.path.basename(o)), src, tu) for o, src, tu in us]
    os.makedirs(obj_dir or OBJ, exist_ok=True)
    todo = [u for u in us if force or _stale(u[0], extra_flags)]
    with ThreadPoolExecutor(max_workers=_jobs()) as pool:
        list(pool.map(lambda u: _compile(u, extra_flags, verbose), todo))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [o for o, _, _ in us]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--profile" in sys.argv:
        build_profile_variant()
    else:
        build(force="--force" in sys.argv)
