"""The product library is built from one translation unit per kernel family plus the host sources (manta_amd/build.py, csrc/wave.hpp:
MANTA_TU_*).  These checks keep the three places that name the units from drifting apart, and pin the CPU-baseline thread harness."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "manta_amd", "csrc")


def test_build_units_match_the_ids_of_wave_hpp():
    from manta_amd import build as b
    ids = {name: int(v) for name, v in re.findall(r"#define (MANTA_TU_[A-Z0-9_]+) (\d+)", open(os.path.join(CSRC, "wave.hpp")).read())}
    assert ids["MANTA_TU_ALL"] == 0 and ids["MANTA_TU_HOST"] == b.HOST_TU
    kernel_ids = {v for k, v in ids.items() if k not in ("MANTA_TU_ALL", "MANTA_TU_HOST", "MANTA_TU_COUNT")}
    assert kernel_ids == set(b.KERNEL_TUS), (sorted(kernel_ids), sorted(b.KERNEL_TUS))
    assert ids["MANTA_TU_COUNT"] == max(kernel_ids) + 1
    # every kernel family of wave.hpp is handled by kernels_tu.cpp, every host source exists and is part of the emulator's unity file
    tu = open(os.path.join(CSRC, "kernels_tu.cpp")).read()
    for name in ids:
        if name not in ("MANTA_TU_ALL", "MANTA_TU_HOST", "MANTA_TU_COUNT"):
            assert name in tu, name
    unity = open(os.path.join(CSRC, "api_unity.cpp")).read()
    for src in b.HOST_SOURCES:
        assert os.path.exists(os.path.join(CSRC, src)) and '#include "%s"' % src in unity, src


def test_every_kernel_is_guarded_by_a_translation_unit():
    """a __global__ definition outside a MANTA_TU guard would be compiled into every unit (duplicate symbols at link time)"""
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hpp") or f in ("wave.hpp", "rt.hpp", "api_internal.hpp"):
            continue
        lines = open(os.path.join(CSRC, f)).read().split("\n")
        for i, line in enumerate(lines):
            if re.match(r"^WV_KERNEL", line) and not line.rstrip().endswith(";"):
                prev = "\n".join(lines[max(0, i - 4):i])
                is_template = "template <" in prev
                guarded = "#else" in prev or "MANTA_TU_DEFINES" in prev
                assert is_template or guarded, "%s:%d %s" % (f, i + 1, line[:80])


def test_cpu_baseline_harness_counts_and_scales(oracle):
    """oracle/bench_harness.hpp behind orc_bench_small_sv_timed: the requested number of loci is processed, the clock stops at the deadline,
    and two pinned threads are not slower than one (the first version pinned the calling thread and every later run with it)"""
    from oracle_lib import asm_opts
    from synth import config2_batch
    sb = config2_batch(24, seed=5)
    opts = asm_opts(minWordLength=31)
    sc = [2, -8, -24, -1, -1, 0]
    rates = []
    for rep in range(2):  # (the second round would collapse onto one CPU if the harness left the caller pinned)
        for t in (1, 2):
            secs, done = oracle.bench_small_sv_timed(opts, sc, -100, sb[0], sb[1], sb[2], sb[3], sb[4], (100, 100, 800, 800), t, 16 * t, 30.0)
            assert done == 16 * t and secs > 0
            rates.append(done / secs)
    assert len(os.sched_getaffinity(0)) < 2 or rates[3] > 0.8 * rates[2]
    secs, done = oracle.bench_small_sv_timed(opts, sc, -100, sb[0], sb[1], sb[2], sb[3], sb[4], (100, 100, 800, 800), 1, 10 ** 9, 0.3)
    assert 0 < done < 10 ** 9 and secs < 5.0
