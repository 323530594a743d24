"""Pins the CPU restatement (oracle/manta_oracle.cpp) against the UNMODIFIED reference sources compiled into
oracle/_ref (skipped where the reference build is unavailable)."""
import os
import random

import pytest

from oracle_lib import asm_opts
from synth import repeat_rich_pile, small_indel_locus, breakend_locus


def test_libstdcxx_order_emulation(oracle):
    rng = random.Random(1)
    for n in [1, 5, 13, 14, 30, 100, 600, 3000, 12000]:
        keys = list({"".join(rng.choice("ACGT") for _ in range(rng.choice([5, 8, 25, 31, 41, 76]))) for _ in range(n)})
        for k in keys[:20]:
            assert oracle.hash_bytes(k) == oracle.hash_bytes(k, True)
        assert oracle.unordered_order(keys) == oracle.unordered_order(keys, True), n


def test_assembler_repeat_rich(oracle, reflib):
    for seed in range(600):
        reads = repeat_rich_pile(seed)
        rng = random.Random(seed)
        k0 = rng.choice([3, 4, 5, 6, 8, 10])
        o = asm_opts(minWordLength=k0, maxWordLength=k0 + rng.choice([0, 4, 9, 15]), wordStepSize=rng.choice([1, 2, 3, 5]),
                     minCoverage=rng.choice([1, 1, 2]), minSupportReads=rng.choice([1, 2]), minUnusedReads=rng.choice([1, 3]),
                     maxAssemblyCount=rng.choice([2, 10]))
        assert reflib.assemble(o, reads) == oracle.assemble(o, reads), seed


def test_small_sv_locus_config2_shape(oracle, reflib):
    o = asm_opts(minWordLength=31)
    sc = [2, -8, -24, -1, -1, 0]
    for seed in range(12):
        reads, ref = small_indel_locus(seed, tandem=(seed % 3 == 0))
        a = reflib.small_sv_locus(o, sc, -100, reads, ref, (100, 100, 800, 800))
        b = oracle.small_sv_locus(o, sc, -100, reads, ref, (100, 100, 800, 800))
        assert a == b, seed


def test_assembler_config5_shape(oracle, reflib):
    for seed in range(3):
        reads, ref1, ref2 = breakend_locus(seed, tandem_frac=1.0 if seed == 0 else 0.1)
        o = asm_opts(minWordLength=[25, 40, 75][seed], maxWordLength=76)
        assert reflib.assemble(o, reads) == oracle.assemble(o, reads), seed


def _rand_align_case(rng, kind, maxlen):
    def rs(n, al="ACGT"):
        return "".join(rng.choice(al) for _ in range(n))

    def mut(s, rate):
        out = []
        for c in s:
            x = rng.random()
            if x < rate / 3:
                continue
            if x < 2 * rate / 3:
                out.append(rng.choice("ACGTN"))
                out.append(c)
                continue
            if x < rate:
                out.append(rng.choice("ACGTN"))
                continue
            out.append(c)
        return "".join(out)

    ref1 = rs(rng.randint(1, maxlen), rng.choice(["ACGT", "AC", "ACGTN"]))
    if kind == 2:
        ref2 = rs(rng.randint(1, maxlen), rng.choice(["ACGT", "AC"]))
        a, b = rng.randint(0, len(ref1)), rng.randint(0, len(ref2))
        q = ref1[max(0, a - rng.randint(0, maxlen // 2)):a] + rs(rng.choice([0, 0, 0, 1, 3, 8])) + ref2[b:b + rng.randint(0, maxlen // 2)]
        return (mut(q, rng.choice([0, 0.05, 0.2])) or rs(rng.randint(1, 10)), ref1, ref2)
    if rng.random() < 0.3:
        return (rs(rng.randint(1, maxlen // 2 + 1)), ref1, None)
    a = rng.randint(0, len(ref1))
    b = rng.randint(a, len(ref1))
    c = rng.randint(b, len(ref1))
    d = rng.randint(c, len(ref1))
    q = ref1[a:b] + rs(rng.choice([0, 0, 2, 30])) + ref1[c:d]
    return (mut(q, rng.choice([0, 0.05, 0.2])) or rs(rng.randint(1, 10)), ref1, None)


SCORE_SETS = [[2, -8, -24, -1, -1, 0], [2, -4, -5, -1, -1, 0], [2, -4, -10, -1, -1, 0], [2, -4, -2, 0, -1, 0],
              [2, -8, -100, 0, -1, 0], [2, -8, -12, -1, -1, 0], [1, -4, -6, -1, -2, 1], [2, -8, -18, -1, -1, 1]]


def test_aligners_random(oracle, reflib):
    rng = random.Random(7)
    for _ in range(1500):
        kind = rng.choice([0, 1, 2])
        sc = list(rng.choice(SCORE_SETS))
        if kind == 2:
            sc[5] = 0
        extra = rng.choice([-100, -3, -20, -50])
        q, r1, r2 = _rand_align_case(rng, kind, 120)
        assert reflib.align(kind, sc, extra, q, r1, r2) == oracle.align(kind, sc, extra, q, r1, r2)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference sources only exist in the authoring container")
def test_oracle_recipe_builds_from_clean(tmp_path):
    """`make -C oracle clean all` in a scratch copy: the committed recipe alone must produce both _ref libraries"""
    import shutil
    import subprocess
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    dst = tmp_path / "oracle"
    shutil.copytree(src, dst, ignore=shutil.ignore_patterns("_ref", "*.so"))
    shutil.copytree(os.path.join(os.path.dirname(src), "include"), tmp_path / "include")  # (the read-gathering restatement uses the ABI structs)
    subprocess.check_call(["make", "-s", "-C", str(dst), "-j4", "all"])
    for f in ("libmanta_oracle.so", "_ref/libmanta_ref.so", "_ref/libmanta_ref_refiner.so"):
        assert (dst / f).exists(), f
