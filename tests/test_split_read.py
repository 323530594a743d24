"""Split-read scoring: manta_split_read_batch / manta_amd/host/split_read.hpp against
  * the UNMODIFIED reference splitReadAligner (oracle/_ref, ref_scoring_driver.cpp)      -- authoring container only
  * the CPU restatement (oracle/scoring_oracle.cpp)                                      -- everywhere
Floats are compared as C99 hexfloat text, i.e. bit for bit (the float policy in include/manta_amd.h)."""
import ctypes
import json
import os
import random
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_refiner.so")
ORC_SO = os.path.join(ROOT, "oracle", "libmanta_oracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "split_read_cases.json")
SNP_PRIOR = 1e-3  # CallOptionsShared.hpp: snpPrior


class SplitLib:
    def __init__(self, path, prefix):
        self.lib = ctypes.CDLL(path)
        self.one = getattr(self.lib, prefix + "split_read_aligner")
        self.tables = getattr(self.lib, prefix + "qscore_snp_tables")
        self.batch = getattr(self.lib, prefix + "split_read_aligner_batch", None)

    def run(self, c):
        q, t = c["query"].encode(), c["target"].encode()
        qual = (ctypes.c_uint8 * len(q))(*c["qual"])
        buf = ctypes.create_string_buffer(1024)
        self.one(ctypes.c_uint(c["flank"]), q, ctypes.c_uint(len(q)), qual, t, ctypes.c_uint(len(t)), ctypes.c_int(c["bp"][0]),
                 ctypes.c_int(c["bp"][1]), ctypes.c_double(SNP_PRIOR), buf, 1024)
        return buf.value.decode()

    def run_batch(self, cases):
        n = len(cases)
        qs = [c["query"].encode() for c in cases]
        ts = [c["target"].encode() for c in cases]
        quals = [(ctypes.c_uint8 * len(q))(*c["qual"]) for q, c in zip(qs, cases)]
        buf = ctypes.create_string_buffer(512 * n + 64)
        self.batch(ctypes.c_uint(n), (ctypes.c_uint * n)(*[c["flank"] for c in cases]), (ctypes.c_char_p * n)(*qs),
                   (ctypes.c_uint * n)(*[len(q) for q in qs]),
                   (ctypes.POINTER(ctypes.c_uint8) * n)(*[ctypes.cast(x, ctypes.POINTER(ctypes.c_uint8)) for x in quals]),
                   (ctypes.c_char_p * n)(*ts), (ctypes.c_uint * n)(*[len(t) for t in ts]), (ctypes.c_int * n)(*[c["bp"][0] for c in cases]),
                   (ctypes.c_int * n)(*[c["bp"][1] for c in cases]), ctypes.c_double(SNP_PRIOR), buf, len(buf))
        return buf.value.decode().splitlines(keepends=True)

    def get_tables(self):
        a, b = (ctypes.c_double * 71)(), (ctypes.c_double * 71)()
        x, y = ctypes.c_float(), ctypes.c_float()
        n = self.tables(ctypes.c_double(SNP_PRIOR), a, b, ctypes.byref(x), ctypes.byref(y))
        return n, bytes(a), bytes(b), x.value, y.value


def build_mine(lib_dir, lib_name, tag):
    so = os.path.join(CPP, "libhost_scoring_%s.so" % tag)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "manta_amd", "host"), os.path.join(CPP, "host_scoring_capi.cpp"), "-o", so,
                           "-L" + lib_dir, "-l" + lib_name, "-Wl,-rpath," + lib_dir])
    return SplitLib(so, "mine_")


@pytest.fixture(scope="module")
def mine_emu(emu):
    return build_mine(os.path.join(ROOT, "tests", "emu"), "manta_amd_emu", "emu")


@pytest.fixture(scope="module")
def mine_gpu(gpu):
    return build_mine(os.path.join(ROOT, "manta_amd"), "manta_amd", "gpu")


@pytest.fixture(scope="module")
def orc(oracle):
    return SplitLib(ORC_SO, "orc_")


@pytest.fixture(scope="module")
def ref():
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libmanta_ref_refiner.so not built (reference sources unavailable)")
    return SplitLib(REF_SO, "ref_")


TARGET = ("GATCACAGGTCTATCACCCTATTAACCACTCACGGGAGCTCTCCATGCATTTGGT"
          "ATTTTCGTCTGGGGGGTGTGCACGCGATAGCATTGCGAGACGCTGGA")  # the target of SplitReadAlignmentTest.cpp:124-126, 198-200


def reference_test_cases():
    """the inputs of the reference's own unit tests run through the whole function (SplitReadAlignmentTest.cpp:120-330: the five
    query reads of test_calculateAlignScore / test_getLnLhood incl. the 'N' read, placed by the scan instead of by hand)"""
    reads = ["TCTATCACCCATCGTACCACTCACGGGAGCTCTCC", "TCTATGTTCCTATTAACCACTCACGGGAGCTCTCC", "TCTATCACCCTATTAACCACTCACGGGATGTGACC",
             "TCTGTTACCCATCGTACCACTCACGGGAGTTCTCC", "TCTATCACCCTATTAACCACTCACGGGAGCTCTCC", "TCTATCACCCATCGTNCCACTCACGGGAGCTCTCC"]
    out = []
    for r in reads:
        for bp in ((18, 24), (19, 24), (19, 23), (8, 50)):
            out.append(dict(query=r, qual=[30] * len(r), target=TARGET, bp=list(bp), flank=50))
    return out


def random_cases(seed, n):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        tl = rng.randint(60, 420)
        target = "".join(rng.choice("ACGT") for _ in range(tl))
        if rng.random() < 0.2:
            target = "".join(c if rng.random() > 0.02 else "N" for c in target)
        ql = rng.randint(20, min(150, tl - 1))
        start = rng.randint(0, tl - ql)
        q = list(target[start:start + ql])
        for i in range(ql):
            x = rng.random()
            if x < 0.04:
                q[i] = rng.choice("ACGT")
            elif x < 0.05:
                q[i] = "N"
        if rng.random() < 0.15:  # low-complexity stretch: many equal placements (first-best-wins matters)
            p = rng.randint(0, ql - 10)
            q[p:p + 10] = list(rng.choice("ACGT") * 10)
        b = rng.randint(max(0, start - 5), min(tl - 1, start + ql + 5))
        e = b + rng.choice([0, 0, 1, 2, 5, 9])
        qual = [rng.choice([2, 11, 25, 30, 37, 40, 41, 0, 70]) for _ in range(ql)]
        out.append(dict(query="".join(q), qual=qual, target=target, bp=[b, e], flank=rng.choice([50, 50, 16, 0, 200])))
    # the reference's exceptions: query not shorter than the target, empty scan range
    out.append(dict(query="ACGTACGTAC", qual=[30] * 10, target="ACGTACGT", bp=[3, 4], flank=50))
    out.append(dict(query="ACGTACGTAC", qual=[30] * 10, target="ACGTACGTACGTACGTACGT", bp=[30, 2], flank=50))
    return out


def unusual_cases():
    """inputs Manta does not produce but the function accepts: a quality above the table's range (the reference's lookup throws --
    but there is no lookup at an 'N' position), and ranges whose end lies before the best placement (the reference's std::min
    runs on unsigned operands there)"""
    t = "GATCACAGGTCTATCACCCTATTAACCACTCACGGGAGCTCTCCATGCATTTGGTATTTTCGTCTGGGGGGTGTGCACGCGATAGC"
    q = t[20:50]
    qn = q[:7] + "N" + q[8:]
    out = [dict(query=qn, qual=[30] * 7 + [90] + [30] * 22, target=t, bp=[30, 34], flank=50),    # over-range quality at the query's N
           dict(query=q, qual=[30] * 7 + [90] + [30] * 22, target=t[:27] + "N" + t[28:], bp=[30, 34], flank=50),  # ... at a target N
           dict(query=q, qual=[30] * 7 + [90] + [30] * 22, target=t, bp=[30, 34], flank=50),     # ... at a scored base: exception
           dict(query=q, qual=[90] + [30] * 29, target=t, bp=[30, 34], flank=2)]                 # ... outside the scored range
    for b, e in ((30, 25), (40, 12), (25, 24), (60, 0)):
        out.append(dict(query=q, qual=[30] * 30, target=t, bp=[b, e], flank=50))
    return out


def test_unusual_inputs_match_the_reference(ref, orc):
    for c in unusual_cases():
        assert orc.run(c) == ref.run(c), c


def test_emulated_split_read_scorer_unusual_inputs(mine_emu, orc):
    cases = unusual_cases()
    for c in cases:
        assert mine_emu.run(c) == orc.run(c), c
    assert mine_emu.run_batch(cases) == [orc.run(c) for c in cases]


def test_tables_and_restatement_match_the_reference(ref, orc):
    assert ref.get_tables() == orc.get_tables()
    for c in reference_test_cases() + random_cases(1, 400):
        assert orc.run(c) == ref.run(c), c


def test_golden_cases_pin_the_restatement(orc):
    g = json.load(open(GOLDEN))
    for c, want in zip(g["cases"], g["ref_texts"]):
        assert orc.run(c) == want


def test_emulated_split_read_scorer(mine_emu, orc):
    assert mine_emu.get_tables() == orc.get_tables()
    cases = reference_test_cases() + random_cases(2, 120)
    for c in cases[:40]:
        assert mine_emu.run(c) == orc.run(c), c
    assert mine_emu.run_batch(cases) == [orc.run(c) for c in cases]


@pytest.mark.gpu
def test_gpu_split_read_scorer(mine_gpu, orc):
    g = json.load(open(GOLDEN))
    assert mine_gpu.run_batch(g["cases"]) == g["ref_texts"]  # the reference's own outputs
    cases = random_cases(3, 3000) + unusual_cases()
    assert mine_gpu.run_batch(cases) == [orc.run(c) for c in cases]
    # config-like shape: 150-base reads against 500-base contigs, many at once
    rng = np.random.default_rng(5)
    big = []
    for _ in range(2000):
        t = "".join("ACGT"[i] for i in rng.integers(0, 4, size=500))
        s = int(rng.integers(0, 350))
        q = list(t[s:s + 150])
        for i in np.nonzero(rng.random(150) < 0.01)[0]:
            q[int(i)] = "ACGT"[int(rng.integers(0, 4))]
        big.append(dict(query="".join(q), qual=[int(x) for x in rng.integers(2, 42, size=150)], target=t, bp=[s + 70, s + 72], flank=50))
    assert mine_gpu.run_batch(big) == [orc.run(c) for c in big]
