"""HIP assembler kernel vs the oracle: wave emulator (CPU tier, small piles) and GPU (-m gpu, full shapes)."""
import json
import os
import random

import pytest

from manta_amd._capi import assembly_text
from oracle_lib import asm_opts
from synth import small_indel_locus, breakend_locus, repeat_rich_pile, config5_locus

EMU_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libmanta_amd_emu.so")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ASM = json.load(open(os.path.join(GOLD, "assembler_reference_tests.json")))


@pytest.fixture(autouse=True)
def _general_kernel_unless_lds_test(request, monkeypatch):
    """The LDS pipeline (asm_lds.hpp: graph_kernel -> contig_kernel) is the library's default for piles that fit its envelope.  The
    tests of this module that were written for the general kernel keep exercising the general kernel (MANTA_AMD_ASM_PATH=general);
    the *fast* / *lds* tests run the pipeline (with its punt list into the general kernel)."""
    if not any(t in request.node.name for t in ("fast", "lds", "lane_order")):
        monkeypatch.setenv("MANTA_AMD_ASM_PATH", "general")


def _mid_cases(seeds):
    out = []
    for seed in seeds:
        rng = random.Random(seed)
        nr, rl = rng.choice([6, 12, 25, 70]), rng.choice([30, 50, 80])
        reads, _ = small_indel_locus(seed, n_reads=nr, read_len=rl, ref_len=400, sub_rate=rng.choice([0, 0.01, 0.03]),
                                     n_rate=rng.choice([0, 0.01]))
        k0 = rng.choice([8, 12, 17, 25, 33])
        o = asm_opts(minWordLength=k0, maxWordLength=k0 + rng.choice([0, 10, 20]), wordStepSize=rng.choice([3, 5]),
                     minCoverage=rng.choice([1, 1, 2]), minSupportReads=rng.choice([1, 2]), minUnusedReads=rng.choice([1, 3]),
                     maxAssemblyCount=rng.choice([2, 10]))
        out.append((o, reads))
    return out


def _check(lib, oracle, cases, allow_unsupported=False):
    n_ok = 0
    for o, reads in cases:
        r = lib.assemble_batch(o, [reads], strict=not allow_unsupported)[0]
        if r["status"] != 0:
            assert allow_unsupported, r
            continue
        assert assembly_text(r) == oracle.assemble(o, reads), (o, reads)
        n_ok += 1
    return n_ok


def check_reference_golden(lib):
    """every vector of the reference's own assembler tests (assembly/test/IterativeAssemblerTest.cpp:63-205), the junk read
    "123456789123" of test_BasicAssembler / test_IterativeKmer included: where its bytes cannot be masked exactly the locus runs on
    the byte-generic kernel (AssemblerT<8>) -- nothing is refused"""
    for c in ASM:
        r = lib.assemble_batch(asm_opts(**c["opts"]), [c["reads"]])[0]
        assert assembly_text(r) == c["ref_text"], c["name"]
    assert len(ASM) == 4 and any(set(r) - set("ACGTN") for c in ASM for r in c["reads"])


def check_circle_detector(lib):
    """test_CircleDetector (assembly/test/IterativeAssemblerTest.cpp:30-61): the reference fills wordCount with eight 5-mers by hand
    and asks getRepeatKmers for the repeat words.  Here the same map comes out of a pile whose reads ARE those words (a read of
    length k holds one word; three / two copies give the counts; first occurrences follow the test's insertion order, which is
    what the hash-order-dependent search sees) and the device's repeat-word set is read back through the introspection call."""
    words = [("TACCA", 3), ("CCACC", 3), ("CACCA", 3), ("ACCAC", 3), ("CCACA", 3), ("CACAC", 3), ("ACACA", 3), ("AAAAA", 2)]
    reads = [w for w, _ in words] + [w for w, c in words for _ in range(c - 1)]
    got = lib.debug_repeat_words(asm_opts(minWordLength=5, maxWordLength=5, minCoverage=1), reads)
    for w in ("ACCAC", "CACCA", "CCACC"):  # the first circle
        assert w in got
    assert "TACCA" not in got and "CCACA" not in got
    assert "CACAC" in got and "ACACA" in got  # the second circle
    assert "AAAAA" in got  # homopolymer: self-circle
    assert got == {"ACCAC", "CACCA", "CCACC", "CACAC", "ACACA", "AAAAA"}


def test_emulated_circle_detector(emu):
    check_circle_detector(emu)


@pytest.mark.gpu
def test_gpu_circle_detector(gpu):
    check_circle_detector(gpu)


def test_emulated_assembler_reference_golden(emu):
    """the reference's own assembler unit-test vectors (assembly/test/IterativeAssemblerTest.cpp:30-205), junk read included
    where it provably cannot matter"""
    check_reference_golden(emu)


@pytest.mark.gpu
def test_gpu_assembler_reference_golden(gpu, oracle):
    check_reference_golden(gpu)
    check_junk_piles(gpu, oracle)


def junk_piles():
    """piles with bytes outside {A,C,G,T,N}: maskable ones (fewer junk reads than minCoverage, acyclic graphs) and all the ways
    they are NOT maskable -- a junk word reaching the seed threshold, minCoverage 1, a cyclic graph next to junk, junk inside
    reads that otherwise assemble (the BAM '=' code), junk as first / last symbol, lower case"""
    base = ["ACGTGTATTACC", "GTGTATTACCTA", "ATTACCTAGTAC", "TACCTAGTACTC", "ACGTGTATTACCTAGTACTC"]
    o2 = dict(minWordLength=6, maxWordLength=6, wordStepSize=5, minCoverage=2, minUnusedReads=1, minSupportReads=1)
    o1 = dict(o2, minCoverage=1)
    o3 = dict(minWordLength=3, maxWordLength=9, wordStepSize=3, minCoverage=2, minUnusedReads=1, minSupportReads=1)
    cyc = ["ACACACACGATG", "GATGTCTCTCTC", "ACACACACGATG", "GATGTCTCTCTC"]
    out = []
    for junk in (["123456789123"], ["ACGTGT=TTACCTAG"], ["GTGTATTAC1TAGTAC"]):
        out.append((o2, base + junk))
        out.append((o1, base + junk))
    out.append((o2, base + ["123456789123", "123456789123"]))
    out.append((o2, base + ["ACGTGT=TTACCTAG", "ACGTGT=TTACCTAG", "GT=TTACCTAGTAC"]))
    out.append((o3, cyc + ["123456789123"]))
    out.append((dict(o3, minCoverage=1), cyc + ["ACAC=CACGATG", "GATGTCT*TCTC"]))
    out.append((o1, ["=CGTGTATTACC", "GTGTATTACCT=", "acgtgtattacc", "ACGTGTATTACC", "GTGTATTACCTA"]))
    out.append((dict(o1, minWordLength=4, maxWordLength=12, wordStepSize=4), base + ["ACGTGT=TTACCTAGTACTC", "NNNN=NNNN", "=", "==", "A=C=G=T=A=C=G=T"]))
    return out


def check_junk_piles(lib, checker):
    for o, reads in junk_piles():
        r = lib.assemble_batch(asm_opts(**o), [reads])[0]
        assert r["status"] == 0 and assembly_text(r) == checker.assemble(asm_opts(**o), reads), (o, reads)
    # one batch: ordinary loci next to byte-generic ones
    o = junk_piles()[0][0]
    piles = [p for oo, p in junk_piles() if oo == o] + [small_indel_locus(5, n_reads=12, read_len=40, ref_len=200)[0]]
    for reads, r in zip(piles, lib.assemble_batch(asm_opts(**o), piles)):
        assert assembly_text(r) == checker.assemble(asm_opts(**o), reads)


def test_junk_bytes_against_the_reference(emu, reflib):
    """the reference is byte-generic (any byte is a symbol; only 'N' words are skipped, only A,C,G,T extend a contig): so is the
    device path -- masked where that is provably exact, the byte-generic kernel everywhere else"""
    check_junk_piles(emu, reflib)


def random_junk_pile(seed):
    """a random pile (small-indel or repeat-rich) with a few bytes outside {A,C,G,T,N} sprinkled in, a possible all-digit read, a
    possible duplicated read, and a random option block; every fourth seed with longer reads and word lengths of 33 .. 111 (keys of
    up to 32 dwords in the byte-generic kernel)"""
    rng = random.Random(1000 + seed)
    long_words = (seed % 4 == 3)
    if long_words:
        reads = small_indel_locus(seed, n_reads=rng.choice([8, 16, 30]), read_len=rng.choice([90, 150]), ref_len=400,
                                  sub_rate=rng.choice([0, 0.01]), n_rate=rng.choice([0, 0.005]))[0]
    elif rng.random() < 0.5:
        reads = small_indel_locus(seed, n_reads=rng.choice([6, 12, 25]), read_len=rng.choice([30, 50]), ref_len=300,
                                  sub_rate=rng.choice([0, 0.02]), n_rate=rng.choice([0, 0.01]))[0]
    else:
        reads = repeat_rich_pile(seed)
    reads = [list(r.decode() if isinstance(r, bytes) else r) for r in reads]
    for _ in range(rng.choice([1, 1, 2, 4])):
        i = rng.randrange(len(reads))
        if reads[i]:
            reads[i][rng.randrange(len(reads[i]))] = rng.choice("=*1acgt.-")
    if rng.random() < 0.3:
        reads.append(list("".join(rng.choice("0123456789") for _ in range(rng.randint(5, 20)))))
    if rng.random() < 0.3:
        reads.append(list(reads[rng.randrange(len(reads))]))
    k0 = rng.choice([33, 41, 47, 60, 76]) if long_words else rng.choice([4, 6, 8, 12, 17, 25, 31])
    kmax = min(128, k0 + rng.choice([0, 10, 35])) if long_words else min(32, k0 + rng.choice([0, 4, 9]))
    o = asm_opts(minWordLength=k0, maxWordLength=kmax, wordStepSize=rng.choice([5, 7]) if long_words else rng.choice([1, 2, 3, 5]),
                 minCoverage=rng.choice([1, 1, 2]), minSupportReads=rng.choice([1, 2]), minUnusedReads=rng.choice([1, 3]),
                 maxAssemblyCount=rng.choice([2, 10]))
    return o, ["".join(r) for r in reads]


def test_emulated_random_junk_piles_against_the_reference(emu, reflib):
    """(the same generator ran 300 seeds against the unmodified reference when the byte-generic kernel was written: 0 mismatches)"""
    for seed in range(40):
        o, reads = random_junk_pile(seed)
        r = emu.assemble_batch(o, [reads])[0]
        assert assembly_text(r) == reflib.assemble(o, reads), (seed, o, reads)


@pytest.mark.gpu
def test_gpu_random_junk_piles(gpu, oracle):
    cases = [random_junk_pile(seed) for seed in range(200)]
    for seed, (o, reads) in enumerate(cases):
        r = gpu.assemble_batch(o, [reads])[0]
        assert assembly_text(r) == oracle.assemble(o, reads), (seed, o)


LONG_WORD_JUNK = (["ACGTACGTTGCATGCAAGGCTTAACCGGTTACGATCGATCGGATCGATTAGC=ATCGGCTA"] * 2 +
                  ["ACGTACGTTGCATGCAAGGCTTAACCGGTTACGATCGATCGGATCGATTAGCGATCGGCTA"])


def test_junk_with_the_default_word_lengths_equals_the_reference(emu, reflib):
    """a non-maskable junk byte at Manta's default word lengths (41 .. 76): was the one combination outside the envelope (status -5)
    while the byte-generic kernel's keys ended at 32 symbols; they hold 128 now, as the 2-bit kernel's do"""
    for o in (asm_opts(minWordLength=41, maxWordLength=41, minCoverage=1), asm_opts(minCoverage=1), asm_opts(minWordLength=33, maxWordLength=128, wordStepSize=19)):
        r = emu.assemble_batch(o, [LONG_WORD_JUNK])[0]
        assert r["status"] == 0 and assembly_text(r) == reflib.assemble(o, LONG_WORD_JUNK)


def test_restatement_matches_the_reference_on_junk_piles(oracle, reflib):
    for o, reads in junk_piles():
        assert oracle.assemble(asm_opts(**o), reads) == reflib.assemble(asm_opts(**o), reads), (o, reads)
    for seed in range(200):  # (what test_gpu_random_junk_piles checks the device against)
        o, reads = random_junk_pile(seed)
        assert oracle.assemble(o, reads) == reflib.assemble(o, reads), (seed, o)


def test_emulated_assembler_mid(emu, oracle):
    assert _check(emu, oracle, _mid_cases(range(40))) == 40


def test_emulated_assembler_repeat_rich(emu, oracle):
    cases = []
    for seed in range(60):
        rng = random.Random(seed)
        k0 = rng.choice([4, 5, 6, 8, 10])
        o = asm_opts(minWordLength=k0, maxWordLength=k0 + rng.choice([0, 4, 9, 15]), wordStepSize=rng.choice([1, 2, 3, 5]),
                     minCoverage=rng.choice([1, 1, 2]), minSupportReads=rng.choice([1, 2]), minUnusedReads=rng.choice([1, 3]),
                     maxAssemblyCount=rng.choice([2, 10]))
        cases.append((o, repeat_rich_pile(seed)))
    assert _check(emu, oracle, cases) == 60


def test_emulated_assembler_config2_locus(emu, oracle):
    reads, _ = small_indel_locus(3)
    assert _check(emu, oracle, [(asm_opts(minWordLength=31), reads)]) == 1


def test_emulated_batch_of_ragged_loci(emu, oracle):
    """one launch, loci of very different sizes incl. an empty pile and reads shorter than k"""
    loci = [small_indel_locus(1, n_reads=10, read_len=40, ref_len=300)[0], [], [b"ACGT", b"AC"],
            small_indel_locus(2, n_reads=30, read_len=60, ref_len=300)[0]]
    o = asm_opts(minWordLength=15, maxWordLength=30)
    res = emu.assemble_batch(o, loci)
    for reads, r in zip(loci, res):
        assert assembly_text(r) == oracle.assemble(o, reads)


@pytest.mark.gpu
def test_gpu_assembler_mid_and_repeat_rich(gpu, oracle):
    cases = _mid_cases(range(200))
    for seed in range(400):
        rng = random.Random(seed)
        k0 = rng.choice([4, 5, 6, 8, 10])
        o = asm_opts(minWordLength=k0, maxWordLength=k0 + rng.choice([0, 4, 9, 15]), wordStepSize=rng.choice([1, 2, 3, 5]),
                     minCoverage=rng.choice([1, 1, 2]), minSupportReads=rng.choice([1, 2]), minUnusedReads=rng.choice([1, 3]),
                     maxAssemblyCount=rng.choice([2, 10]))
        cases.append((o, repeat_rich_pile(seed)))
    assert _check(gpu, oracle, cases) == len(cases)


@pytest.mark.gpu
def test_gpu_assembler_config2_batch(gpu, oracle):
    o = asm_opts(minWordLength=31)
    loci = [small_indel_locus(s, tandem=(s % 4 == 0))[0] for s in range(96)]
    res = gpu.assemble_batch(o, loci)
    for reads, r in zip(loci, res):
        assert assembly_text(r) == oracle.assemble(o, reads)


@pytest.mark.gpu
def test_gpu_assembler_config5_batch(gpu, oracle):
    loci, opts = [], None
    for k in (25, 50, 75):
        o = asm_opts(minWordLength=k, maxWordLength=max(76, k))
        loci = [breakend_locus(1000 * k + s, tandem_frac=0.3)[0] for s in range(12)]
        res = gpu.assemble_batch(o, loci)
        for reads, r in zip(loci, res):
            assert assembly_text(r) == oracle.assemble(o, reads), k


def test_emulated_assembler_many_reads_uses_wide_sets(emu, oracle):
    """> 256 reads: read sets no longer fit lane-private registers -> contigs are built one at a time (wide-set path)"""
    reads, _ = small_indel_locus(11, n_reads=300, read_len=40, ref_len=300, sub_rate=0.01)
    assert _check(emu, oracle, [(asm_opts(minWordLength=15, maxWordLength=25), reads)]) == 1


def test_emulated_serial_and_speculative_walks_agree(emu, oracle, monkeypatch):
    cases = _mid_cases(range(100, 112))
    assert _check(emu, oracle, cases) == len(cases)
    monkeypatch.setenv("MANTA_AMD_SERIAL_WALK", "1")
    assert _check(emu, oracle, cases) == len(cases)


# ---------------------------------------------------------------------------------------------------------------
# The LDS pipeline (graph_kernel -> contig_kernel, asm_lds.hpp / asm_contig.hpp; the default, MANTA_AMD_ASM_PATH=general switches it
# off): same results as the oracle, and everything it does not cover (cycles, next word length, wide read sets; N-masked piles are
# fine) goes to the general kernel through the device-side punt list
# ---------------------------------------------------------------------------------------------------------------
def _lds_cases():
    cases = [(asm_opts(minWordLength=31), small_indel_locus(s)[0]) for s in range(3)]
    cases += [(asm_opts(minWordLength=31), small_indel_locus(50 + s, n_rate=0.01)[0]) for s in range(2)]   # 'N' windows are skipped
    cases += [(asm_opts(minWordLength=25, maxWordLength=45), small_indel_locus(60 + s, tandem=True)[0]) for s in range(3)]  # cycles -> fallback
    cases += _mid_cases(range(30))
    for seed in range(30):
        rng = random.Random(seed)
        k0 = rng.choice([4, 5, 6, 8, 10])
        cases.append((asm_opts(minWordLength=k0, maxWordLength=k0 + rng.choice([0, 4, 9, 15]), wordStepSize=rng.choice([1, 2, 3, 5]),
                               minCoverage=rng.choice([1, 1, 2]), minSupportReads=rng.choice([1, 2]), minUnusedReads=rng.choice([1, 3]),
                               maxAssemblyCount=rng.choice([2, 10])), repeat_rich_pile(seed)))
    cases.append((asm_opts(minWordLength=15, maxWordLength=25), small_indel_locus(11, n_reads=140, read_len=40, ref_len=300, sub_rate=0.01)[0]))  # W > 2
    cases.append((asm_opts(minWordLength=15), []))
    cases.append((asm_opts(minWordLength=15), [b"ACGT", b"AC"]))
    return cases


def test_emulated_fast_kernel_matches_oracle(emu, oracle, monkeypatch):
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    cases = _lds_cases()
    assert _check(emu, oracle, cases) == len(cases)


@pytest.mark.parametrize("classes", ["4096,8192,16384,54272", "12288", "54272"])
def test_emulated_fast_kernel_lds_size_classes(emu, oracle, monkeypatch, classes):
    """contig_kernel is launched once per LDS size class (graph_kernel sorts the loci into the smallest class their compact graph
    fits); a graph that fits no class goes to the general kernel.  Whatever the classes are, the results are the same."""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    monkeypatch.setenv("MANTA_AMD_LG_CLASSES", classes)
    cases = _lds_cases()[:24]
    assert _check(emu, oracle, cases) == len(cases)


def test_emulated_kernels_do_not_depend_on_lane_order(emu, oracle):
    """The emulator steps the lanes of a wave 0..63 between two rendezvous; MANTA_EMU_LANE_ORDER=reverse steps them 63..0.  A
    kernel that gives different results that way reads what another lane writes without a wv::sync() in between -- which the
    hardware (lock step) would resolve in yet another way.  Both assembler kernels, a sample of the cases, in a fresh process
    (the order is read once)."""
    import subprocess
    import sys
    code = ("import os, sys\n"
            "sys.path.insert(0, %r)\n"
            "import conftest, test_assemble_kernels as t\n"
            "from oracle_lib import OracleLib\n"
            "from manta_amd._capi import Lib\n"
            "lib, oracle = Lib(path=t.EMU_PATH), OracleLib()\n"
            "cases = t._lds_cases()\n"
            "cases = cases[:6] + cases[8:14] + cases[38:46]\n"
            "for path in ('fast', 'general'):\n"
            "    os.environ['MANTA_AMD_ASM_PATH'] = path\n"
            "    assert t._check(lib, oracle, cases) == len(cases)\n"
            "print('lane-order-ok')\n") % os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MANTA_EMU_LANE_ORDER="reverse")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert "lane-order-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_emulated_fast_kernel_batch_with_mixed_loci(emu, oracle, monkeypatch):
    """one launch: fast-path loci, fallback loci (tandem repeats, > 108 reads) and an empty pile side by side"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    o = asm_opts(minWordLength=21, maxWordLength=41)
    loci = [small_indel_locus(1, n_reads=30, read_len=60, ref_len=300)[0], [],
            small_indel_locus(2, n_reads=30, read_len=60, ref_len=300, tandem=True)[0],
            small_indel_locus(3, n_reads=150, read_len=50, ref_len=300)[0],
            small_indel_locus(4, n_reads=40, read_len=70, ref_len=300, n_rate=0.02)[0]]
    for reads, r in zip(loci, emu.assemble_batch(o, loci)):
        assert assembly_text(r) == oracle.assemble(o, reads)


@pytest.mark.gpu
def test_gpu_fast_kernel_matches_oracle(gpu, oracle, monkeypatch):
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    cases = _lds_cases() + [(asm_opts(minWordLength=31), small_indel_locus(100 + s)[0]) for s in range(64)]
    assert _check(gpu, oracle, cases) == len(cases)
    o = asm_opts(minWordLength=31)
    loci = [small_indel_locus(s, tandem=(s % 4 == 0), n_rate=(0.01 if s % 5 == 0 else 0.0))[0] for s in range(200)]
    for reads, r in zip(loci, gpu.assemble_batch(o, loci)):
        assert assembly_text(r) == oracle.assemble(o, reads)


def _fast_stats(emu):
    import ctypes
    st = (ctypes.c_ulonglong * 8)()
    emu.lib.manta_emu_fast_stats(st)
    return dict(loci=st[0], rounds=st[1], walks=st[2], cands=st[3], evictions=st[4], reclaimed=st[5], proofs=st[6], sent_on=st[7])


def test_emulated_fast_kernel_speculation_hits(emu, oracle, monkeypatch):
    """the contig kernel's loop runs on speculation (asm_contig.hpp): config-2 loci must come out of (close to) ONE walk round
    each -- a regression guard for the seed prediction, which can only cost time, never results (those are checked as well)"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    o = asm_opts(minWordLength=31)
    loci = [small_indel_locus(s)[0] for s in range(16)]
    _fast_stats(emu)
    res = emu.assemble_batch(o, loci)
    st = _fast_stats(emu)
    for reads, r in zip(loci, res):
        assert assembly_text(r) == oracle.assemble(o, reads)
    assert st["loci"] == 16 and st["cands"] == 16 * 20, st
    assert st["rounds"] <= 16 * 1.5, st
    # reads of one haplotype that differ by substitutions only: graph_kernel proves the graph acyclic from the reads' offsets
    # (a potential that rises along every edge) and contig_kernel skips its peel.  Reads inserted at the same time may end up in
    # separate anchor trees; readOffsets ties the trees together through a shared word (16 of 16 here; without that pass 12-14)
    assert st["proofs"] >= 15, st


def test_emulated_fast_kernel_without_the_acyclicity_proof(emu, oracle, monkeypatch):
    """MANTA_AMD_LG_NO_PROOF: contig_kernel runs its own cycle test (the two-sided peel) on every locus -- same results"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    monkeypatch.setenv("MANTA_AMD_LG_NO_PROOF", "1")
    cases = _lds_cases()[:16]
    _fast_stats(emu)
    assert _check(emu, oracle, cases) == len(cases)
    assert _fast_stats(emu)["proofs"] == 0


# ---- the LDS pipeline's big class (graph_big_kernel -> contig_big_kernel, asm_lds_big.hpp): piles of up to 256 reads ----
def _big_cases():
    """config-5 shaped loci (200 reads x 250 bases, 1 % N, per-locus word length 25..75) and piles around the class' edges"""
    cases = []
    for i in (1, 2, 4, 7, 9):  # k = 55, 25, 65, 40, 75
        reads, _, _, k, kmax = config5_locus(i)
        cases.append((asm_opts(minWordLength=k, maxWordLength=kmax, minContigLength=75), reads))
    reads, _, _, k, kmax = config5_locus(37)  # a tandem-repeat locus: cyclic graph at its one word length (repeat_big_kernel, rounds below)
    cases.append((asm_opts(minWordLength=k, maxWordLength=kmax, minContigLength=75), reads))
    # 129 reads (one more than the small class takes), 236 reads (the most this class takes with maxAssemblyCount 10), 237 (one too many)
    cases.append((asm_opts(minWordLength=31), small_indel_locus(21, n_reads=129, read_len=100, ref_len=600)[0]))
    cases.append((asm_opts(minWordLength=41), small_indel_locus(22, n_reads=236, read_len=150, ref_len=900, sub_rate=0.004, n_rate=0.005)[0]))
    cases.append((asm_opts(minWordLength=41), small_indel_locus(23, n_reads=237, read_len=120, ref_len=700)[0]))
    cases.append((asm_opts(minWordLength=21, minCoverage=2), small_indel_locus(24, n_reads=150, read_len=120, ref_len=700, sub_rate=0.01)[0]))
    cases.append((asm_opts(minWordLength=15, maxAssemblyCount=3), small_indel_locus(25, n_reads=180, read_len=60, ref_len=400, sub_rate=0.02)[0]))
    return cases


def test_emulated_big_class_matches_oracle(emu, oracle, monkeypatch):
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    cases = _big_cases()
    _fast_stats(emu)
    assert _check(emu, oracle, cases) == len(cases)
    # the class took them: every locus but the 237-read pile (and at most one more) went through contig_big_kernel
    assert _fast_stats(emu)["loci"] >= len(cases) - 2


# ---- ... and its word-length rounds: cyclic graphs, repeat hits and pseudo reads on the pipeline (IterativeAssembler.cpp:555-642, 856-910) ----
def _round_cases(n_sweep=12):
    """piles of 129..236 reads with tandem repeats: config-5 loci whose graph is cyclic at one / two / four word lengths, and random
    repeat-rich piles under random options (minCoverage up to 3: a pseudo read then counts more than a read; word steps down to 1;
    maxAssemblyCount 2..10; word lengths from 8) -- every one needs the repeat search in the reference's order and, most, pseudo reads"""
    import numpy as np
    cases = []
    for i in (37, 0, 64):
        reads, _, _, k, kmax = config5_locus(i)
        cases.append((asm_opts(minWordLength=k, maxWordLength=kmax, minContigLength=75), reads))
    for s in range(1, 1 + n_sweep):
        rng = np.random.default_rng(9000 + s)
        nr = int(rng.integers(129, 237))
        if s % 3 == 0:
            reads = repeat_rich_pile(s, n_reads=nr, read_len=int(rng.integers(40, 90)))
        elif s % 3 == 1:
            reads = small_indel_locus(s, n_reads=nr, read_len=int(rng.integers(60, 120)), ref_len=500, sub_rate=0.01, n_rate=0.005, tandem=True)[0]
        else:
            reads = breakend_locus(s, n_reads=min(nr, 200), read_len=int(rng.integers(80, 160)), ref_len=600, tandem_frac=1.0)[0]
        mac = int(rng.integers(2, 11))
        k0 = int(rng.integers(8, 33))
        o = asm_opts(minWordLength=k0, maxWordLength=k0 + int(rng.integers(0, 40)), wordStepSize=int(rng.integers(1, 8)), minCoverage=int(rng.integers(1, 4)),
                     minConservativeCoverage=int(rng.integers(1, 4)), maxAssemblyCount=mac, minContigLength=15,
                     minUnusedReads=int(rng.integers(1, 5)), minSupportReads=int(rng.integers(1, 4)))
        if len(reads) + 2 * mac <= 256:
            cases.append((o, reads))
    return cases


def test_emulated_big_class_word_length_rounds(emu, oracle, monkeypatch):
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    cases = _round_cases()
    _fast_stats(emu)
    assert _check(emu, oracle, cases) == len(cases)
    st = _fast_stats(emu)
    # they stayed on the pipeline: nearly every locus finished in contig_big_kernel, most after being sent on to a further word length
    assert st["loci"] >= len(cases) - 2 and st["sent_on"] >= 2 * len(cases), st


def test_emulated_big_class_rounds_off_hands_repeats_back(emu, oracle, monkeypatch):
    """MANTA_AMD_BIG_ROUNDS=0 (A/B runs): one word length on the pipeline, a cyclic graph or a repeat hit goes to assemble_kernel"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    monkeypatch.setenv("MANTA_AMD_BIG_ROUNDS", "0")
    cases = _round_cases(3)
    _fast_stats(emu)
    assert _check(emu, oracle, cases) == len(cases)
    assert _fast_stats(emu)["sent_on"] == 0


def test_emulated_big_class_in_one_launch_with_the_small_class(emu, oracle, monkeypatch):
    """one batch: small-class loci, big-class loci, an empty pile and a pile outside both classes (general kernel) side by side"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    loci = [small_indel_locus(1, n_reads=30, read_len=60, ref_len=300)[0], config5_locus(5)[0], [],
            small_indel_locus(3, n_reads=300, read_len=50, ref_len=300)[0], config5_locus(6)[0],
            small_indel_locus(4, n_reads=40, read_len=70, ref_len=300, n_rate=0.02)[0]]
    o = asm_opts(minWordLength=31, maxWordLength=51, minContigLength=75)
    for reads, r in zip(loci, emu.assemble_batch(o, loci)):
        assert assembly_text(r) == oracle.assemble(o, reads)


@pytest.mark.gpu
def test_gpu_big_class_matches_oracle(gpu, oracle, monkeypatch):
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    cases = _big_cases()
    for i in range(20, 52):
        reads, _, _, k, kmax = config5_locus(i)
        cases.append((asm_opts(minWordLength=k, maxWordLength=kmax, minContigLength=75), reads))
    assert _check(gpu, oracle, cases) == len(cases)


@pytest.mark.gpu
def test_gpu_big_class_word_length_rounds(gpu, oracle, monkeypatch):
    """the rounds on hardware: tandem-repeat loci of the config-5 generator (up to eleven word lengths, every graph cyclic) and the random
    repeat-rich piles under random options"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    cases = _round_cases(40)
    for i in (3, 14, 25, 35, 72, 77, 114, 129, 260):
        reads, _, _, k, kmax = config5_locus(i)
        cases.append((asm_opts(minWordLength=k, maxWordLength=kmax, minContigLength=75), reads))
    assert _check(gpu, oracle, cases) == len(cases)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_gpu_rounds_sweep_first_500_seeds(gpu, oracle, monkeypatch):
    """the first 500 seeds of the round-5 hardware sweep of the word-length rounds (tools/sweeps/sweep_rounds.py, profiles/r05_rounds_sweep.txt:
    8 000 piles by hand): random repeat-rich piles of 129..236 reads, every pile under its own random option block (word lengths 8..72,
    steps 1..7, minCoverage 1..3, maxAssemblyCount 2..10), each against the CPU restatement; every fifth one also against the unmodified
    reference where oracle/_ref is present"""
    from oracle_lib import RefLib, have_ref
    from rounds_cases import rounds_case
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    ref = RefLib() if have_ref() else None
    bad, n, iters = [], 0, 0
    for s in range(1000, 1500):
        c = rounds_case(s)
        if not c:
            continue
        o, reads = c
        r = gpu.assemble_batch(o, [reads])[0]
        got = assembly_text(r) if r["status"] == 0 else "STATUS %d" % r["status"]
        n += 1
        iters += r["n_iterations"]
        if got != oracle.assemble(o, reads) or (ref is not None and s % 5 == 0 and got != ref.assemble(o, reads)):
            bad.append(s)
    assert not bad, "%d of %d piles differ: seeds %s" % (len(bad), n, bad[:10])
    assert n > 450 and iters > 2000  # the piles do go through several word lengths each
