"""The supported envelope at its edges: maximum word length, read sets of several hundred reads (the wide-set walk),
and what lies just beyond -- reported per item (MANTA_E_UNSUPPORTED, -5), never a crash and never a guess.
(The file sorts last on purpose: its GPU test was written after this round's GPU minutes were spent, so it has only run on the
emulator; with `pytest -x` a surprise here must not hide the rest of the tier.)"""
import random

import pytest

from manta_amd._capi import MantaError, assembly_text, align_text
from oracle_lib import asm_opts


def pile(seed, n_reads, read_len, ref_len, err=0.005):
    rng = random.Random(seed)
    ref = "".join(rng.choice("ACGT") for _ in range(ref_len))
    reads = []
    for _ in range(n_reads):
        s = rng.randint(0, ref_len - read_len)
        r = [c if rng.random() > err else rng.choice("ACGT") for c in ref[s:s + read_len]]
        reads.append("".join(r))
    return reads


def check_limits(lib, oracle, wide_reads):
    # word length 128 = the maximum (8 key dwords), and the word-length ladder up to it
    o = asm_opts(minWordLength=108, maxWordLength=128, wordStepSize=10, minCoverage=2)
    reads = pile(1, 40, 250, 600)
    assert assembly_text(lib.assemble_batch(o, [reads])[0]) == oracle.assemble(o, reads)
    o = asm_opts(minWordLength=128, maxWordLength=128, wordStepSize=5, minCoverage=2)
    r = lib.assemble_batch(o, [reads])[0]
    assert r["final_word_length"] == 128 and r["contigs"] and assembly_text(r) == oracle.assemble(o, reads)
    # several hundred reads: read sets of many qwords (cooperative walk, wide selectContigs path)
    o = asm_opts(minWordLength=21, maxWordLength=31, wordStepSize=10, minCoverage=3)
    reads = pile(2, wide_reads, 40, 160)
    r = lib.assemble_batch(o, [reads])[0]
    assert r["n_words"] > 4 and assembly_text(r) == oracle.assemble(o, reads)
    # beyond the envelope: word length 129, more than ~1000 reads
    with pytest.raises(MantaError):
        lib.assemble_batch(asm_opts(minWordLength=100, maxWordLength=129), [pile(3, 5, 200, 300)])
    # (the pile sizes a workspace shared by the whole call: the call is refused as a whole and EVERY record carries the code)
    res = lib.assemble_batch(asm_opts(minWordLength=21, maxWordLength=21), [pile(4, 1100, 30, 90), pile(5, 10, 30, 90)], strict=False)
    assert [r["status"] for r in res] == [-5, -5]


def test_emulated_assembler_limits(emu, oracle):
    check_limits(emu, oracle, 300)


def long_query_cases(seed, n_per_kind, qlens):
    """queries longer than one strip of the widest kernel (64 lanes x 32 columns = 2048): indel-bearing copies of the reference,
    breakend-spanning ones for the jump aligner, plus a query unrelated to its reference"""
    rng = random.Random(seed)

    def rs(n):
        return "".join(rng.choice("ACGT") for _ in range(n))

    def mut(x):
        out = []
        for c in x:
            u = rng.random()
            if u < 0.004:
                continue
            if u < 0.008:
                out.append(rng.choice("ACGT"))
            elif u < 0.02:
                c = rng.choice("ACGTN")
            out.append(c)
        return "".join(out)

    cases = []
    for kind in (0, 1, 2):
        for i in range(n_per_kind):
            ql = qlens[i % len(qlens)]
            if kind == 2:
                a, b = rs(ql // 2 + rng.randint(20, 200)), rs(ql // 2 + rng.randint(20, 200))
                q = mut(a[-(ql // 2):] + rs(rng.choice([0, 0, 5])) + b[:ql - ql // 2])
                cases.append((kind, (q, a, b)))
            else:
                q = rs(ql)
                cut = rng.randint(100, ql - 100)
                ref = rs(rng.randint(0, 60)) + q[:cut] + rs(rng.choice([0, 7, 40, 300])) + q[cut + rng.choice([0, 0, 25]):] + rs(rng.randint(0, 60))
                if i == n_per_kind - 1:
                    ref = rs(ql // 3)  # unrelated, shorter than the query
                cases.append((kind, (mut(q), ref, None)))
    return cases


def check_align_long_queries(lib, checker, n_per_kind=2, qlens=(2049, 4096, 4100, 6500)):
    """the reference has no query length limit (GlobalLargeIndelAlignerImpl.hpp:52-54, GlobalJumpAlignerImpl.hpp:60-63): neither
    has the device path -- queries past 2048 bases run in strips of 2048 columns"""
    by_kind = {0: ([2, -8, -12, -1, -1, 0], 0), 1: ([2, -8, -24, -1, -1, 0], -100), 2: ([2, -8, -12, -1, -1, 0], -100)}
    for kind, (sc, extra) in by_kind.items():
        probs = [p for k, p in long_query_cases(11 + kind, n_per_kind, qlens) if k == kind]
        res = lib.align_batch(kind, sc, extra, [tuple(x for x in p if x is not None) for p in probs])
        for p, r in zip(probs, res):
            assert r["status"] == 0
            assert align_text(kind, r) == checker.align(kind, sc, extra, *p), (kind, len(p[0]), len(p[1]))


def check_align_limits(lib, oracle):
    rng = random.Random(7)
    sc = [2, -8, -12, -1, -1, 0]
    q = "".join(rng.choice("ACGT") for _ in range(2048))  # the longest query of a single strip
    ref = q[:1000] + "".join(rng.choice("ACGT") for _ in range(30)) + q[1000:]
    res = lib.align_batch(1, sc, -24, [(q, ref), (q + "A", ref), (q + q[:77], ref + q[:60])])
    for (a, b), r in zip([(q, ref), (q + "A", ref), (q + q[:77], ref + q[:60])], res):
        assert r["status"] == 0 and align_text(1, r) == oracle.align(1, sc, -24, a, b)  # one base past the strip: two strips, same answer


def test_emulated_long_queries_match_the_reference(emu, reflib):
    check_align_long_queries(emu, reflib, n_per_kind=2, qlens=(2049, 4100))


def test_restatement_matches_the_reference_on_long_queries(oracle, reflib):
    by_kind = {0: ([2, -8, -12, -1, -1, 0], 0), 1: ([2, -8, -24, -1, -1, 0], -100), 2: ([2, -8, -12, -1, -1, 0], -100)}
    for kind, p in long_query_cases(31, 2, (2049, 3000, 5000)):
        sc, extra = by_kind[kind]
        assert oracle.align(kind, sc, extra, *p) == reflib.align(kind, sc, extra, *p)


def test_emulated_aligner_limits(emu, oracle):
    check_align_limits(emu, oracle)


@pytest.mark.gpu
def test_gpu_limits(gpu, oracle):
    check_limits(gpu, oracle, 700)
    check_align_limits(gpu, oracle)
    check_align_long_queries(gpu, oracle, n_per_kind=6, qlens=(2049, 2500, 4096, 4100, 6500, 8192))


def check_degenerate_piles(lib, oracle):
    """empty and ragged inputs of the iterative assembler, all in ONE batch next to an ordinary pile"""
    o = asm_opts(minWordLength=11, maxWordLength=21, wordStepSize=5, minCoverage=1, minUnusedReads=1, minSupportReads=1)
    piles = [[], ["ACGTACGTAC"], ["ACG", "T", "GGCCA"], ["NNNNNNNNNNNNNNNNNNNNNN"] * 3, ["ACGTTGCAAGGCTTACCGGATTACCA"] * 4,
             pile(9, 30, 60, 200), ["A" * 40, "A" * 35, "C" * 40]]
    res = lib.assemble_batch(o, piles)
    for reads, r in zip(piles, res):
        assert r["status"] == 0 and assembly_text(r) == oracle.assemble(o, reads), reads[:2]


def test_emulated_degenerate_piles(emu, oracle, reflib):
    check_degenerate_piles(emu, oracle)
    check_degenerate_piles(emu, reflib)  # and the unmodified reference says the same


@pytest.mark.gpu
def test_gpu_degenerate_piles(gpu, oracle):
    check_degenerate_piles(gpu, oracle)


# piles that do not fit the typical-case workspace (found by a random sweep against the reference: maxAssemblyCount 1 leaves almost
# no slack for contigs that come back as pseudo reads / for the repeat search's scratch): they run again on a worst-case workspace
CAPACITY_CASES = [
    (610448, dict(minWordLength=6, maxWordLength=9, wordStepSize=3, minCoverage=1, minConservativeCoverage=3, minUnusedReads=2,
                  minSupportReads=1, maxAssemblyCount=1)),
    (549478, dict(minWordLength=11, maxWordLength=12, wordStepSize=1, minCoverage=2, minConservativeCoverage=3, minUnusedReads=2,
                  minSupportReads=2, maxAssemblyCount=1)),
    (192159, dict(minWordLength=11, maxWordLength=11, wordStepSize=3, minCoverage=1, minConservativeCoverage=1, minUnusedReads=3,
                  minSupportReads=2, maxAssemblyCount=1)),
]


def check_capacity_rerun(lib, oracle):
    import numpy as np
    from manta_amd._capi import BatchOutput, small_sv_text
    from small_asm_cases import random_case
    from synth import config2_batch
    for seed, o in CAPACITY_CASES:
        reads = random_case(seed)[1]
        opts = asm_opts(**o)
        # next to ordinary piles in the same call: only the one that needs it runs again
        res = lib.assemble_batch(opts, [pile(3, 20, 40, 120), reads, pile(4, 25, 40, 120)])
        assert [r["status"] for r in res] == [0, 0, 0]
        assert assembly_text(res[1]) == oracle.assemble(opts, reads), seed
    # the same through the fused small-SV batch call (streamed upload, schedule + aligners behind the rerun)
    seed, o = CAPACITY_CASES[0]
    reads = random_case(seed)[1]
    opts = asm_opts(**o)
    batch = list(config2_batch(3, seed=77, n_reads=12, read_len=30, ref_len=300))  # (reads no longer than the pile's: same tight capacities)
    bases, read_off, begin = batch[0], batch[1], batch[2]
    # splice the pile in as locus 1 (keeping its reference window and cuts)
    loci = []
    for l in range(3):
        rb, re = int(begin[l]), int(begin[l + 1])
        loci.append([bytes(bases[int(read_off[r]):int(read_off[r + 1])]).decode("latin-1") for r in range(rb, re)])
    loci[1] = reads
    flat = [r.encode("latin-1") for rs in loci for r in rs]
    new_off = np.zeros(len(flat) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in flat], out=new_off[1:])
    new_bases = np.frombuffer(b"".join(flat) + b"\\0" * 64, dtype=np.uint8)
    new_begin = np.zeros(4, dtype=np.uint32)
    np.cumsum([len(rs) for rs in loci], out=new_begin[1:])
    cuts = np.tile(np.array([20, 20, 100, 100], dtype=np.int32), (3, 1))
    nb = (new_bases, new_off, new_begin, batch[3], batch[4], cuts)
    out = BatchOutput(lib, "smallsv", 3, 10, 1 << 20, 1 << 16, 1 << 18)
    sc = [2, -8, -24, -1, -1, 0]
    lib.smallsv_batch(opts, sc, -100, nb, out)
    res = out.decode(np.diff(new_begin))
    for l in range(3):
        ref = bytes(batch[3][int(batch[4][l]):int(batch[4][l + 1])]).decode("latin-1")
        assert small_sv_text(res[l]) == oracle.small_sv_locus(opts, sc, -100, loci[l], ref, tuple(int(x) for x in cuts[l])), l


def test_emulated_capacity_rerun(emu, oracle):
    check_capacity_rerun(emu, oracle)


@pytest.mark.gpu
def test_gpu_capacity_rerun(gpu, oracle):
    check_capacity_rerun(gpu, oracle)


def check_jump_overrun(lib, oracle):
    """offEdge 0 (never used by Manta, legal for the aligner): path 1 runs past the end of ref1 and the reference's jumpRange loop
    starts beyond its string (JumpAlignerBaseImpl.hpp:203-226); kernel and restatement stop there, = the reference's answer"""
    import json
    import os
    c = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "jump_overrun_case.json")))
    res = lib.align_batch(c["kind"], c["scores"], c["extra"], [(c["q"], c["r1"], c["r2"])])[0]
    assert align_text(c["kind"], res) == c["want"] == oracle.align(c["kind"], c["scores"], c["extra"], c["q"], c["r1"], c["r2"])


def test_emulated_jump_overrun(emu, oracle):
    check_jump_overrun(emu, oracle)


@pytest.mark.gpu
def test_gpu_jump_overrun(gpu, oracle):
    check_jump_overrun(gpu, oracle)
