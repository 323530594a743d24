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


def check_align_limits(lib, oracle):
    rng = random.Random(7)
    sc = [2, -8, -12, -1, -1, 0]
    q = "".join(rng.choice("ACGT") for _ in range(2048))  # the longest supported query
    ref = q[:1000] + "".join(rng.choice("ACGT") for _ in range(30)) + q[1000:]
    res = lib.align_batch(1, sc, -24, [(q, ref), (q + "A", ref)], strict=False)
    assert res[0]["status"] == 0 and align_text(1, res[0]) == oracle.align(1, sc, -24, q, ref)
    assert res[1]["status"] == -5  # one base too long: that alignment only


def test_emulated_aligner_limits(emu, oracle):
    check_align_limits(emu, oracle)


@pytest.mark.gpu
def test_gpu_limits(gpu, oracle):
    check_limits(gpu, oracle, 700)
    check_align_limits(gpu, oracle)


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
