"""HIP assembler kernel vs the oracle: wave emulator (CPU tier, small piles) and GPU (-m gpu, full shapes)."""
import json
import os
import random

import pytest

from manta_amd._capi import assembly_text
from oracle_lib import asm_opts
from synth import small_indel_locus, breakend_locus, repeat_rich_pile

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ASM = json.load(open(os.path.join(GOLD, "assembler_reference_tests.json")))


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


def test_emulated_assembler_reference_golden(emu):
    for c in ASM:
        if any(set(r) - set("ACGTN") for r in c["reads"]):
            continue  # junk-alphabet reads: outside the supported envelope (reported as MANTA_E_UNSUPPORTED)
        r = emu.assemble_batch(asm_opts(**c["opts"]), [c["reads"]])[0]
        assert assembly_text(r) == c["ref_text"], c["name"]


def test_junk_alphabet_is_reported_not_guessed(emu):
    c = ASM[0]
    r = emu.assemble_batch(asm_opts(**c["opts"]), [c["reads"]], strict=False)[0]
    assert r["status"] == -5


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
