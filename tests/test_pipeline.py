"""Fused small-SV pipeline (assemble -> 10-mer trim -> large-indel align) vs the oracle's ref_small_sv_locus text."""
import pytest

from manta_amd._capi import SmallSvBatch, small_sv_text
from oracle_lib import asm_opts
from synth import small_indel_locus, config2_batch, unpack_locus

SC = [2, -8, -24, -1, -1, 0]  # SVRefinerOptions.hpp:40 largeSVAlignScores, largeGapOpenScore -100 (:44)


def _run(lib, oracle, loci, opts, cuts):
    b = SmallSvBatch(lib, opts, SC, -100)
    b.upload([l[0] for l in loci], [l[1] for l in loci], [cuts] * len(loci))
    b.run()
    res = b.download()
    for (reads, ref), r in zip(loci, res):
        assert small_sv_text(r) == oracle.small_sv_locus(opts, SC, -100, reads, ref, cuts)
    return b.stats()


def test_emulated_pipeline_small(emu, oracle):
    loci = [small_indel_locus(s, n_reads=24, read_len=60, ref_len=500) for s in range(4)]
    _run(emu, oracle, loci, asm_opts(minWordLength=21, maxWordLength=41), (40, 40, 200, 200))


def test_emulated_pipeline_output_sizes_cover_many_contigs(emu, oracle):
    """the staged download sizes the caller's arenas from manta_smallsv_output_sizes: the CIGAR bound must hold for batches whose
    CIGARs add up to more than a handful of words (round 4: the schedule kernel's allocator went away, its total must not)"""
    loci = [small_indel_locus(100 + s, n_reads=20, read_len=50, ref_len=400, sub_rate=0.02) for s in range(40)]
    st = _run(emu, oracle, loci, asm_opts(minWordLength=15, maxWordLength=25), (20, 20, 150, 150))
    assert st["n_alignments"] >= 40


def test_emulated_pipeline_no_trim_hit_and_edges(emu, oracle):
    """reference windows unrelated to the reads (no 10-mer hit): the trim falls back to its loop bounds"""
    loci = [(small_indel_locus(1, n_reads=20, read_len=50, ref_len=400)[0], small_indel_locus(2, ref_len=400)[1]),
            small_indel_locus(3, n_reads=20, read_len=50, ref_len=400)]
    _run(emu, oracle, loci, asm_opts(minWordLength=15, maxWordLength=25), (20, 20, 150, 150))
    # whole window searchable (cuts 0, max cuts = window): only meaningful when a hit exists (otherwise the reference
    # itself forms a negative-length range, SVCandidateAssemblyRefiner.cpp:2032-2037)
    _run(emu, oracle, loci[1:], asm_opts(minWordLength=15, maxWordLength=25), (0, 0, 400, 400))


@pytest.mark.gpu
def test_gpu_pipeline_config2(gpu, oracle):
    batch = config2_batch(192, seed=777)
    o = asm_opts(minWordLength=31)
    b = SmallSvBatch(gpu, o, SC, -100)
    b.upload_packed(*batch)
    b.run()
    res = b.download()
    n_contigs = 0
    for l, r in enumerate(res):
        reads, ref, cuts = unpack_locus(batch, l)
        assert small_sv_text(r) == oracle.small_sv_locus(o, SC, -100, reads, ref, cuts), l
        n_contigs += len(r["contigs"])
    assert n_contigs >= 192
    st = b.stats()
    assert st["n_alignments"] == n_contigs


@pytest.mark.gpu
def test_gpu_pipeline_is_deterministic_and_reusable(gpu):
    """same resident batch, two runs -> identical results (work-queue order must not leak into outputs)"""
    batch = config2_batch(300, seed=5)
    b = SmallSvBatch(gpu, asm_opts(minWordLength=31), SC, -100)
    b.upload_packed(*batch)
    b.run()
    first = [small_sv_text(r) for r in b.download()]
    b.run()
    assert [small_sv_text(r) for r in b.download()] == first


@pytest.mark.gpu
def test_gpu_pipeline_planted_indel_recovered_at_full_size(gpu):
    """size-independent property at BASELINE's full config-2 size: every locus yields a contig whose CIGAR holds the
    planted indel (10..60 bp) as a single D or I segment."""
    import re
    batch = config2_batch(10000, seed=12345)
    b = SmallSvBatch(gpu, asm_opts(minWordLength=31), SC, -100)
    b.upload_packed(*batch)
    b.run()
    res = b.download()
    ok = 0
    for r in res:
        assert r["status"] == 0
        hit = False
        for a in r["aligns"]:
            for n, op in re.findall(r"(\d+)([=XIDS])", a["cigar1"]):
                if op in "ID" and 10 <= int(n) <= 60:
                    hit = True
        ok += hit
    assert ok >= 9950, ok


def _contigless_loci():
    full = small_indel_locus(5, n_reads=24, read_len=60, ref_len=500)
    return [(full[0][:2], full[1]), ([], full[1]), (full[0][:1], full[1]), full]


def test_emulated_pipeline_loci_without_contigs(emu, oracle):
    _run(emu, oracle, _contigless_loci(), asm_opts(minWordLength=21, maxWordLength=41), (40, 40, 200, 200))


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_gpu_pipeline_loci_without_contigs(gpu, oracle):
    """piles of 0-2 reads assemble to nothing.  Regression: the schedule kernel once spun forever on such batches on
    hardware only (lane-divergent loop exit produced by the compiler, see pipeline_kernels.hpp scheduleSlot)."""
    loci = _contigless_loci()
    _run(gpu, oracle, loci[:1], asm_opts(minWordLength=21, maxWordLength=41), (40, 40, 200, 200))
    _run(gpu, oracle, loci[:3], asm_opts(minWordLength=21, maxWordLength=41), (40, 40, 200, 200))
    _run(gpu, oracle, loci, asm_opts(minWordLength=21, maxWordLength=41), (40, 40, 200, 200))
