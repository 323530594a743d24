"""Whole-batch calls (manta_smallsv_batch / manta_spanning_batch): blocks pulled from a cost-ordered queue by several host
workers, results compacted into the caller's arenas.  Every locus must equal the oracle and the staged API; mixed word
lengths per locus go through the same launch."""
import numpy as np
import pytest

from manta_amd._capi import BatchOutput, SmallSvBatch, SpanningBatch, pack_spanning, small_sv_text
from oracle_lib import asm_opts
from synth import breakend_locus, config2_batch, unpack_locus
from test_spanning_pipeline import SC as SPAN_SC, oracle_locus

SCORES = [2, -8, -24, -1, -1, 0]


def small_batch(n, seed):
    return config2_batch(n, seed=seed, n_reads=20, read_len=80, ref_len=500)


def check_smallsv(lib, oracle, n, block, workers, mixed, streamed=True):
    batch = small_batch(n, 4242)
    batch = batch[:5] + (np.tile(np.array([40, 40, 200, 200], dtype=np.int32), (n, 1)),)
    opts = asm_opts(minWordLength=25, maxWordLength=45)
    min_wl = max_wl = None
    if mixed:
        min_wl = np.array([21 + 4 * (l % 4) for l in range(n)], dtype=np.uint32)
        max_wl = np.maximum(min_wl + 10, 41).astype(np.uint32)
    out = BatchOutput(lib, "smallsv", n, 10, 1 << 20, 1 << 16, 1 << 18)
    lib.smallsv_batch(opts, SCORES, -100, batch, out, min_wl=min_wl, max_wl=max_wl, block_loci=block, n_workers=workers,
                      streamed_upload=streamed)
    st = out.stats_dict()
    assert st["n_blocks"] == (n + block - 1) // block and st["n_workers"] == min(workers, st["n_blocks"])
    res = out.decode(np.diff(batch[2]))
    for l, r in enumerate(res):
        reads, ref, cuts = unpack_locus(batch, l)
        o = opts if not mixed else asm_opts(minWordLength=int(min_wl[l]), maxWordLength=int(max_wl[l]))
        assert small_sv_text(r) == oracle.small_sv_locus(o, SCORES, -100, reads, ref, cuts), l
    return res


def test_emulated_smallsv_batch_blocks_and_workers(emu, oracle):
    res = check_smallsv(emu, oracle, 11, block=4, workers=3, mixed=False)
    # the staged API gives the same thing
    batch = small_batch(11, 4242)
    batch = batch[:5] + (np.tile(np.array([40, 40, 200, 200], dtype=np.int32), (11, 1)),)
    p = SmallSvBatch(emu, asm_opts(minWordLength=25, maxWordLength=45), SCORES, -100)
    p.upload_packed(*batch)
    p.run()
    assert [small_sv_text(r) for r in p.download()] == [small_sv_text(r) for r in res]


def test_emulated_smallsv_batch_mixed_word_lengths(emu, oracle):
    check_smallsv(emu, oracle, 8, block=8, workers=1, mixed=True)


def spanning_case(n):
    loci = [breakend_locus(300 + s, n_reads=24, read_len=70, ref_len=320, tandem_frac=0.0) for s in range(n)]
    cuts = [(30, 30, 30, 30)] * n
    return loci, cuts, pack_spanning([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], cuts)


def check_spanning(lib, oracle, n, block, workers, streamed=True):
    loci, cuts, batch = spanning_case(n)
    ks = [25, 30, 35]
    min_wl = np.array([ks[l % 3] for l in range(n)], dtype=np.uint32)
    max_wl = np.full(n, 45, dtype=np.uint32)
    opts = asm_opts(minWordLength=25, maxWordLength=45, minContigLength=40)
    out = BatchOutput(lib, "spanning", n, 10, 1 << 20, 1 << 16, 1 << 18)
    lib.spanning_batch(opts, SPAN_SC, -100, batch, out, min_wl=min_wl, max_wl=max_wl, block_loci=block, n_workers=workers,
                       streamed_upload=streamed)
    res = out.decode(np.diff(batch[2]))
    for l, r in enumerate(res):
        o = asm_opts(minWordLength=int(min_wl[l]), maxWordLength=45, minContigLength=40)
        text, want = oracle_locus(oracle, o, loci[l][0], loci[l][1], loci[l][2], cuts[l])
        got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"])
               for a in r["aligns"]]
        assert got == want, l
        assert r["final_word_length"] >= int(min_wl[l])


def test_emulated_spanning_batch_mixed_word_lengths(emu, oracle):
    check_spanning(emu, oracle, 7, block=3, workers=2)


def test_emulated_batches_with_stage_gates(emu, oracle):
    """several workers: the blocks take turns in the assemble and align stages (stage gates); results are the same"""
    check_smallsv(emu, oracle, 11, block=3, workers=3, mixed=True)
    check_spanning(emu, oracle, 7, block=2, workers=2)


def test_emulated_batch_host_ranges(emu, oracle, monkeypatch):
    """the batch call scans its inputs and compacts its results in several host-thread ranges (forced here on a small batch)"""
    monkeypatch.setenv("MANTA_AMD_HOST_PARTS", "3")
    check_smallsv(emu, oracle, 11, block=11, workers=1, mixed=False)
    check_smallsv(emu, oracle, 10, block=4, workers=2, mixed=True)
    check_spanning(emu, oracle, 5, block=5, workers=1)


def overlap_knobs(lib, oracle, monkeypatch, n, block, workers):
    """the whole-batch small-SV call starts the DMA of the read bases before it sizes the batch (AsmStage::startStream) and compacts
    the assembler's outputs while the aligners run (manta_smallsv::whileAligning): each switched off gives the same records"""
    want = check_smallsv(lib, oracle, n, block=block, workers=workers, mixed=True)
    for knob in ("MANTA_AMD_NO_EARLY_STREAM", "MANTA_AMD_NO_EARLY_STAGE"):
        monkeypatch.setenv(knob, "1")
        got = check_smallsv(lib, oracle, n, block=block, workers=workers, mixed=True)
        monkeypatch.delenv(knob)
        assert [small_sv_text(r) for r in got] == [small_sv_text(r) for r in want], knob
    check_spanning(lib, oracle, max(4, n // 4), block=max(2, block // 4), workers=workers)  # (every locus against the oracle)
    monkeypatch.setenv("MANTA_AMD_NO_EARLY_STAGE", "1")
    check_spanning(lib, oracle, max(4, n // 4), block=max(2, block // 4), workers=workers)
    monkeypatch.delenv("MANTA_AMD_NO_EARLY_STAGE")


def test_emulated_batch_overlap_knobs(emu, oracle, monkeypatch):
    monkeypatch.setenv("MANTA_AMD_HOST_PARTS", "3")
    overlap_knobs(emu, oracle, monkeypatch, 10, block=4, workers=2)
    overlap_knobs(emu, oracle, monkeypatch, 9, block=9, workers=1)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_gpu_batch_overlap_knobs(gpu, oracle, monkeypatch):
    overlap_knobs(gpu, oracle, monkeypatch, 300, block=300, workers=1)
    monkeypatch.setenv("MANTA_AMD_HOST_PARTS", "4")
    overlap_knobs(gpu, oracle, monkeypatch, 200, block=64, workers=3)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_gpu_batch_calls(gpu, oracle):
    check_smallsv(gpu, oracle, 300, block=64, workers=4, mixed=False)
    check_smallsv(gpu, oracle, 96, block=32, workers=3, mixed=True)
    check_spanning(gpu, oracle, 60, block=16, workers=4)
    check_smallsv(gpu, oracle, 300, block=64, workers=2, mixed=True)
    check_spanning(gpu, oracle, 60, block=16, workers=2)
    check_smallsv(gpu, oracle, 200, block=200, workers=1, mixed=False, streamed=False)
    check_spanning(gpu, oracle, 40, block=40, workers=1, streamed=False)


def staged_smallsv_texts(lib, batch, opts):
    p = SmallSvBatch(lib, opts, SCORES, -100)
    p.upload_packed(*batch)
    p.run()
    return [small_sv_text(r) for r in p.download()]


def stress_shapes(lib, shapes, seed):
    """whole-batch call == staged API on odd shapes: tiny batches (fewer loci than upload chunks), blocks that do not divide the
    batch, more workers than blocks; the streamed upload, the stage gates and the host ranges all take their edge paths"""
    opts = asm_opts(minWordLength=25, maxWordLength=45)
    for i, (n, block, workers) in enumerate(shapes):
        batch = small_batch(n, seed + i)
        batch = batch[:5] + (np.tile(np.array([40, 40, 200, 200], dtype=np.int32), (n, 1)),)
        out = BatchOutput(lib, "smallsv", n, 10, 1 << 22, 1 << 18, 1 << 20)
        lib.smallsv_batch(opts, SCORES, -100, batch, out, block_loci=block, n_workers=workers)
        got = [small_sv_text(r) for r in out.decode(np.diff(batch[2]))]
        assert got == staged_smallsv_texts(lib, batch, opts), (n, block, workers)


def test_emulated_batches_with_blocking_upload(emu, oracle):
    """MANTA_BATCH_NO_STREAMED_UPLOAD (what a host that shares a GPU between processes passes): same results"""
    check_smallsv(emu, oracle, 9, block=4, workers=2, mixed=True, streamed=False)
    check_spanning(emu, oracle, 5, block=2, workers=1, streamed=False)


def check_history_across_different_batches(lib, n):
    """a pipeline launches its aligner buckets and stages its results from what its PREVIOUS run looked like (api.cpp: bucket
    history, speculative staging); batches of different shapes one after the other on the same pipelines -- short contigs, then
    long reference windows and longer reads, then bigger, then the first again -- must each equal the staged API"""
    opts = asm_opts(minWordLength=25, maxWordLength=45)
    shapes = [dict(n_reads=20, read_len=80, ref_len=500, cut=(40, 40, 200, 200)), dict(n_reads=30, read_len=140, ref_len=1400, cut=(10, 10, 600, 600)),
              dict(n_reads=12, read_len=60, ref_len=300, cut=(0, 0, 100, 100)), dict(n_reads=20, read_len=80, ref_len=500, cut=(40, 40, 200, 200))]
    for i, sh in enumerate(shapes):
        m = n * (3 if i == 2 else 1)
        batch = config2_batch(m, seed=7000 + i, n_reads=sh["n_reads"], read_len=sh["read_len"], ref_len=sh["ref_len"])
        batch = batch[:5] + (np.tile(np.array(sh["cut"], dtype=np.int32), (m, 1)),)
        out = BatchOutput(lib, "smallsv", m, 10, 1 << 22, 1 << 18, 1 << 21)
        lib.smallsv_batch(opts, SCORES, -100, batch, out, block_loci=0, n_workers=1)
        assert [small_sv_text(r) for r in out.decode(np.diff(batch[2]))] == staged_smallsv_texts(lib, batch, opts), i


def test_emulated_batch_history_across_different_batches(emu):
    check_history_across_different_batches(emu, 5)


def test_emulated_batch_odd_shapes(emu):
    stress_shapes(emu, [(1, 0, 0), (3, 2, 3), (9, 4, 2), (17, 0, 1)], 900)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_gpu_batch_odd_shapes_and_repeats(gpu):
    stress_shapes(gpu, [(1, 0, 0), (3, 2, 3), (7, 0, 1), (9, 4, 2), (65, 16, 5), (333, 100, 2), (1024, 0, 1), (1500, 700, 3)], 910)
    # the same pipelines again and again (buffers are reused, counters and events must start clean every call)
    for rep in range(6):
        stress_shapes(gpu, [(257, 64, 2), (40, 0, 1)], 950 + rep)
    check_history_across_different_batches(gpu, 120)
