"""Read-pile construction: ReadPileBuilder (manta_amd/host/read_pile.hpp) turns BAM records into the packed pile
layout exactly as insertAssemblyRead's string handling would (SVCandidateAssembler.cpp:102-136), and the packed-input path of the
pipelines gives the same results as the 1-byte-per-base path."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from manta_amd._capi import BatchOutput, SmallSvBatch, SpanningBatch, pack_piles, pack_spanning, small_sv_text
from oracle_lib import asm_opts
from synth import breakend_locus, config2_batch, small_indel_locus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BAM_CHARS = "=ACMGRSVTWYHKDBN"  # htslib seq_nt16_str; bam_seq.hpp:41-59 reads A,C,G,T,= as themselves and everything else as 'N'
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


@pytest.fixture(scope="module")
def pile_lib():
    so = os.path.join(CPP, "libhost_pile.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "manta_amd", "host"), os.path.join(CPP, "host_pile_capi.cpp"), "-o", so])
    return ctypes.CDLL(so)


def reference_string(nibbles, qual, min_qval, is_reversed):
    """what insertAssemblyRead leaves in `reads` (restated: get_string, Q mask, reverseCompStr)"""
    s = [c if c in "ACGT=" else "N" for c in (BAM_CHARS[n] for n in nibbles)]
    s = ["N" if q < min_qval else c for c, q in zip(s, qual)]
    if is_reversed:
        s = [COMP[c] for c in reversed(s)]
    return "".join(s)


def pack_records(pile_lib, records, min_qval):
    n = len(records)
    seqs, quals = [], []
    for nib, q, _ in records:
        by = bytearray((len(nib) + 1) // 2)
        for i, v in enumerate(nib):
            by[i >> 1] |= v << (4 if i % 2 == 0 else 0)
        seqs.append((ctypes.c_uint8 * max(1, len(by)))(*by))
        quals.append((ctypes.c_uint8 * max(1, len(q)))(*q))
    lens = [len(r[0]) for r in records]
    cap = sum((l + 15) // 16 + 1 for l in lens) + 4
    codes, mask = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32)
    coff, moff, rlen = np.zeros(n + 1, dtype=np.uint64), np.zeros(n + 1, dtype=np.uint64), np.zeros(n + 1, dtype=np.uint32)
    acc = (ctypes.c_int * n)()
    P8 = ctypes.POINTER(ctypes.c_uint8)
    k = pile_lib.mine_pack_bam_reads(n, (P8 * n)(*[ctypes.cast(s, P8) for s in seqs]), (P8 * n)(*[ctypes.cast(q, P8) for q in quals]),
                                     (ctypes.c_uint * n)(*lens), (ctypes.c_int * n)(*[int(r[2]) for r in records]), min_qval,
                                     codes.ctypes.data_as(ctypes.c_void_p), mask.ctypes.data_as(ctypes.c_void_p),
                                     coff.ctypes.data_as(ctypes.c_void_p), moff.ctypes.data_as(ctypes.c_void_p),
                                     rlen.ctypes.data_as(ctypes.c_void_p), acc)
    return k, list(acc), codes, mask, coff, moff, rlen


def test_builder_matches_insert_assembly_read_semantics(pile_lib):
    rng = np.random.default_rng(3)
    records = []
    for i in range(300):
        ln = int(rng.integers(1, 260))
        nib = rng.choice([1, 2, 4, 8], size=ln)
        amb = rng.random(ln) < 0.03
        nib[amb] = rng.choice([3, 5, 6, 7, 9, 10, 14, 15], size=int(amb.sum()))
        if i % 37 == 5:
            nib[int(rng.integers(0, ln))] = 0  # '=': the record is refused
        q = rng.integers(0, 42, size=ln)
        records.append(([int(x) for x in nib], [int(x) for x in q], bool(rng.random() < 0.5)))
    min_q = 5
    k, acc, codes, mask, coff, moff, rlen = pack_records(pile_lib, records, min_q)
    kept = [reference_string(*r[:2], min_q, r[2]) for r in records if 0 not in r[0]]  # (a reversed '=' is a fatal base_error in the reference)
    assert acc == [int(0 not in r[0]) for r in records] and k == len(kept)
    # the same strings through the text packer (numpy restatement) give the same arrays
    bases = np.frombuffer("".join(kept).encode() + b"\0", dtype=np.uint8)
    off = np.zeros(len(kept) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in kept], out=off[1:])
    p = pack_piles(bases, off, np.array([0, len(kept)], dtype=np.uint32))
    assert np.array_equal(p.read_len, rlen[:k]) and np.array_equal(p.code_off, coff[:k + 1]) and np.array_equal(p.mask_off, moff[:k + 1])
    assert np.array_equal(p.codes, codes[:len(p.codes)]) and np.array_equal(p.nmask, mask[:len(p.nmask)])


def n_masked_batch(n, seed):
    loci = [small_indel_locus(seed + s, n_reads=24, read_len=70, ref_len=400, n_rate=(0.02 if s % 2 else 0.0)) for s in range(n)]
    from manta_amd._capi import pack_loci
    bases, read_off, begin = pack_loci([l[0] for l in loci])
    refs = np.frombuffer(b"".join(l[1] for l in loci) + b"\0", dtype=np.uint8)
    ref_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(400)
    cuts = np.tile(np.array([40, 40, 160, 160], dtype=np.int32), (n, 1))
    return bases, read_off, begin, refs, ref_off, np.ascontiguousarray(cuts)


def check_piles_equal_bytes(lib, n):
    opts, sc = asm_opts(minWordLength=21, maxWordLength=41), [2, -8, -24, -1, -1, 0]
    batch = n_masked_batch(n, 500)
    a = SmallSvBatch(lib, opts, sc, -100)
    a.upload_packed(*batch)
    a.run()
    want = [small_sv_text(r) for r in a.download()]
    piles = pack_piles(batch[0], batch[1], batch[2])
    b = SmallSvBatch(lib, opts, sc, -100)
    b.upload_piles(piles, batch[3], batch[4], batch[5])
    b.run()
    assert [small_sv_text(r) for r in b.download()] == want
    out = BatchOutput(lib, "smallsv", n, 10, 1 << 20, 1 << 16, 1 << 18)
    lib.smallsv_batch_piles(opts, sc, -100, piles, batch[3], batch[4], batch[5], out, block_loci=max(1, n // 3), n_workers=2)
    assert [small_sv_text(r) for r in out.decode(np.diff(batch[2]))] == want
    assert any("N" not in w for w in want)


def test_emulated_packed_piles_equal_byte_piles(emu):
    check_piles_equal_bytes(emu, 7)


@pytest.mark.gpu
def test_gpu_packed_piles_equal_byte_piles(gpu):
    check_piles_equal_bytes(gpu, 120)
    # config-2 shape through the whole-batch call: packed input against the 1-byte-per-base path
    batch = config2_batch(1000, seed=12345)
    piles = pack_piles(batch[0], batch[1], batch[2])
    out = BatchOutput(gpu, "smallsv", 1000, 10, 8 << 20, 1 << 20, 2 << 20)
    gpu.smallsv_batch_piles(asm_opts(minWordLength=31, maxWordLength=76, wordStepSize=5), [2, -8, -24, -1, -1, 0], -100, piles, batch[3],
                            batch[4], batch[5], out, block_loci=500, n_workers=2)
    ref_out = BatchOutput(gpu, "smallsv", 1000, 10, 8 << 20, 1 << 20, 2 << 20)
    gpu.smallsv_batch(asm_opts(minWordLength=31, maxWordLength=76, wordStepSize=5), [2, -8, -24, -1, -1, 0], -100, batch, ref_out,
                      block_loci=500, n_workers=2)
    n_reads = np.diff(batch[2])
    assert [small_sv_text(r) for r in out.decode(n_reads)] == [small_sv_text(r) for r in ref_out.decode(n_reads)]
