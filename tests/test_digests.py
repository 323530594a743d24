"""Full-size reference parity on the GPU through digests.

tests/golden/make_digests.py ran the UNMODIFIED reference sources (oracle/_ref) over every locus of the metric's workload
(config 2, 10 000 loci, seed 12345) and over 2 048 config-5 shaped loci (mixed word lengths) and stored the SHA-256 of the
canonical text per locus.  Here the device results are rendered to the same text and every locus is compared.
The CPU tier pins the digest files themselves: the CPU restatement (oracle/) must reproduce a sample of them."""
import hashlib
import os

import numpy as np
import pytest

from manta_amd._capi import BatchOutput, assembly_text, pack_spanning, small_sv_text
from oracle_lib import asm_opts
from synth import config2_batch, config5_locus, mixed_shape_batch, unpack_locus
from test_spanning_pipeline import SC as SPAN_SC, oracle_locus

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
C2_OPTS = asm_opts(minWordLength=31, maxWordLength=76, wordStepSize=5)
C2_SCORES = [2, -8, -24, -1, -1, 0]
C5_CUTS = (100, 100, 100, 100)


def digests(name):
    raw = open(os.path.join(GOLD, name), "rb").read()
    return [raw[i:i + 32] for i in range(0, len(raw), 32)]


def c5_text(asm_text, aligns):
    return asm_text + "".join("span %d score=%d ins=%d range=%d begin1=%d cigar1=%s begin2=%d cigar2=%s uncut=%d\n" % ((i,) + tuple(a))
                              for i, a in enumerate(aligns))


def test_digest_files_match_the_restatement_on_a_sample(oracle):
    d2 = digests("config2_digests.bin")
    assert len(d2) == 10000
    batch = config2_batch(10000, seed=12345)
    for l in (0, 1, 4999, 9999):
        reads, ref, cuts = unpack_locus(batch, l)
        assert hashlib.sha256(oracle.small_sv_locus(C2_OPTS, C2_SCORES, -100, reads, ref, cuts).encode("latin-1")).digest() == d2[l]
    d5 = digests("config5_digests.bin")
    assert len(d5) == 2048
    for i in (0, 2047):
        reads, ref1, ref2, k, kmax = config5_locus(i)
        o = asm_opts(minWordLength=k, maxWordLength=kmax, minContigLength=75)
        text, aligns = oracle_locus(oracle, o, reads, ref1, ref2, C5_CUTS)
        assert hashlib.sha256(c5_text(text, aligns).encode("latin-1")).digest() == d5[i]


MX_OPTS = asm_opts(minWordLength=41, maxWordLength=76, wordStepSize=5)


def test_mixed_digest_file_matches_the_restatement_on_a_sample(oracle):
    """bench.py's mixed_shape batch (read counts 3..1000): the reference's digests against the CPU restatement on a few small piles"""
    d = digests("mixed_digests.bin")
    assert len(d) == 2048
    batch = mixed_shape_batch(2048, seed=777)
    n_reads = np.diff(batch[2])
    small = [int(l) for l in np.argsort(n_reads, kind="stable")[:3]] + [int(np.flatnonzero((n_reads > 60) & (n_reads < 120))[0])]
    for l in small:
        reads, ref, cuts = unpack_locus(batch, l)
        assert hashlib.sha256(oracle.small_sv_locus(MX_OPTS, C2_SCORES, -100, reads, ref, cuts).encode("latin-1")).digest() == d[l], l


def test_emulated_spanning_batch_with_tandem_piles_matches_reference_digests(emu, monkeypatch):
    """the whole spanning call (per-locus word lengths, assembler + jump aligner) on config-5 loci 0, 1, 37 and 64 against the reference's
    digests: two of them are tandem-repeat piles whose k-mer graph is cyclic at one / at all four of their word lengths -- the big class'
    word-length rounds (graph_big -> repeat_big -> contig_big) inside the fused pipeline, two blocks"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    want = digests("config5_digests.bin")
    ids = (0, 1, 37, 64)
    loci = [config5_locus(i) for i in ids]
    batch = pack_spanning([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [C5_CUTS] * len(ids))
    min_wl = np.array([l[3] for l in loci], dtype=np.uint32)
    max_wl = np.array([l[4] for l in loci], dtype=np.uint32)
    out = BatchOutput(emu, "spanning", len(ids), 10, 8 << 20, 1 << 20, 2 << 20)
    emu.spanning_batch(asm_opts(minWordLength=41, minContigLength=75), SPAN_SC, -100, batch, out, min_wl=min_wl, max_wl=max_wl, block_loci=2, n_workers=1)
    res = out.decode(np.diff(batch[2]))
    for i, r in zip(ids, res):
        got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"])
               for a in r["aligns"]]
        assert hashlib.sha256(c5_text(assembly_text(r), got).encode("latin-1")).digest() == want[i], i
        assert r["n_iterations"] == {0: 2, 1: 1, 37: 1, 64: 4}[i] and r["cyclic_iterations"] == {0: 1, 1: 0, 37: 1, 64: 4}[i]


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("asm_path", ["general", "fast"])
def test_gpu_config2_all_10000_loci_match_reference_digests(gpu, monkeypatch, asm_path):
    """fast = the default: the LDS pipeline (graph_kernel -> contig_kernel) with assemble_kernel on what it hands back;
    general = assemble_kernel alone (MANTA_AMD_ASM_PATH)"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", asm_path)
    want = digests("config2_digests.bin")
    batch = config2_batch(10000, seed=12345)
    out = BatchOutput(gpu, "smallsv", 10000, 10, 64 << 20, 8 << 20, 16 << 20)
    gpu.smallsv_batch(C2_OPTS, C2_SCORES, -100, batch, out, block_loci=2500, n_workers=4)
    res = out.decode(np.diff(batch[2]))
    bad = [l for l, r in enumerate(res) if hashlib.sha256(small_sv_text(r).encode("latin-1")).digest() != want[l]]
    assert not bad, "%d of 10000 loci differ from the reference, first: %s" % (len(bad), bad[:10])


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("asm_path", ["general", "fast"])
def test_gpu_config5_2048_loci_match_reference_digests(gpu, monkeypatch, asm_path):
    """fast = the default: the LDS pipeline's big class, one round of graph_big_kernel -> repeat_big_kernel -> contig_big_kernel per word
    length (the 186 tandem-repeat piles of the set stay on it through up to eleven word lengths); general = assemble_kernel alone"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", asm_path)
    want = digests("config5_digests.bin")
    n = len(want)
    loci = [config5_locus(i) for i in range(n)]
    batch = pack_spanning([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [C5_CUTS] * n)
    min_wl = np.array([l[3] for l in loci], dtype=np.uint32)
    max_wl = np.array([l[4] for l in loci], dtype=np.uint32)
    out = BatchOutput(gpu, "spanning", n, 10, 64 << 20, 8 << 20, 16 << 20)
    gpu.spanning_batch(asm_opts(minWordLength=41, minContigLength=75), SPAN_SC, -100, batch, out, min_wl=min_wl, max_wl=max_wl,
                       block_loci=512, n_workers=4)
    res = out.decode(np.diff(batch[2]))
    bad = []
    for i, r in enumerate(res):
        got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"])
               for a in r["aligns"]]
        if hashlib.sha256(c5_text(assembly_text(r), got).encode("latin-1")).digest() != want[i]:
            bad.append(i)
    assert not bad, "%d of %d loci differ from the reference, first: %s" % (len(bad), n, bad[:10])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_mixed_shape_2048_loci_match_reference_digests(gpu):
    """bench.py's mixed_shape batch -- read counts log-uniform 3..1000, Manta's production word lengths 41..76 -- takes every route of
    the assembler stage in one call: the LDS pipeline's small and big class, what they hand back, and the general kernel for the piles
    outside both envelopes.  Every locus against the unmodified reference's digests."""
    want = digests("mixed_digests.bin")
    n = len(want)
    batch = mixed_shape_batch(n, seed=777)
    n_reads = np.diff(batch[2])
    tot_b = int(batch[1][-1])
    out = BatchOutput(gpu, "smallsv", n, 10, 2 * tot_b // 8 + 4096 * n + (1 << 20), 40 * int(n_reads.sum()) // 64 + 128 * n + 4096, 512 * n + 4096)
    gpu.smallsv_batch(MX_OPTS, C2_SCORES, -100, batch, out)
    st = out.stats_dict()
    res = out.decode(n_reads)
    bad = [l for l, r in enumerate(res) if r["status"] != 0 or hashlib.sha256(small_sv_text(r).encode("latin-1")).digest() != want[l]]
    assert not bad, "%d of %d loci differ from the reference, first: %s (reads: %s)" % (len(bad), n, bad[:10], [int(n_reads[l]) for l in bad[:10]])
    assert st["n_loci_lds_small"] > 0 and st["n_loci_lds_big"] > 0 and st["n_loci_general"] > 0  # every route was taken


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_config5_one_block_early_alignment_pass_matches_reference_digests(gpu, monkeypatch):
    """one block, one worker: the spanning pipeline aligns the loci that are final after the first word length while the word-length
    rounds of the tandem piles still run (spanningRunImpl's early pass; pipelined workers -- the test above -- do not take it)"""
    monkeypatch.setenv("MANTA_AMD_ASM_PATH", "fast")
    want = digests("config5_digests.bin")
    n = len(want)
    loci = [config5_locus(i) for i in range(n)]
    batch = pack_spanning([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [C5_CUTS] * n)
    min_wl = np.array([l[3] for l in loci], dtype=np.uint32)
    max_wl = np.array([l[4] for l in loci], dtype=np.uint32)
    for rep in range(2):  # (the second call reuses the pipeline: the scratch of the early pass, the masked streams)
        out = BatchOutput(gpu, "spanning", n, 10, 64 << 20, 8 << 20, 16 << 20)
        gpu.spanning_batch(asm_opts(minWordLength=41, minContigLength=75), SPAN_SC, -100, batch, out, min_wl=min_wl, max_wl=max_wl, block_loci=n, n_workers=1)
        res = out.decode(np.diff(batch[2]))
        bad = []
        for i, r in enumerate(res):
            got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"])
                   for a in r["aligns"]]
            if hashlib.sha256(c5_text(assembly_text(r), got).encode("latin-1")).digest() != want[i]:
                bad.append(i)
        assert not bad, "%d of %d loci differ from the reference, first: %s" % (len(bad), n, bad[:10])
