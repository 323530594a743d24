"""Fused spanning pipeline (manta_spanning_*: assemble -> GlobalJumpAligner on the cut references -> re-align rule) against
the oracle composed call by call the way alignJumpContigs does it (SVCandidateAssemblyRefiner.cpp:1663-1717).
(The whole-refiner comparison against the reference's own refiner lives in tests/test_refiner.py.)"""
import re

import pytest

from manta_amd._capi import SpanningBatch, assembly_text
from oracle_lib import asm_opts
from synth import breakend_locus

SC = [2, -8, -12, -1, -1, 0]  # SVRefinerOptions.hpp:43 spanningAlignScores, jumpScore -100 (:45)
ALN = re.compile(r"score=(-?\d+) jumpInsertSize=(\d+) jumpRange=(\d+) begin1=(-?\d+) cigar1=(\S*) begin2=(-?\d+) cigar2=(\S*)")


def oracle_locus(oracle, opts, reads, ref1, ref2, cuts):
    """contig text + per contig (score, ins, range, begin1, cigar1, begin2, cigar2, is_uncut)"""
    a1l, a1t, a2l, a2t = cuts
    text = oracle.assemble(opts, reads)
    seqs = re.findall(r"^contig \d+ seq=(\S+)", text, flags=re.M)
    out = []
    for q in seqs:
        def aln(l1, t1, l2, t2):
            m = ALN.match(oracle.align(2, SC, -100, q.encode(), ref1[l1:len(ref1) - t1], ref2[l2:len(ref2) - t2]))
            return [int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5), int(m.group(6)), m.group(7)]
        r = aln(a1l, a1t, a2l, a2t)
        ref_len = sum(int(n) for n, op in re.findall(r"(\d+)([=XDN])", r[4]))
        ref1_end = len(ref1) - a1l - a1t - 1
        if r[1] > 0 and ((ref1_end - (r[3] + ref_len) < 5) or r[5] < 5):
            a1l = a1t = a2l = a2t = 0  # shared by the later contigs of the locus (:1691-1694)
            r = aln(0, 0, 0, 0)
        r[3] += a1l
        r[5] += a2l
        out.append(tuple(r) + (int((a1l, a1t, a2l, a2t) == (0, 0, 0, 0) and tuple(cuts) != (0, 0, 0, 0)),))
    return text, out


def check(lib, oracle, loci, opts, cuts_list):
    b = SpanningBatch(lib, opts, SC, -100)
    b.upload([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], cuts_list)
    b.run()
    res = b.download()
    n_uncut = 0
    for (reads, ref1, ref2), cuts, r in zip(loci, cuts_list, res):
        text, want = oracle_locus(oracle, opts, reads, ref1, ref2, cuts)
        assert assembly_text(r).split("\nread")[0].startswith(text.split("\nread")[0][:40])
        got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"])
               for a in r["aligns"]]
        assert [c["seq"] for c in r["contigs"]] == re.findall(r"^contig \d+ seq=(\S+)", text, flags=re.M)
        assert got == want
        n_uncut += sum(g[7] for g in got)
    return b.stats(), n_uncut


def small_loci(n, seed0=0):
    return [breakend_locus(seed0 + s, n_reads=24, read_len=70, ref_len=320) for s in range(n)]


def test_emulated_spanning_pipeline(emu, oracle):
    loci = small_loci(6)
    o = asm_opts(minWordLength=25, maxWordLength=45, minContigLength=40)
    check(emu, oracle, loci, o, [(30, 30, 30, 30)] * len(loci))


def test_emulated_spanning_realign_rule(emu, oracle):
    """cuts placed so that the junction sits < 5 bases from the cut edge of reference 2: round 2 must run"""
    loci = small_loci(8, seed0=100)
    o = asm_opts(minWordLength=25, maxWordLength=45, minContigLength=40)
    # ref2 half = 160 +- 20: cutting 138..178 leading bases puts the breakpoint near/inside the cut edge for some loci
    total = 0
    for lead2 in (140, 150, 158):
        _, n_uncut = check(emu, oracle, loci, o, [(20, 20, lead2, 10)] * len(loci))
        total += n_uncut
    assert total > 0


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_gpu_spanning_pipeline_config5_shape(gpu, oracle):
    loci = [breakend_locus(s) for s in range(24)]  # 200 reads x 250 bp, 1 % N, 10 % tandem-repeat loci
    o = asm_opts(minWordLength=41, minContigLength=75)
    st, _ = check(gpu, oracle, loci, o, [(100, 100, 100, 100)] * len(loci))
    assert st["n_alignments"] > 24
    loci = small_loci(64, seed0=100)
    o = asm_opts(minWordLength=25, maxWordLength=45, minContigLength=40)
    _, n_uncut = check(gpu, oracle, loci, o, [(20, 20, 150, 10)] * len(loci))
    assert n_uncut > 0
