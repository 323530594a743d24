"""Read-pile construction (second half): SVCandidateAssembler::getBreakendReads' per-read tests as a kernel
(manta_read_piles_batch, manta_amd/csrc/read_class_kernels.hpp).

Pinning chain:
  * the restatement (oracle/read_class_oracle.cpp) against the UNMODIFIED reference (oracle/_ref/libmanta_ref_bam.so: the real
    SVCandidateAssembler.cpp + htsapi + htslib over BAM files): on the demo BAMs through tests/golden/read_class_demo.json.gz
    (records and reference piles stored by tests/golden/make_read_class_golden.py) and, where the _ref library is present, live on
    synthetic BAM files written for the purpose (SAM -> BAM with the redist htslib);
  * the kernel against the restatement on random record batches (decisions, pile positions, per-candidate results, pile text),
    and directly against the reference piles of the demo."""
import ctypes
import gzip
import json
import os
import random
import tempfile

import numpy as np
import pytest

import read_class_util as u
from manta_amd._capi import BatchOutput, read_class_options, small_sv_text
from oracle_lib import asm_opts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.loads(gzip.open(os.path.join(ROOT, "tests", "golden", "read_class_demo.json.gz")).read())


def demo_batch(cases):
    b = u.Batch()
    for c in cases:
        scans = [dict(s, records=[u.parse_record(l) for l in G["regions"][s["region"]]]) for s in c["scans"]]
        f = np.float32(c["chrom_depth"])
        b.add_locus(scans, is_max_depth=c["is_max_depth"], search_remote=False, max_depth=float(f * np.float32(12)),
                    max_local=float(f * np.float32(7)))
    return b


def demo_groups():
    """candidates grouped by the one option that differs between them"""
    groups = {}
    for c in G["cases"]:
        groups.setdefault(c["min_variant"], []).append(c)
    return groups


def test_restatement_reproduces_the_reference_piles_of_the_demo():
    assert len(G["cases"]) == 11
    for minvar, cases in demo_groups().items():
        o = u.run_oracle(demo_batch(cases), read_class_options(min_candidate_variant_size=minvar))
        for l, c in enumerate(cases):
            assert o["piles"][l] == c["pile"], c["name"]
            assert o["results"][l].status == 0
    assert sum(len(c["pile"]) for c in G["cases"]) > 200


def synthetic_bam_case(rb, tmp, seed):
    """random alignments -> SAM -> BAM; the candidate through the reference and through the restatement"""
    rng = random.Random(seed)
    chrom0 = "".join(rng.choice("ACGT") for _ in range(9000))
    chroms = [("chrA", chrom0), ("chrB", "".join(rng.choice("ACGT") for _ in range(100000)))]
    fa = os.path.join(tmp, "g%d.fa" % seed)
    u.write_fasta(fa, chroms)
    n_bam, n_bp = rng.choice([1, 2, 3]), rng.choice([1, 2])
    bps = []
    for _ in range(n_bp):
        centre, half = rng.randrange(2500, 6000), rng.choice([10, 60, 150, 260])
        state = rng.choice([u.RIGHT_OPEN, u.LEFT_OPEN]) if n_bp == 2 else rng.choice([u.COMPLEX, u.RIGHT_OPEN, u.LEFT_OPEN, u.UNKNOWN])
        bps.append((0, centre - half, centre + half, state))
    bams, tumor, pool = [], [], ["shared%d" % k for k in range(12)]
    for bi in range(n_bam):
        recs = []
        for bp in bps:
            recs += u.random_scan(rng, rng.randrange(5, 120), bp[1], bp[2], bp[3], 0, chrom0, bi, False, True, False, pool)["records"]
        recs = [r for r in recs if r["pos"] >= 0 and r["mpos"] >= 0]
        recs = [recs[i] for i in sorted(range(len(recs)), key=lambda i: (recs[i]["pos"], i))]  # (a shadow stays behind its anchor)
        sam = os.path.join(tmp, "s%d_%d.sam" % (seed, bi))
        u.write_sam(sam, chroms, recs)
        rb.sam_to_bam(sam, sam[:-3] + "bam")
        bams.append(sam[:-3] + "bam")
        tumor.append(bi >= max(1, n_bam - 1) and n_bam > 1)
    depth, chrom_depth, cd = rng.random() < 0.6, rng.choice([0.3, 0.8, 2.0]), ""
    if depth:
        cd = os.path.join(tmp, "cd%d.txt" % seed)
        open(cd, "w").write("chrA\t%g\nchrB\t%g\n" % (chrom_depth, chrom_depth))
    minvar = rng.choice([10, 4, 30])
    ref = rb.pile(bams, tumor, fa, cd, minvar, False, bps[0], bps[1] if n_bp == 2 else None)
    rev = [False, False]
    if n_bp == 2 and bps[0][3] == bps[1][3]:  # SVCandidateAssemblyRefiner.cpp:1795-1808
        rev = [False, True] if bps[0][3] == u.RIGHT_OPEN else [True, False]
    scans = []
    for k, bp in enumerate(bps):
        sb, se = ctypes.c_int32(), ctypes.c_int32()
        u._oracle_lib().oracle_read_search_range(bp[1], bp[2], ctypes.byref(sb), ctypes.byref(se))
        roff, rseq = ref["ref%d" % (k + 1)]
        for bi in range(n_bam):
            scans.append(dict(records=rb.region_records(bams[bi], fa, 0, sb.value, se.value), bam_index=bi, is_tumor=tumor[bi],
                              is_locus_reversed=rev[k], first_of_breakend=(bi == 0), bp_begin=bp[1], bp_end=bp[2], bp_state=bp[3],
                              ref_begin=roff, ref_seq=rseq))
    b = u.Batch()
    f = np.float32(chrom_depth)
    b.add_locus(scans, is_max_depth=depth, search_remote=False, max_depth=float(f * np.float32(12)), max_local=float(f * np.float32(7)))
    return b, read_class_options(min_candidate_variant_size=minvar), ref["reads"]


@pytest.mark.skipif(not u.have_ref_bam(), reason="oracle/_ref/libmanta_ref_bam.so not built (reference sources unavailable)")
def test_restatement_matches_the_reference_on_synthetic_bam_files():
    rb = u.RefBam()
    n_reads = 0
    with tempfile.TemporaryDirectory(prefix="manta_rc_") as tmp:
        for seed in range(14):
            b, opt, want = synthetic_bam_case(rb, tmp, seed)
            assert u.run_oracle(b, opt)["piles"][0] == want, seed
            n_reads += len(want)
    assert n_reads > 100


def remote_mate_case(rb, tmp, seed):
    """A complex candidate with chimeric pairs whose mates lie elsewhere (a second chromosome, or > 10 kb away), written as BAM files:
    the unmodified reference (assembleComplexSVCandidate with isSearchRemoteInsertionReads: getBreakendReads + retrieveRemoteReads,
    SVCandidateAssembler.cpp:141-256,570-655) against kernel flags + manta_amd/host/read_gather.hpp's retrieval, with the
    reference's own BAM layer answering the region queries of both."""
    rng = random.Random(7000 + seed)
    chrom_a = "".join(rng.choice("ACGT") for _ in range(30000))
    chrom_b = "".join(rng.choice("ACGT") for _ in range(100000))
    chroms = [("chrA", chrom_a), ("chrB", chrom_b)]
    fa = os.path.join(tmp, "rg%d.fa" % seed)
    u.write_fasta(fa, chroms)
    n_bam = rng.choice([1, 2, 2, 3])
    centre, half = rng.randrange(12500, 17500), rng.choice([10, 60, 150])
    state = rng.choice([u.COMPLEX, u.COMPLEX, u.UNKNOWN, u.RIGHT_OPEN, u.LEFT_OPEN])
    bp = (0, centre - half, centre + half, state)
    sb, se = ctypes.c_int32(), ctypes.c_int32()
    u._oracle_lib().oracle_read_search_range(bp[1], bp[2], ctypes.byref(sb), ctypes.byref(se))
    clusters = [(1, rng.randrange(2000, 95000)) for _ in range(rng.choice([1, 2, 3]))]
    clusters.append((0, centre + rng.choice([-1, 1]) * rng.randrange(10300, 11500)))
    pool = ["shared%d" % k for k in range(12)]
    bams, tumor = [], []
    for bi in range(n_bam):
        recs = u.random_scan(rng, rng.randrange(5, 70), bp[1], bp[2], bp[3], 0, chrom_a, bi, False, True, False, pool)["records"]
        for k in range(rng.randrange(3, 45)):  # chimeric pairs: the local read
            rl, pos = rng.choice([50, 75, 100]), rng.randrange(sb.value - 60, se.value + 20)
            ctid, cpos = rng.choice(clusters)
            flag = 0x1 | rng.choice([0x40, 0x80]) | rng.choice([0, 0x10]) | rng.choice([0, 0x20])
            a = rng.randrange(0, 25)
            cigar = "%dM" % rl if a == 0 else rng.choice(["%dS%dM" % (a, rl - a), "%dM%dS" % (rl - a, a)])
            bases = "".join(chrom_a[pos + i] if rng.random() > 0.03 else rng.choice("ACGTN") for i in range(rl))
            name = rng.choice(pool) if rng.random() < 0.1 else "chim%d_%d" % (bi, k)
            recs.append(u.record_from_bases(name, flag, 0, pos, rng.choice([0, 14, 15, 20, 60, 60]), ctid, cpos + rng.randrange(0, 350), cigar, bases,
                                            [rng.choice([2, 4, 5, 20, 30, 38]) for _ in range(rl)], 1 if rng.random() < 0.1 else 0))
        mates = []
        for r in recs:  # ... and, mostly, the mate where the local read says it is
            far = (r["flag"] & 0x1) and not (r["flag"] & 0x8) and (r["mtid"] != 0 or abs(r["pos"] - r["mpos"]) >= 9000)
            if not far or rng.random() < 0.12:
                continue
            rl = rng.choice([50, 75, 100])
            other = 0x40 if (r["flag"] & 0x80) and not (r["flag"] & 0x40) else 0x80
            if rng.random() < 0.08:
                other ^= 0xc0  # a record of the same name and the WRONG read number
            flag = 0x1 | other | rng.choice([0, 0x10]) | rng.choice([0, 0x20])
            src = chroms[r["mtid"]][1]
            bases = "".join(src[r["mpos"] + i] if rng.random() > 0.05 else rng.choice("ACGTN") for i in range(rl))
            quals = [rng.choice([2, 4, 5, 20, 30, 38]) for _ in range(rl)]
            mapq = 0 if rng.random() < 0.75 else rng.choice([0, 1, 30])
            mpos = r["mpos"] + (rng.choice([-3, 1, 2]) if rng.random() < 0.06 else 0)  # (a wrong mate position: before / behind the query)
            m = u.record_from_bases(r["qname"], flag, r["mtid"], mpos, mapq, 0, r["pos"], "%dM" % rl, bases, quals)
            if rng.random() < 0.1:  # a supplementary / secondary copy in front of it
                mates.append(dict(m, flag=flag | rng.choice([0x800, 0x100]), sa=1))
            mates.append(m)
            if rng.random() < 0.1:  # the same read twice: only the first is taken
                mates.append(dict(m, mapq=0))
        for ctid, cpos in clusters:  # bystanders in the remote regions
            for _ in range(rng.randrange(0, 12)):
                pos = cpos + rng.randrange(-80, 420)
                mates.append(u.record_from_bases("by%d" % rng.randrange(10 ** 6), 0x1 | 0x40, ctid, pos, rng.choice([0, 30]), ctid, pos + 200, "50M",
                                                 chroms[ctid][1][pos:pos + 50], [30] * 50))
        recs = [r for r in recs + mates if r["pos"] >= 0 and r["mpos"] >= 0]
        recs = [recs[i] for i in sorted(range(len(recs)), key=lambda i: (recs[i]["tid"], recs[i]["pos"], i))]
        sam = os.path.join(tmp, "rg%d_%d.sam" % (seed, bi))
        u.write_sam(sam, chroms, recs)
        rb.sam_to_bam(sam, sam[:-3] + "bam")
        bams.append(sam[:-3] + "bam")
        tumor.append(bi >= max(1, n_bam - 1) and n_bam > 1)
    depth, chrom_depth, cd = rng.random() < 0.6, rng.choice([0.4, 3.0, 3.0, 20.0]), ""
    if depth:
        cd = os.path.join(tmp, "rgcd%d.txt" % seed)
        open(cd, "w").write("chrA\t%g\nchrB\t%g\n" % (chrom_depth, chrom_depth))
    ref = rb.pile(bams, tumor, fa, cd, 10, True, bp)
    roff, rseq = ref["ref1"]
    scans = [dict(records=rb.region_records(bams[bi], fa, 0, sb.value, se.value), bam_index=bi, is_tumor=tumor[bi], is_locus_reversed=False,
                  first_of_breakend=(bi == 0), bp_begin=bp[1], bp_end=bp[2], bp_state=bp[3], ref_begin=roff, ref_seq=rseq) for bi in range(n_bam)]
    f = np.float32(chrom_depth)
    cand = dict(scans=scans, is_max_depth=depth, search_remote=True, max_depth=float(f * np.float32(12)), max_local=float(f * np.float32(7)))
    return cand, ref, (lambda bam_index, tid, begin, end: rb.region_records(bams[bam_index], fa, tid, begin, end))


def check_remote_mates(gather, n_cases):
    rb = u.RefBam()
    fetched, tripped = 0, 0
    with tempfile.TemporaryDirectory(prefix="manta_rg_") as tmp:
        for seed in range(n_cases):
            cand, ref, fetch = remote_mate_case(rb, tmp, seed)
            out, stats = gather.gather([cand], read_class_options(), fetch)
            assert out[0]["status"] == 0 and out[0]["pile"] == ref["reads"], seed
            assert out[0]["cache"] == ref["remote"], seed
            assert stats["inserted"] >= len(ref["remote"])
            fetched += stats["inserted"]
            tripped += stats["targets"] == 0
    return fetched, tripped


def check_random(lib, seeds, n_loci=4, reads_per_scan=(5, 160)):
    n = 0
    for seed in seeds:
        b = u.random_batch(seed, n_loci=n_loci, reads_per_scan=reads_per_scan, eq_rate=0.02 if seed % 3 == 0 else 0.0)
        opt = read_class_options(max_reads=[0, 7, 25][seed % 3], min_candidate_variant_size=[10, 4, 30][seed % 3],
                                 use_overlap_pair_evidence=seed % 2)
        u.same(u.run_product(lib, b, opt), u.run_oracle(b, opt), len(b.loci))
        n += len(b.reads)
    return n


def check_demo(lib):
    for minvar, cases in demo_groups().items():
        p = u.run_product(lib, demo_batch(cases), read_class_options(min_candidate_variant_size=minvar), strict=True)
        for l, c in enumerate(cases):
            assert p["piles_text"][l] == c["pile"], c["name"]


@pytest.fixture(scope="module")
def gather_emu(emu):
    return u.GatherLib(os.path.join(ROOT, "tests", "emu"), "manta_amd_emu", "emu")


@pytest.fixture(scope="module")
def gather_gpu(gpu):
    return u.GatherLib(os.path.join(ROOT, "manta_amd"), "manta_amd", "gpu")


@pytest.mark.skipif(not u.have_ref_bam(), reason="oracle/_ref/libmanta_ref_bam.so not built (reference sources unavailable)")
def test_emulated_remote_mates_match_the_reference(gather_emu):
    fetched, tripped = check_remote_mates(gather_emu, 16)
    assert fetched > 60 and tripped < 10


def check_remote_mate_goldens(gather):
    """tests/golden/remote_mate_cases.json.gz (make_remote_mate_golden.py): the same candidates with the stored answers of the region
    queries; piles and RemoteReadCache of the unmodified reference"""
    RG = json.loads(gzip.open(os.path.join(ROOT, "tests", "golden", "remote_mate_cases.json.gz")).read())
    n = 0
    for c in RG["cases"]:
        scans = [dict({k: v for k, v in s.items() if k != "lines"}, records=[u.parse_record(l) for l in s["lines"]]) for s in c["scans"]]
        cand = dict(scans=scans, is_max_depth=c["is_max_depth"], search_remote=True, max_depth=c["max_depth"], max_local=c["max_local"])
        asked = []

        def fetch(bam_index, tid, begin, end):
            key = "%d:%d:%d-%d" % (bam_index, tid, begin, end)
            asked.append(key)
            return [u.parse_record(l) for l in c["remote_regions"][key]]  # (a query the reference would not make is a KeyError)
        out, stats = gather.gather([cand], read_class_options(), fetch)
        assert out[0]["status"] == 0 and out[0]["pile"] == c["ref_pile"] and out[0]["cache"] == c["ref_cache"], c["seed"]
        assert asked == list(c["remote_regions"]) or sorted(asked) == sorted(c["remote_regions"]), c["seed"]
        n += stats["inserted"]
    return n


def test_emulated_remote_mate_goldens(gather_emu):
    assert check_remote_mate_goldens(gather_emu) == 104


@pytest.mark.gpu
def test_gpu_remote_mate_goldens(gather_gpu):
    assert check_remote_mate_goldens(gather_gpu) == 104


def test_emulated_kernel_matches_restatement_on_random_records(emu):
    assert check_random(emu, range(100, 108)) > 5000


def test_emulated_kernel_reproduces_the_reference_piles_of_the_demo(emu):
    check_demo(emu)


def test_emulated_empty_and_degenerate_batches(emu):
    b = u.Batch()
    b.add_locus([])  # a candidate without a region query
    b.add_locus([dict(records=[], bam_index=0, is_tumor=False, is_locus_reversed=False, first_of_breakend=True, bp_begin=100, bp_end=130,
                      bp_state=u.COMPLEX, ref_begin=0, ref_seq="ACGT" * 100)])
    z = u.record_from_bases("empty", 0x1, 0, 120, 60, 0, 300, "*", "", [])  # a record without sequence
    b.add_locus([dict(records=[z], bam_index=0, is_tumor=False, is_locus_reversed=True, first_of_breakend=True, bp_begin=100, bp_end=130,
                      bp_state=u.COMPLEX, ref_begin=0, ref_seq="ACGT" * 100)], is_max_depth=True, max_depth=1.0, max_local=1.0)
    u.same(u.run_product(emu, b), u.run_oracle(b), 3)


def test_emulated_read_piles_argument_and_capacity_errors(emu):
    """the ABI's error behaviour: records that point outside their arenas are refused before anything runs; pile arrays that are
    too small come back MANTA_E_CAPACITY with the sizes that would have been needed"""
    import ctypes
    from manta_amd._capi import MantaError
    b = u.random_batch(5, n_loci=2, reads_per_scan=(20, 40))
    loci, scans, reads, cigars, names, seqs, quals, refs = b.arrays()
    good = u._call(emu, read_class_options(), b, loci, scans, reads, cigars, names, seqs, quals, refs, False)
    assert len(good["pile_read"]) > 0
    # search range: the breakend interval widened to 400 bases (SVCandidateAssembler.cpp:285-303), left alone when it is wider
    sb, se = ctypes.c_int32(), ctypes.c_int32()
    emu.lib.manta_read_search_range(1000, 1100, ctypes.byref(sb), ctypes.byref(se))
    assert (sb.value, se.value) == (850, 1250)
    emu.lib.manta_read_search_range(1000, 1500, ctypes.byref(sb), ctypes.byref(se))
    assert (sb.value, se.value) == (1000, 1500)
    # a record whose qualities lie outside the arena
    keep = reads[0].qual_off
    reads[0].qual_off = len(quals) + 5
    with pytest.raises(MantaError) as e:
        u._call(emu, read_class_options(), b, loci, scans, reads, cigars, names, seqs, quals, refs, True)
    assert e.value.code == -1 and "outside the arenas" in str(e.value)
    reads[0].qual_off = keep
    # a region query that names records beyond the array
    keep = scans[0].read_end
    scans[0].read_end = len(b.reads) + 3
    with pytest.raises(MantaError) as e:
        u._call(emu, read_class_options(), b, loci, scans, reads, cigars, names, seqs, quals, refs, True)
    assert e.value.code == -1
    scans[0].read_end = keep
    # pile arrays too small: MANTA_E_CAPACITY, *_used say what is needed
    f = emu.lib.manta_read_piles_batch
    n = len(b.reads)
    dec, pidx = (ctypes.c_uint8 * n)(), (ctypes.c_uint32 * n)()
    from manta_amd._capi import ReadLocusResult
    res = (ReadLocusResult * 2)()
    used = (ctypes.c_uint64 * 3)()
    small = (ctypes.c_uint32 * 4)()
    off = (ctypes.c_uint64 * (n + 2))()
    begin = (ctypes.c_uint32 * 3)()
    opt = read_class_options()
    rc = f(emu.ctx, ctypes.byref(opt), 2, ctypes.cast(loci, ctypes.c_void_p), len(b.scans), ctypes.cast(scans, ctypes.c_void_p), n,
           ctypes.cast(reads, ctypes.c_void_p), cigars.ctypes.data, len(cigars), names.ctypes.data, len(names), seqs.ctypes.data, len(seqs),
           quals.ctypes.data, len(quals), refs.ctypes.data, len(refs), dec, pidx, ctypes.cast(res, ctypes.c_void_p), small, 1,
           ctypes.byref(used, 0), small, 1, ctypes.byref(used, 8), small, off, off, small, 1, ctypes.byref(used, 16), begin)
    assert rc == -6
    assert used[2] == len(good["pile_read"]) and used[0] == len(good["piles"].codes) and used[1] == len(good["piles"].nmask)
    assert bytes(dec) == good["decision"].tobytes()  # (decisions and per-candidate results are complete all the same)


def check_piles_feed_the_pipeline(lib, oracle):
    """the piles the kernel builds from the demo's BAM records go straight into the small-SV pipeline (manta_smallsv_batch_piles)
    and give what the oracle gives on the reference's pile text"""
    cases = [c for c in G["cases"] if c["min_variant"] == 10 and len(c["scans"]) == 2 and c["pile"]]
    p = u.run_product(lib, demo_batch(cases), strict=True)
    n = len(cases)
    refs, cuts = [], []
    for c in cases:
        ref = c["scans"][0]["ref_seq"]
        mid = len(ref) // 2
        refs.append(ref[mid - 300:mid + 300].encode())
        cuts.append((100, 100, 200, 200))
    ref_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in refs], out=ref_off[1:])
    refs_np = np.frombuffer(b"".join(refs) + b"\0" * 64, dtype=np.uint8)
    cuts_np = np.ascontiguousarray(np.array(cuts, dtype=np.int32))
    opts, sc = asm_opts(minWordLength=41, maxWordLength=76, wordStepSize=5), [2, -8, -24, -1, -1, 0]
    out = BatchOutput(lib, "smallsv", n, 10, 1 << 20, 1 << 16, 1 << 18)
    lib.smallsv_batch_piles(opts, sc, -100, p["piles"], refs_np, ref_off, cuts_np, out)
    res = out.decode(np.diff(p["piles"].begin))
    for l, c in enumerate(cases):
        want = oracle.small_sv_locus(opts, sc, -100, [r.encode() for r in c["pile"]], refs[l], cuts[l])
        assert small_sv_text(res[l]) == want, c["name"]


def test_emulated_piles_feed_the_pipeline(emu, oracle):
    check_piles_feed_the_pipeline(emu, oracle)


@pytest.mark.gpu
def test_gpu_kernel_matches_restatement_on_random_records(gpu):
    assert check_random(gpu, range(100, 124)) > 15000
    assert check_random(gpu, range(300, 303), n_loci=40, reads_per_scan=(100, 900)) > 100000


@pytest.mark.gpu
def test_gpu_kernel_reproduces_the_reference_piles_of_the_demo(gpu):
    check_demo(gpu)


@pytest.mark.gpu
def test_gpu_piles_feed_the_pipeline(gpu, oracle):
    check_piles_feed_the_pipeline(gpu, oracle)
