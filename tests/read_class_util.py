"""Helpers of the read-gathering tests (tests/test_read_class.py): decoded BAM records as text lines (the format of
oracle/ref_bam_driver.cpp::ref_region_records) -> the arrays of manta_read_piles_batch; the restatement (oracle/read_class_oracle.cpp)
and the reference (oracle/_ref/libmanta_ref_bam.so) behind the same batches."""
import ctypes
import os
import random

import numpy as np

from manta_amd._capi import BamRead, ReadLocus, ReadLocusResult, ReadScan, read_class_options

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = "MIDNSHP=XB"
UNKNOWN, RIGHT_OPEN, LEFT_OPEN, COMPLEX = 0, 1, 2, 3


def cigar_words(text, for_mate=False):
    """CIGAR text -> BAM words (length << 4 | op).  for_mate: as cigar_to_apath reads the MC tag (P and zero lengths dropped)"""
    if text in ("*", ""):
        return []
    out, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            n, op = int(num), OPS.index(ch)
            num = ""
            if for_mate and (op == 6 or n == 0):
                continue
            out.append((n << 4) | op)
    return out


def parse_record(line):
    f = line.split()
    qname, flag, tid, pos, mapq, mtid, mpos, cigar, seq4, qual, sa, mc = f
    seq = bytes.fromhex(seq4) if seq4 != "*" else b""
    q = bytes.fromhex(qual) if qual != "*" else b""
    return dict(qname=qname, flag=int(flag), tid=int(tid), pos=int(pos), mapq=int(mapq), mtid=int(mtid), mpos=int(mpos), cigar=cigar,
                seq4=seq, qual=q, read_len=len(q), sa=int(sa), mc=mc)


def record_from_bases(qname, flag, tid, pos, mapq, mtid, mpos, cigar, bases, quals, sa=0, mc="*"):
    """bases: text over =ACGTN (and IUPAC letters, which BAM stores as their own codes)"""
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    nib = [code[c] for c in bases] + [0]
    seq4 = bytes((nib[2 * i] << 4) | nib[2 * i + 1] for i in range((len(bases) + 1) // 2))
    return dict(qname=qname, flag=flag, tid=tid, pos=pos, mapq=mapq, mtid=mtid, mpos=mpos, cigar=cigar, seq4=seq4, qual=bytes(quals),
                read_len=len(bases), sa=sa, mc=mc)


class Batch:
    def __init__(self):
        self.loci, self.scans, self.reads = [], [], []
        self.cigars, self.names, self.seqs, self.quals, self.refs = [], bytearray(), bytearray(), bytearray(), bytearray()

    def add_locus(self, scans, is_max_depth=False, search_remote=False, max_depth=0.0, max_local=0.0):
        """scans: dicts(records, bam_index, is_tumor, is_locus_reversed, first_of_breakend, bp_begin, bp_end, bp_state, ref_begin, ref_seq)"""
        lo = ReadLocus()
        lo.scan_begin = len(self.scans)
        for sc in scans:
            s = ReadScan()
            s.read_begin = len(self.reads)
            for r in sc["records"]:
                b = BamRead()
                b.tid, b.pos, b.mate_tid, b.mate_pos, b.flag, b.mapq = r["tid"], r["pos"], r["mtid"], r["mpos"], r["flag"], r["mapq"]
                b.tags = (1 if r["sa"] else 0) | (2 if r["mc"] != "*" else 0)
                b.read_len = r["read_len"]
                cw = cigar_words(r["cigar"])
                b.n_cigar, b.cigar_off = len(cw), len(self.cigars)
                self.cigars += cw
                mw = cigar_words(r["mc"], for_mate=True) if r["mc"] != "*" else []
                b.n_mate_cigar, b.mate_cigar_off = len(mw), len(self.cigars)
                self.cigars += mw
                qn = r["qname"].encode()
                b.qname_len, b.qname_off = len(qn), len(self.names)
                self.names += qn
                b.seq_off = len(self.seqs)
                self.seqs += r["seq4"]
                b.qual_off = len(self.quals)
                self.quals += r["qual"]
                self.reads.append(b)
            s.read_end = len(self.reads)
            s.bam_index, s.is_tumor, s.is_locus_reversed = sc["bam_index"], int(sc["is_tumor"]), int(sc["is_locus_reversed"])
            s.first_of_breakend = int(sc["first_of_breakend"])
            s.bp_begin, s.bp_end, s.bp_state = sc["bp_begin"], sc["bp_end"], sc["bp_state"]
            s.ref_begin, s.ref_len, s.ref_off = sc["ref_begin"], len(sc["ref_seq"]), len(self.refs)
            self.refs += sc["ref_seq"].encode()
            self.scans.append(s)
        lo.scan_end = len(self.scans)
        lo.is_max_depth, lo.search_remote, lo.max_depth, lo.max_local_depth_remote = int(is_max_depth), int(search_remote), max_depth, max_local
        self.loci.append(lo)

    def arrays(self):
        def arr(t, items):
            a = (t * max(len(items), 1))()
            for i, x in enumerate(items):
                a[i] = x
            return a
        pad = lambda b: np.frombuffer(bytes(b) + b"\0" * 16, dtype=np.uint8)  # noqa: E731
        return (arr(ReadLocus, self.loci), arr(ReadScan, self.scans), arr(BamRead, self.reads),
                np.array(self.cigars + [0], dtype=np.uint32), pad(self.names), pad(self.seqs), pad(self.quals), pad(self.refs))


def _oracle_lib():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "libmanta_oracle.so"))


def run_oracle(batch, opt=None):
    """restatement -> dict(decision, pile_index, results, piles = [[read text] per candidate])"""
    lib = _oracle_lib()
    opt = opt or read_class_options()
    loci, scans, reads, cigars, names, seqs, quals, refs = batch.arrays()
    n_loci, n_reads = len(batch.loci), len(batch.reads)
    decision = np.zeros(max(n_reads, 1), dtype=np.uint8)
    pile_index = np.zeros(max(n_reads, 1), dtype=np.uint32)
    results = (ReadLocusResult * max(n_loci, 1))()
    cap = sum(int(r.read_len) + 1 for r in batch.reads) + 16
    text = ctypes.create_string_buffer(cap)
    used = ctypes.c_uint64()
    lib.oracle_read_piles.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 11 + [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    lib.oracle_read_piles(ctypes.byref(opt), n_loci, ctypes.cast(loci, ctypes.c_void_p), ctypes.cast(scans, ctypes.c_void_p),
                          ctypes.cast(reads, ctypes.c_void_p), cigars.ctypes.data, names.ctypes.data, seqs.ctypes.data, quals.ctypes.data,
                          refs.ctypes.data, decision.ctypes.data, pile_index.ctypes.data, ctypes.cast(results, ctypes.c_void_p), text, cap,
                          ctypes.byref(used))
    lines = text.value.decode().split("\n")[:-1] if used.value else []
    piles, at = [], 0
    for l in range(n_loci):
        n = results[l].n_pile_reads
        piles.append(lines[at:at + n])
        at += n
    return dict(decision=decision[:n_reads], pile_index=pile_index[:n_reads], results=[results[i] for i in range(n_loci)], piles=piles)


def run_product(lib, batch, opt=None, strict=False):
    opt = opt or read_class_options()
    loci, scans, reads, cigars, names, seqs, quals, refs = batch.arrays()
    out = _call(lib, opt, batch, loci, scans, reads, cigars, names, seqs, quals, refs, strict)
    out["piles_text"] = piles_text(out["piles"], len(batch.loci))
    return out


def _call(lib, opt, batch, loci, scans, reads, cigars, names, seqs, quals, refs, strict):
    # (the ctypes arrays are padded to one element when empty: hand over the true counts)
    L = (ReadLocus * len(batch.loci)).from_buffer(loci) if batch.loci else (ReadLocus * 0)()
    S = (ReadScan * len(batch.scans)).from_buffer(scans) if batch.scans else (ReadScan * 0)()
    R = (BamRead * len(batch.reads)).from_buffer(reads) if batch.reads else (BamRead * 0)()
    return lib.read_piles_batch(opt, L, S, R, cigars, names, seqs, quals, refs, strict=strict)


def piles_text(p, n_loci):
    """PackedPiles -> [[read text] per candidate]"""
    out = []
    for l in range(n_loci):
        reads = []
        for r in range(int(p.begin[l]), int(p.begin[l + 1])):
            n, co, mo = int(p.read_len[r]), int(p.code_off[r]), int(p.mask_off[r])
            s = []
            for i in range(n):
                if (int(p.nmask[mo + (i >> 5)]) >> (i & 31)) & 1:
                    s.append("N")
                else:
                    s.append("ACGT"[(int(p.codes[co + (i >> 4)]) >> (30 - 2 * (i & 15))) & 3])
            reads.append("".join(s))
        out.append(reads)
    return out


def same(a, b, n_loci):
    """product / restatement outputs equal: decisions, pile positions, per-candidate results, pile text"""
    assert np.array_equal(a["decision"], b["decision"]), np.nonzero(a["decision"] != b["decision"])[0][:10]
    assert np.array_equal(a["pile_index"], b["pile_index"])
    for l in range(n_loci):
        ra, rb = a["results"][l], b["results"][l]
        assert (ra.status, ra.n_pile_reads, ra.retrieve_remote) == (rb.status, rb.n_pile_reads, rb.retrieve_remote), l
        if ra.status == 0:
            assert a["piles_text"][l] == b["piles"][l], l


# ---------------------------------------------------------------------------------------------------------------
# random records (sorted by position, as a region query returns them) that reach every branch of the scan
# ---------------------------------------------------------------------------------------------------------------
def random_scan(rng, n, bp_begin, bp_end, bp_state, ref_begin, ref_seq, bam_index, is_tumor, first, reversed_, name_pool, eq_rate=0.0):
    recs = []
    sb = bp_begin - max(0, (400 - (bp_end - bp_begin)) // 2)
    se = bp_end + max(0, (400 - (bp_end - bp_begin)) // 2)
    positions = sorted(rng.randrange(sb - 120, se + 40) for _ in range(n))
    i = 0
    while i < len(positions):
        pos = positions[i]
        rl = rng.choice([36, 50, 75, 100])
        kind = rng.random()
        flag = 0x1 | (0x40 if rng.random() < 0.5 else 0x80)
        if rng.random() < 0.5:
            flag |= 0x10
        if rng.random() < 0.5:
            flag |= 0x20
        # cigar shapes: plain, soft clipped on either end, insertion / deletion, hard clips, '=' / 'X' operations
        shape = rng.random()
        if shape < 0.35:
            cigar = "%dM" % rl
        elif shape < 0.55:
            a = rng.randrange(1, 30)
            cigar = "%dS%dM" % (a, rl - a) if rng.random() < 0.5 else "%dM%dS" % (rl - a, a)
        elif shape < 0.65:
            a, b = rng.randrange(1, 15), rng.randrange(1, 15)
            cigar = "%dS%dM%dS" % (a, rl - a - b, b)
        elif shape < 0.8:
            k, a = rng.randrange(1, 12), rng.randrange(5, rl - 20)
            cigar = "%dM%dI%dM" % (a, k, rl - a - k) if rng.random() < 0.5 else "%dM%dD%dM" % (a, k, rl - a)
        elif shape < 0.88:
            a = rng.randrange(5, rl - 10)
            cigar = "3H%d=%dX%dM2H" % (a, 2, rl - a - 2)
        elif shape < 0.94:
            a = rng.randrange(5, rl - 25)
            cigar = "%dM2I%dM7D%dM" % (a, 8, rl - a - 10)
        else:
            a = rng.randrange(2, 10)
            cigar = "%dS%dM30N%dM" % (a, 20, rl - a - 20)
        # bases: the reference under the alignment with a few substitutions, random under the clips
        bases = []
        rp = pos
        for w in cigar_words(cigar):
            ln, op = w >> 4, w & 15
            for _ in range(ln if op in (0, 1, 4, 7, 8) else 0):
                if op in (0, 7, 8) and ref_begin <= rp < ref_begin + len(ref_seq) and rng.random() > 0.04:
                    bases.append(ref_seq[rp - ref_begin])
                else:
                    bases.append(rng.choice("ACGT"))
                if op in (0, 7, 8):
                    rp += 1
            if op in (2, 3):
                rp += ln
        bases = "".join(b if rng.random() > 0.01 else "N" for b in bases)
        if eq_rate and rng.random() < eq_rate:
            k = rng.randrange(len(bases))
            bases = bases[:k] + "=" + bases[k + 1:]
        quals = [rng.choice([2, 4, 5, 12, 19, 20, 21, 30, 38]) for _ in range(rl)]
        mapq = rng.choice([0, 3, 14, 15, 16, 40, 60])
        mtid, mpos = 0, pos + rng.randrange(-400, 600)
        sa, mc = 0, "*"
        if kind < 0.06:
            flag |= 0x400  # duplicate
        elif kind < 0.10:
            flag |= 0x200  # QC fail
        elif kind < 0.16:
            flag |= 0x800  # supplementary
            sa = 1 if rng.random() < 0.6 else 0
        elif kind < 0.20:
            flag |= 0x100  # secondary
            sa = 1 if rng.random() < 0.5 else 0
        elif kind < 0.26:
            mtid, mpos = (1, rng.randrange(1000, 90000)) if rng.random() < 0.6 else (0, pos + rng.choice([-1, 1]) * rng.randrange(9990, 10020))
        elif kind < 0.30:
            flag &= ~0x1  # unpaired
        if rng.random() < 0.35:
            mc = rng.choice(["%dM" % rl, "5S%dM" % (rl - 5), "%dM6S" % (rl - 6), "2H10S%dM3P0M" % (rl - 10)])
        if rng.random() < 0.15:
            sa = 1
        if rng.random() < 0.3 and (flag & 0x1):  # close, inward pointing pair: overlap / adapter tests
            mpos = pos + rng.randrange(-rl, rl)
            flag = (flag & ~0x30) | (0x20 if rng.random() < 0.5 else 0x10)
        name = rng.choice(name_pool) if rng.random() < 0.25 else "q%d_%d" % (bam_index, len(recs) + rng.randrange(10 ** 6))
        rec = record_from_bases(name, flag, 0, pos, mapq, mtid, mpos, cigar, bases, quals, sa, mc)
        recs.append(rec)
        # singleton + its shadow right behind it (sometimes with the wrong name, low quality, or something in between)
        if rng.random() < 0.18:
            aflag = (flag | 0x1 | 0x8) & ~(0x4 | 0x100 | 0x800 | 0x400 | 0x200)
            recs[-1] = dict(rec, flag=aflag, sa=0)
            squals = [rng.choice([10, 24, 25, 26, 35]) for _ in range(rl)] if rng.random() < 0.5 else [30] * rl
            sname = name if rng.random() < 0.85 else name + "x"
            sflag = 0x1 | 0x4 | (0x80 if aflag & 0x40 else 0x40) | (0x20 if aflag & 0x10 else 0)
            if rng.random() < 0.1:
                sflag |= 0x8
            sbases = "".join(rng.choice("ACGTN") for _ in range(rl))
            if rng.random() < 0.2:
                recs.append(record_from_bases("mid%d" % len(recs), 0x1 | 0x400, 0, pos, 30, 0, pos, "%dM" % rl, sbases, squals))
            recs.append(record_from_bases(sname, sflag, 0, pos, 0, 0, pos, "*", sbases, squals))
        i += 1
    # (positions were drawn sorted; a shadow sits right behind its anchor at the same position)
    return dict(records=recs, bam_index=bam_index, is_tumor=is_tumor, is_locus_reversed=reversed_, first_of_breakend=first, bp_begin=bp_begin,
                bp_end=bp_end, bp_state=bp_state, ref_begin=ref_begin, ref_seq=ref_seq)


def random_batch(seed, n_loci=6, reads_per_scan=(5, 160), eq_rate=0.0, small_cap=False):
    rng = random.Random(seed)
    b = Batch()
    for _ in range(n_loci):
        n_bp = rng.choice([1, 2])
        n_bam = rng.choice([1, 2, 3])
        pool = ["shared%d" % k for k in range(12)]
        scans = []
        for bp in range(n_bp):
            centre = rng.randrange(2000, 5000)
            half = rng.choice([10, 60, 150, 260])
            ref_begin = centre - half - rng.randrange(150, 700)
            ref_seq = "".join(rng.choice("ACGT") for _ in range(2 * (centre - ref_begin)))
            if rng.random() < 0.3:
                k = rng.randrange(len(ref_seq) - 40)
                ref_seq = ref_seq[:k] + "N" * 30 + ref_seq[k + 30:]
            state = rng.choice([UNKNOWN, RIGHT_OPEN, LEFT_OPEN, COMPLEX])
            rev = rng.random() < 0.5
            for bi in range(n_bam):
                n = rng.randrange(*reads_per_scan)
                scans.append(random_scan(rng, n, centre - half, centre + half, state, ref_begin, ref_seq, bi, bi >= max(1, n_bam - 1) and n_bam > 1,
                                         bi == 0, rev, pool, eq_rate))
        depth = rng.random() < 0.6
        b.add_locus(scans, is_max_depth=depth, search_remote=rng.random() < 0.5, max_depth=float(rng.choice([3, 8, 20, 60])),
                    max_local=float(rng.choice([2, 5, 12])))
    return b


# ---------------------------------------------------------------------------------------------------------------
# the reference behind the same records: synthetic SAM -> BAM (htslib of the reference's redist) -> the real getBreakendReads
# ---------------------------------------------------------------------------------------------------------------
REF_BAM_LIB = os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_bam.so")


def have_ref_bam():
    return os.path.exists(REF_BAM_LIB)


class RefBam:
    def __init__(self):
        self.lib = ctypes.CDLL(REF_BAM_LIB)
        self.buf = ctypes.create_string_buffer(1 << 26)

    def sam_to_bam(self, sam, bam):
        rc = self.lib.ref_sam_to_bam(sam.encode(), bam.encode())
        assert rc == 0, rc

    def region_records(self, bam, fasta, tid, begin, end):
        self.lib.ref_region_records(bam.encode(), fasta.encode(), tid, begin, end, self.buf, len(self.buf))
        text = self.buf.value.decode()
        assert not text.startswith("EXCEPTION"), text[:300]
        return [parse_record(l) for l in text.splitlines()]

    def region_text(self, bam, fasta, tid, begin, end):
        self.lib.ref_region_records(bam.encode(), fasta.encode(), tid, begin, end, self.buf, len(self.buf))
        return self.buf.value.decode()

    def pile(self, bams, is_tumor, fasta, chrom_depth, min_variant, search_remote, bp1, bp2=None):
        """bp = (tid, begin, end, state).  -> dict(reads, ref1=(offset, seq), ref2=...)"""
        n = len(bams)
        b2 = bp2 or (0, 0, 0, -1)
        self.lib.ref_breakend_pile(n, (ctypes.c_char_p * n)(*[x.encode() for x in bams]), (ctypes.c_int * n)(*[int(t) for t in is_tumor]),
                                   fasta.encode(), chrom_depth.encode(), min_variant, int(search_remote), bp1[0], bp1[1], bp1[2], bp1[3],
                                   b2[0], b2[1], b2[2], b2[3], self.buf, len(self.buf))
        lines = self.buf.value.decode().splitlines()
        assert lines and lines[0].startswith("reads "), lines[:1]
        n_reads = int(lines[0].split()[1])
        out = dict(reads=lines[1:1 + n_reads], remote=[])
        for l in lines[1 + n_reads:]:
            f = l.split(" ")
            if f[0] in ("ref1", "ref2"):
                out[f[0]] = (int(f[1]), f[2] if len(f) > 2 else "")
            elif f[0] == "remote":  # the RemoteReadCache, by name: qname, read number, read
                out["remote"].append(" ".join(f[1:]))
        return out


class GatherLib:
    """tests/cpp/host_gather_capi.cpp: manta_amd/host/read_gather.hpp (ReadGatherBatch + remote-mate retrieval) for Python"""
    SCAN_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32)

    def __init__(self, lib_dir, lib_name, tag):
        import subprocess
        cpp = os.path.join(ROOT, "tests", "cpp")
        so = os.path.join(cpp, "libhost_gather_%s.so" % tag)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "manta_amd", "host"), os.path.join(cpp, "host_gather_capi.cpp"), "-o", so,
                               "-L" + lib_dir, "-l" + lib_name, "-Wl,-rpath," + lib_dir])
        self.lib = ctypes.CDLL(so)
        self.lib.rg_new.restype = ctypes.c_void_p
        self.lib.rg_counts.restype = ctypes.c_ulonglong
        self.buf = ctypes.create_string_buffer(1 << 24)

    def gather(self, candidates, opt, fetch):
        """candidates: [dict(scans, is_max_depth, search_remote, max_depth, max_local)] with scans as Batch.add_locus takes them;
        fetch(bam_index, tid, begin, end) -> records of that region query.  -> [dict(status, pile, cache)], dict(queries, targets, inserted)"""
        L, c = self.lib, ctypes
        h = c.c_void_p(L.rg_new())
        try:
            for cand in candidates:
                L.rg_begin_candidate(h, int(cand["is_max_depth"]), c.c_float(cand["max_depth"]), c.c_float(cand["max_local"]), int(cand["search_remote"]))
                for s in cand["scans"]:
                    L.rg_begin_query(h, s["bp_begin"], s["bp_end"], s["bp_state"], int(s["is_locus_reversed"]), s["bam_index"], int(s["is_tumor"]),
                                     int(s["first_of_breakend"]), s["ref_begin"], s["ref_seq"].encode())
                    for r in s["records"]:
                        cw = cigar_words(r["cigar"])
                        L.rg_add_record(h, r["tid"], r["pos"], r["mtid"], r["mpos"], r["flag"], r["mapq"], (c.c_uint32 * max(1, len(cw)))(*cw), len(cw),
                                        r["qname"].encode(), bytes(r["seq4"]) + b"\0", bytes(r["qual"]) + b"\0", r["read_len"], int(r["sa"]),
                                        None if r["mc"] == "*" else r["mc"].encode())
            assert L.rg_run(h, c.byref(opt)) == 0, self._err(h)

            def on_scan(_user, bam_index, tid, begin, end):
                for r in fetch(bam_index, tid, begin, end):
                    if not L.rg_remote_record(h, r["pos"], r["flag"], r["mapq"], r["qname"].encode(), bytes(r["seq4"]) + b"\0", bytes(r["qual"]) + b"\0",
                                              r["read_len"], int(r["sa"])):
                        break
            cb = self.SCAN_CB(on_scan)
            nq = L.rg_retrieve_remote(h, c.byref(opt), cb, None)
            assert nq >= 0, self._err(h)
            out = []
            for l in range(len(candidates)):
                assert L.rg_pile_text(h, l, self.buf, len(self.buf)) >= 0
                lines = self.buf.value.decode().splitlines()
                head = lines[0].split()
                assert int(head[3]) == len(lines) - 1
                assert L.rg_remote_cache(h, l, self.buf, len(self.buf)) >= 0
                out.append(dict(status=int(head[1]), pile=lines[1:], cache=self.buf.value.decode().splitlines()))
            return out, dict(queries=nq, targets=L.rg_counts(h, 0), inserted=L.rg_counts(h, 1))
        finally:
            L.rg_free(h)

    def _err(self, h):
        self.lib.rg_error(h, self.buf, len(self.buf))
        return self.buf.value.decode()


def write_fasta(path, chroms):
    """chroms: [(name, seq)]; writes path and path.fai"""
    off = 0
    with open(path, "w") as f, open(path + ".fai", "w") as fi:
        for name, seq in chroms:
            head = ">%s\n" % name
            f.write(head)
            off += len(head)
            fi.write("%s\t%d\t%d\t60\t61\n" % (name, len(seq), off))
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
            off += len(seq) + (len(seq) + 59) // 60


def write_sam(path, chroms, records):
    with open(path, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n")
        for name, seq in chroms:
            f.write("@SQ\tSN:%s\tLN:%d\n" % (name, len(seq)))
        for r in records:
            bases = "".join("=ACMGRSVTWYHKDBN"[(r["seq4"][i >> 1] >> (4 * (1 - (i & 1)))) & 15] for i in range(r["read_len"]))
            qual = "".join(chr(33 + q) for q in r["qual"])
            rnext = "=" if r["mtid"] == r["tid"] else chroms[r["mtid"]][0]
            tags = []
            if r["sa"]:
                tags.append("SA:Z:%s,100,+,50M,30,0;" % chroms[0][0])
            if r["mc"] != "*":
                tags.append("MC:Z:" + r["mc"])
            f.write("\t".join([r["qname"], str(r["flag"]), chroms[r["tid"]][0], str(r["pos"] + 1), str(r["mapq"]), r["cigar"], rnext,
                               str(r["mpos"] + 1), "0", bases or "*", qual or "*"] + tags) + "\n")
