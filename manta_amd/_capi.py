"""ctypes binding of include/manta_amd.h.

``Lib()`` loads ``manta_amd/libmanta_amd.so`` and fails loudly if it is missing or no GPU is usable.
(The test suite may pass an explicit ``path`` to exercise the same ABI on the wave emulator build under
tests/emu/ -- that library is test infrastructure and is never picked up implicitly.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

CIGAR_OPS = "MIDNSHP=X"

ALIGNER_GLOBAL, ALIGNER_LARGE_INDEL, ALIGNER_JUMP = 0, 1, 2


ABI_VERSION = 6  # include/manta_amd.h: MANTA_ABI_VERSION


class MantaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("manta_amd error %d: %s" % (code, msg))
        self.code = code


def default_library_path():
    # MANTA_AMD_LIB: a developer build of the same HIP library (profile counters, register budgets) that tools/ put next to
    # the product build -- only libraries under manta_amd/ are accepted; never a CPU path
    alt = os.environ.get("MANTA_AMD_LIB")
    if alt:
        alt = os.path.abspath(alt)
        if os.path.dirname(alt) not in (_HERE, os.path.join(_HERE, "variants")) or not os.path.basename(alt).startswith("libmanta_amd"):
            raise MantaError(-2, "MANTA_AMD_LIB must name a libmanta_amd*.so under manta_amd/ (got %s)" % alt)
        return alt
    return os.path.join(_HERE, "libmanta_amd.so")


class AlignScores(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("match", "mismatch", "open", "extend", "off_edge", "is_allow_edge_insertion")]


class AlignTask(ctypes.Structure):
    _fields_ = [("query_off", ctypes.c_uint64), ("ref1_off", ctypes.c_uint64), ("ref2_off", ctypes.c_uint64),
                ("query_len", ctypes.c_uint32), ("ref1_len", ctypes.c_uint32), ("ref2_len", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


class AlignResult(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("score", ctypes.c_int32), ("is_jumped", ctypes.c_int32),
                ("begin_pos1", ctypes.c_int32), ("begin_pos2", ctypes.c_int32), ("jump_insert_size", ctypes.c_uint32),
                ("jump_range", ctypes.c_uint32), ("cigar1_len", ctypes.c_uint32), ("cigar2_len", ctypes.c_uint32),
                ("cigar1_off", ctypes.c_uint64), ("cigar2_off", ctypes.c_uint64)]


class AsmOptions(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("min_word_length", "max_word_length", "word_step_size", "min_contig_length",
                                               "min_coverage", "min_conservative_coverage", "min_unused_reads",
                                               "min_support_reads", "max_assembly_count")]


class AsmContig(ctypes.Structure):
    _fields_ = [("seq_off", ctypes.c_uint64), ("support_off", ctypes.c_uint64), ("reject_off", ctypes.c_uint64),
                ("seq_len", ctypes.c_uint32), ("seed_read_count", ctypes.c_uint32), ("conservative_begin", ctypes.c_int32),
                ("conservative_end", ctypes.c_int32)]


class AsmLocusResult(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("n_contigs", ctypes.c_uint32), ("first_contig", ctypes.c_uint32),
                ("n_words", ctypes.c_uint32), ("n_pseudo", ctypes.c_uint32), ("final_word_length", ctypes.c_uint32),
                ("n_iterations", ctypes.c_uint32), ("cyclic_iterations", ctypes.c_uint32), ("pseudo_seq_off", ctypes.c_uint64),
                ("pseudo_len_off", ctypes.c_uint64)]


class RefCuts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("leading_cut", "trailing_cut", "max_leading_cut", "max_trailing_cut")]


class SmallSvAlignment(ctypes.Structure):
    _fields_ = [("adjusted_leading_cut", ctypes.c_int32), ("adjusted_trailing_cut", ctypes.c_int32), ("align", AlignResult)]


class SmallSvStats(ctypes.Structure):
    _fields_ = [("assemble_ms", ctypes.c_float), ("schedule_ms", ctypes.c_float), ("align_ms", ctypes.c_float),
                ("total_ms", ctypes.c_float), ("n_align_launches", ctypes.c_uint32), ("n_alignments", ctypes.c_uint32),
                ("ptr_matrix_bytes", ctypes.c_uint64), ("dp_cells", ctypes.c_uint64)]


def _bits_members(words):
    out = []
    for wi, w in enumerate(words):
        w = int(w)
        while w:
            b = (w & -w).bit_length() - 1
            out.append(wi * 64 + b)
            w &= w - 1
    return out


def cigar_string(packed):
    return "".join("%d%s" % (int(v) >> 4, CIGAR_OPS[int(v) & 15]) for v in packed)


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode("latin-1")


# ---- read-pile construction (manta_read_piles_batch) ----
class BamRead(ctypes.Structure):
    _fields_ = [("tid", ctypes.c_int32), ("pos", ctypes.c_int32), ("mate_tid", ctypes.c_int32), ("mate_pos", ctypes.c_int32),
                ("flag", ctypes.c_uint16), ("mapq", ctypes.c_uint8), ("tags", ctypes.c_uint8), ("read_len", ctypes.c_uint32),
                ("n_cigar", ctypes.c_uint32), ("cigar_off", ctypes.c_uint32), ("n_mate_cigar", ctypes.c_uint32),
                ("mate_cigar_off", ctypes.c_uint32), ("qname_len", ctypes.c_uint32), ("qname_off", ctypes.c_uint32),
                ("seq_off", ctypes.c_uint64), ("qual_off", ctypes.c_uint64)]


class ReadScan(ctypes.Structure):
    _fields_ = [("read_begin", ctypes.c_uint32), ("read_end", ctypes.c_uint32), ("bam_index", ctypes.c_uint32), ("is_tumor", ctypes.c_uint8),
                ("is_locus_reversed", ctypes.c_uint8), ("first_of_breakend", ctypes.c_uint8), ("reserved", ctypes.c_uint8),
                ("bp_begin", ctypes.c_int32), ("bp_end", ctypes.c_int32), ("bp_state", ctypes.c_int32), ("ref_begin", ctypes.c_int32),
                ("ref_len", ctypes.c_uint32), ("ref_off", ctypes.c_uint64)]


class ReadLocus(ctypes.Structure):
    _fields_ = [("scan_begin", ctypes.c_uint32), ("scan_end", ctypes.c_uint32), ("is_max_depth", ctypes.c_uint8),
                ("search_remote", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 2), ("max_depth", ctypes.c_float),
                ("max_local_depth_remote", ctypes.c_float)]


class ReadClassOptions(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("min_qval", "min_candidate_variant_size", "min_singleton_mapq_candidates", "min_mapq",
                                               "use_overlap_pair_evidence", "max_reads")]


class ReadLocusResult(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("n_pile_reads", ctypes.c_uint32), ("retrieve_remote", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


def read_class_options(**kw):
    """the reference's defaults (IterativeAssemblerOptions::minQval, ReadScannerOptions)"""
    o = ReadClassOptions(5, 10, 15, 15, 0, 0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Lib:
    def __init__(self, path=None, device=-1):
        self.path = path or default_library_path()
        if not os.path.exists(self.path):
            raise MantaError(-2, "HIP library %s is not built (run `python -c 'import __graft_entry__ as g; g.build()'`);"
                                 " this package has no CPU fallback" % self.path)
        self.lib = ctypes.CDLL(self.path)
        L = self.lib
        L.manta_last_error.restype = ctypes.c_char_p
        L.manta_last_error.argtypes = [ctypes.c_void_p]
        L.manta_ctx_device_name.restype = ctypes.c_char_p
        L.manta_ctx_device_name.argtypes = [ctypes.c_void_p]
        L.manta_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.manta_ctx_destroy.argtypes = [ctypes.c_void_p]
        # the ctypes mirrors below follow include/manta_amd.h of this ABI version: the library writes whole records
        L.manta_abi_version.restype = ctypes.c_uint32
        L.manta_batch_stats_size.restype = ctypes.c_uint64
        if L.manta_abi_version() != ABI_VERSION or L.manta_batch_stats_size() != ctypes.sizeof(BatchStats):
            raise MantaError(-2, "%s speaks ABI version %d (manta_batch_stats_t: %d bytes); this binding mirrors version %d (%d bytes): rebuild the library"
                             % (self.path, L.manta_abi_version(), L.manta_batch_stats_size(), ABI_VERSION, ctypes.sizeof(BatchStats)))
        self.ctx = ctypes.c_void_p()
        rc = L.manta_ctx_create(device, ctypes.byref(self.ctx))
        if rc != 0:
            raise MantaError(rc, L.manta_last_error(None).decode())

    def close(self):
        if self.ctx:
            self.lib.manta_ctx_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_name(self):
        return self.lib.manta_ctx_device_name(self.ctx).decode()

    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise MantaError(rc, self.lib.manta_last_error(self.ctx).decode())

    # ------------------------------------------------------------------ read piles
    def read_piles_batch(self, opt, loci, scans, reads, cigars, names, seqs, quals, refs, strict=True):
        """loci / scans / reads: ctypes arrays of ReadLocus / ReadScan / BamRead; the arenas: numpy (cigars uint32, the others uint8).
        -> dict(decision, pile_index, results, piles=PackedPiles, pile_read)"""
        n_loci, n_scans, n_reads = len(loci), len(scans), len(reads)
        decision = np.zeros(max(n_reads, 1), dtype=np.uint8)
        pile_index = np.zeros(max(n_reads, 1), dtype=np.uint32)
        results = (ReadLocusResult * max(n_loci, 1))()
        total_len = sum(int(r.read_len) for r in reads)
        codes = np.zeros(total_len // 16 + n_reads + 4, dtype=np.uint32)
        nmask = np.zeros(total_len // 32 + n_reads + 4, dtype=np.uint32)
        read_len = np.zeros(n_reads + 1, dtype=np.uint32)
        code_off = np.zeros(n_reads + 2, dtype=np.uint64)
        mask_off = np.zeros(n_reads + 2, dtype=np.uint64)
        pile_read = np.zeros(n_reads + 1, dtype=np.uint32)
        begin = np.zeros(n_loci + 1, dtype=np.uint32)
        used = (ctypes.c_uint64 * 3)()
        arenas = [np.ascontiguousarray(a) for a in (cigars, names, seqs, quals, refs)]
        f = self.lib.manta_read_piles_batch
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                      ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        rc = f(self.ctx, ctypes.byref(opt), n_loci, ctypes.cast(loci, ctypes.c_void_p), n_scans, ctypes.cast(scans, ctypes.c_void_p), n_reads,
               ctypes.cast(reads, ctypes.c_void_p), arenas[0].ctypes.data, len(arenas[0]), arenas[1].ctypes.data, len(arenas[1]),
               arenas[2].ctypes.data, len(arenas[2]), arenas[3].ctypes.data, len(arenas[3]), arenas[4].ctypes.data, len(arenas[4]),
               decision.ctypes.data, pile_index.ctypes.data, ctypes.cast(results, ctypes.c_void_p), codes.ctypes.data, len(codes),
               ctypes.byref(used, 0), nmask.ctypes.data, len(nmask), ctypes.byref(used, 8), read_len.ctypes.data, code_off.ctypes.data,
               mask_off.ctypes.data, pile_read.ctypes.data, n_reads, ctypes.byref(used, 16), begin.ctypes.data)
        self._check(rc, allow=() if strict else (-5, -7))
        r = int(used[2])
        piles = PackedPiles(codes[:int(used[0])], nmask[:int(used[1])], read_len[:r], code_off[:r + 1], mask_off[:r + 1], begin)
        return dict(decision=decision[:n_reads], pile_index=pile_index[:n_reads], results=[results[i] for i in range(n_loci)], piles=piles,
                    pile_read=pile_read[:r])

    # ------------------------------------------------------------------ aligners
    def align_batch(self, kind, scores, extra, problems, strict=True):
        """problems: list of (query, ref1[, ref2]) byte strings.  Returns list of dicts (per-task status kept)."""
        arena = bytearray()
        tasks = (AlignTask * max(1, len(problems)))()
        for i, pr in enumerate(problems):
            q, r1 = _b(pr[0]), _b(pr[1])
            r2 = _b(pr[2]) if len(pr) > 2 and pr[2] is not None else b""
            t = tasks[i]
            t.query_off, t.query_len = len(arena), len(q)
            arena += q
            t.ref1_off, t.ref1_len = len(arena), len(r1)
            arena += r1
            t.ref2_off, t.ref2_len = len(arena), len(r2)
            arena += r2
        arena_np = np.frombuffer(bytes(arena) + b"\0", dtype=np.uint8)
        res = (AlignResult * max(1, len(problems)))()
        cap = sum(2 * len(_b(p[0])) + 8 for p in problems) + 8
        cig = np.zeros(cap, dtype=np.uint32)
        used = ctypes.c_uint64(0)
        sc = AlignScores(*scores)
        rc = self.lib.manta_align_batch(self.ctx, kind, ctypes.byref(sc), extra, len(problems), tasks,
                                        arena_np.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(arena)), res,
                                        cig.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(cap), ctypes.byref(used))
        self._check(rc, allow=() if strict else (-4, -5))
        out = []
        for i in range(len(problems)):
            r = res[i]
            out.append(dict(status=r.status, score=r.score, is_jumped=r.is_jumped, begin1=r.begin_pos1, begin2=r.begin_pos2,
                            jump_insert_size=r.jump_insert_size, jump_range=r.jump_range,
                            cigar1=cigar_string(cig[r.cigar1_off:r.cigar1_off + r.cigar1_len]),
                            cigar2=cigar_string(cig[r.cigar2_off:r.cigar2_off + r.cigar2_len])))
        return out


def pack_loci(loci_reads):
    """list (per locus) of lists of read byte strings -> (bases uint8, read_off uint64, locus_read_begin uint32)"""
    flat = [_b(r) for reads in loci_reads for r in reads]
    lens = np.fromiter((len(r) for r in flat), dtype=np.uint64, count=len(flat))
    read_off = np.zeros(len(flat) + 1, dtype=np.uint64)
    np.cumsum(lens, out=read_off[1:])
    bases = np.frombuffer(b"".join(flat) + b"\0", dtype=np.uint8)
    begin = np.zeros(len(loci_reads) + 1, dtype=np.uint32)
    np.cumsum([len(r) for r in loci_reads], out=begin[1:])
    return bases, read_off, begin


def _assemble_batch(self, opts, loci_reads, strict=True):
    """opts: 9 values in manta_asm_options_t order.  Returns one dict per locus."""
    bases, read_off, begin = pack_loci(loci_reads)
    n_loci = len(loci_reads)
    o = AsmOptions(*opts)
    res = (AsmLocusResult * max(1, n_loci))()
    ccap = n_loci * o.max_assembly_count + 1
    contigs = (AsmContig * ccap)()
    seq_cap = int(read_off[-1]) * 2 + 65536 * max(1, n_loci)
    seq = np.zeros(seq_cap, dtype=np.uint8)
    bits_cap = n_loci * (o.max_assembly_count * 2 * 16 + 64) + 64
    bits = np.zeros(bits_cap, dtype=np.uint64)
    su, bu = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = self.lib.manta_assemble_batch(self.ctx, ctypes.byref(o), n_loci, bases.ctypes.data_as(ctypes.c_void_p),
                                       read_off.ctypes.data_as(ctypes.c_void_p), begin.ctypes.data_as(ctypes.c_void_p), res,
                                       contigs, ctypes.c_uint64(ccap), seq.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.c_uint64(seq_cap), ctypes.byref(su), bits.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.c_uint64(bits_cap), ctypes.byref(bu))
    self._check(rc, allow=() if strict else (-5, -6, -7))
    out = []
    for l in range(n_loci):
        r = res[l]
        d = dict(status=r.status, n_reads=len(loci_reads[l]), n_words=r.n_words, final_word_length=r.final_word_length,
                 n_iterations=r.n_iterations, cyclic_iterations=r.cyclic_iterations, contigs=[], pseudo=[])
        if r.status == 0:
            for c in range(r.n_contigs):
                cc = contigs[r.first_contig + c]
                d["contigs"].append(dict(seq=seq[cc.seq_off:cc.seq_off + cc.seq_len].tobytes().decode("latin-1"),
                                         seed=cc.seed_read_count, cons=(cc.conservative_begin, cc.conservative_end),
                                         support=_bits_members(bits[cc.support_off:cc.support_off + r.n_words]),
                                         reject=_bits_members(bits[cc.reject_off:cc.reject_off + r.n_words])))
            off = int(r.pseudo_seq_off)
            for p in range(r.n_pseudo):
                ln = int(bits[r.pseudo_len_off + p])
                d["pseudo"].append(seq[off:off + ln].tobytes().decode("latin-1"))
                off += ln
        out.append(d)
    return out


Lib.assemble_batch = _assemble_batch


def _debug_repeat_words(self, opts, reads):
    """manta_debug_repeat_words: the repeat-word set (getRepeatKmers) the device computed for one pile at minWordLength"""
    bases, read_off, _ = pack_loci([reads])
    o = AsmOptions(*opts)
    cap = 1 << 20
    buf = ctypes.create_string_buffer(cap)
    n = ctypes.c_uint32(0)
    rc = self.lib.manta_debug_repeat_words(self.ctx, ctypes.byref(o), ctypes.c_uint32(len(reads)), bases.ctypes.data_as(ctypes.c_void_p),
                                           read_off.ctypes.data_as(ctypes.c_void_p), buf, ctypes.c_uint64(cap), ctypes.byref(n))
    self._check(rc)
    words = buf.value.decode().split()
    assert len(words) == n.value
    return set(words)


Lib.debug_repeat_words = _debug_repeat_words


class SmallAsmOptions(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("min_word_length", "max_word_length", "word_step_size", "min_contig_length",
                                               "min_coverage", "min_conservative_coverage", "min_seed_reads",
                                               "max_assembly_iterations")]


SMALL_FILTERED_MARK = 0xffffffff


def _small_assemble_batch(self, opts, loci_reads, strict=True):
    """manta_small_assemble_batch (runSmallAssembler).  opts: 8 values in manta_small_asm_options_t order.  One dict per
    locus: contigs as assemble_batch, plus `filtered` = the reads the reference marks isUsed && isFiltered."""
    bases, read_off, begin = pack_loci(loci_reads)
    n_loci = len(loci_reads)
    o = SmallAsmOptions(*opts)
    res = (AsmLocusResult * max(1, n_loci))()
    ccap = n_loci * (o.max_assembly_iterations + 1) + 1
    contigs = (AsmContig * ccap)()
    seq_cap = int(read_off[-1]) * 2 + 65536 * max(1, n_loci)
    seq = np.zeros(seq_cap, dtype=np.uint8)
    bits_cap = n_loci * ((o.max_assembly_iterations + 1) * 2 * 16 + 64) + 64
    bits = np.zeros(bits_cap, dtype=np.uint64)
    su, bu = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = self.lib.manta_small_assemble_batch(self.ctx, ctypes.byref(o), n_loci, bases.ctypes.data_as(ctypes.c_void_p),
                                             read_off.ctypes.data_as(ctypes.c_void_p), begin.ctypes.data_as(ctypes.c_void_p), res,
                                             contigs, ctypes.c_uint64(ccap), seq.ctypes.data_as(ctypes.c_void_p),
                                             ctypes.c_uint64(seq_cap), ctypes.byref(su), bits.ctypes.data_as(ctypes.c_void_p),
                                             ctypes.c_uint64(bits_cap), ctypes.byref(bu))
    self._check(rc, allow=() if strict else (-5, -6, -7))
    out = []
    for l in range(n_loci):
        r = res[l]
        d = dict(status=r.status, n_reads=len(loci_reads[l]), n_words=r.n_words, final_word_length=r.final_word_length,
                 n_iterations=r.n_iterations, contigs=[], pseudo=[], filtered=[])
        if r.status == 0:
            for c in range(r.n_contigs):
                cc = contigs[r.first_contig + c]
                support = _bits_members(bits[cc.support_off:cc.support_off + r.n_words])
                if cc.seed_read_count == SMALL_FILTERED_MARK:
                    assert c == r.n_contigs - 1 and cc.seq_len == 0
                    d["filtered"] = support
                    continue
                d["contigs"].append(dict(seq=seq[cc.seq_off:cc.seq_off + cc.seq_len].tobytes().decode("latin-1"),
                                         seed=cc.seed_read_count, cons=(cc.conservative_begin, cc.conservative_end),
                                         support=support,
                                         reject=_bits_members(bits[cc.reject_off:cc.reject_off + r.n_words])))
        out.append(d)
    return out


Lib.small_assemble_batch = _small_assemble_batch


def small_assembly_text(d):
    """canonical text of oracle/ref_driver.cpp (ref_small_assemble) for one small_assemble_batch locus dict"""
    lines = ["contigs %d" % len(d["contigs"])]
    for i, c in enumerate(d["contigs"]):
        lines.append("contig %d seq=%s seed=%d cons=%d,%d support=%s reject=%s" % (
            i, c["seq"], c["seed"], c["cons"][0], c["cons"][1], ",".join(map(str, c["support"])), ",".join(map(str, c["reject"]))))
    lines.append("reads %d normal %d" % (d["n_reads"], d["n_reads"]))
    filt = set(d["filtered"])
    for r in range(d["n_reads"]):
        ids = [i for i, c in enumerate(d["contigs"]) if r in c["support"]][:1]
        lines.append("read %d used=%d filtered=%d pseudo=0 ids=%s" % (r, 1 if (ids or r in filt) else 0, 1 if r in filt else 0,
                                                                      ",".join(map(str, ids))))
    return "\n".join(lines) + "\n"


def _decode_loci(kind, n_reads, res, contigs, aligns, seq, bits, cig):
    """raw result records/arenas -> one dict per locus (kind: "smallsv" | "spanning")"""
    out = []
    for l in range(len(n_reads)):
        r = res[l]
        d = dict(status=r.status, n_reads=int(n_reads[l]), n_words=r.n_words, final_word_length=r.final_word_length,
                 n_iterations=r.n_iterations, cyclic_iterations=r.cyclic_iterations, contigs=[], pseudo=[], aligns=[])
        if r.status == 0:
            for c in range(r.n_contigs):
                cc = contigs[r.first_contig + c]
                d["contigs"].append(dict(seq=seq[cc.seq_off:cc.seq_off + cc.seq_len].tobytes().decode("latin-1"),
                                         seed=cc.seed_read_count, cons=(cc.conservative_begin, cc.conservative_end),
                                         support=_bits_members(bits[cc.support_off:cc.support_off + r.n_words]),
                                         reject=_bits_members(bits[cc.reject_off:cc.reject_off + r.n_words])))
                a = aligns[r.first_contig + c]
                if kind == "smallsv":
                    d["aligns"].append(dict(status=a.align.status, lead=a.adjusted_leading_cut, trail=a.adjusted_trailing_cut,
                                            score=a.align.score, is_jumped=a.align.is_jumped, begin1=a.align.begin_pos1,
                                            cigar1=cigar_string(cig[a.align.cigar1_off:a.align.cigar1_off + a.align.cigar1_len])))
                else:
                    d["aligns"].append(dict(status=a.align.status, is_uncut=a.is_uncut, score=a.align.score, begin1=a.align.begin_pos1,
                                            begin2=a.align.begin_pos2, jump_insert_size=a.align.jump_insert_size,
                                            jump_range=a.align.jump_range,
                                            cigar1=cigar_string(cig[a.align.cigar1_off:a.align.cigar1_off + a.align.cigar1_len]),
                                            cigar2=cigar_string(cig[a.align.cigar2_off:a.align.cigar2_off + a.align.cigar2_len])))
            off = int(r.pseudo_seq_off)
            for p in range(r.n_pseudo):
                ln = int(bits[r.pseudo_len_off + p])
                d["pseudo"].append(seq[off:off + ln].tobytes().decode("latin-1"))
                off += ln
        out.append(d)
    return out


class SmallSvBatch:
    """Staged fused pipeline (manta_smallsv_*): upload once, run many times (bench), download."""

    def __init__(self, lib, opts, scores, large_indel_score):
        self.lib = lib
        self.h = ctypes.c_void_p()
        o, sc = AsmOptions(*opts), AlignScores(*scores)
        self.max_asm = o.max_assembly_count
        lib.lib.manta_smallsv_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                                 ctypes.POINTER(ctypes.c_void_p)]
        lib.lib.manta_smallsv_destroy.argtypes = [ctypes.c_void_p]
        for f in ("manta_smallsv_run",):
            getattr(lib.lib, f).argtypes = [ctypes.c_void_p]
        lib.lib.manta_smallsv_output_sizes.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_uint64)] * 4
        lib._check(lib.lib.manta_smallsv_create(lib.ctx, ctypes.byref(o), ctypes.byref(sc), large_indel_score, ctypes.byref(self.h)))

    def close(self):
        if self.h:
            self.lib.lib.manta_smallsv_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_packed(self, bases, read_off, begin, refs, ref_off, cuts):
        """numpy arrays; cuts: int32 array (n_loci, 4)"""
        self.n_loci = len(begin) - 1
        self.n_reads = np.diff(begin)
        self.total_bases = int(read_off[-1]) + int(ref_off[-1])
        self._keep = (bases, read_off, begin, refs, ref_off, cuts)
        self.lib._check(self.lib.lib.manta_smallsv_upload(
            self.h, self.n_loci, bases.ctypes.data_as(ctypes.c_void_p), read_off.ctypes.data_as(ctypes.c_void_p),
            begin.ctypes.data_as(ctypes.c_void_p), refs.ctypes.data_as(ctypes.c_void_p),
            ref_off.ctypes.data_as(ctypes.c_void_p), cuts.ctypes.data_as(ctypes.c_void_p)))

    def upload(self, loci_reads, loci_refs, cuts):
        bases, read_off, begin = pack_loci(loci_reads)
        rb = [_b(r) for r in loci_refs]
        ref_off = np.zeros(len(rb) + 1, dtype=np.uint64)
        np.cumsum([len(r) for r in rb], out=ref_off[1:])
        refs = np.frombuffer(b"".join(rb) + b"\0", dtype=np.uint8)
        c = np.ascontiguousarray(np.array(cuts, dtype=np.int32).reshape(len(rb), 4))
        self.upload_packed(bases, read_off, begin, refs, ref_off, c)

    def run(self):
        self.lib._check(self.lib.lib.manta_smallsv_run(self.h))

    def stats(self):
        st = SmallSvStats()
        self.lib._check(self.lib.lib.manta_smallsv_stats(self.h, ctypes.byref(st)))
        return {f[0]: getattr(st, f[0]) for f in SmallSvStats._fields_}

    def download(self, strict=True):
        n = self.n_loci
        res = (AsmLocusResult * n)()
        sizes = [ctypes.c_uint64(0) for _ in range(4)]
        self.lib._check(self.lib.lib.manta_smallsv_output_sizes(self.h, *[ctypes.byref(x) for x in sizes]))
        ccap, seq_cap, bits_cap, cig_cap = [int(x.value) for x in sizes]
        contigs = (AsmContig * ccap)()
        aligns = (SmallSvAlignment * ccap)()
        seq = np.zeros(seq_cap, dtype=np.uint8)
        bits = np.zeros(bits_cap, dtype=np.uint64)
        cig = np.zeros(cig_cap, dtype=np.uint32)
        su, bu, cu = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = self.lib.lib.manta_smallsv_download(
            self.h, res, contigs, aligns, ctypes.c_uint64(ccap), seq.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(seq_cap),
            ctypes.byref(su), bits.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(bits_cap), ctypes.byref(bu),
            cig.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(cig_cap), ctypes.byref(cu))
        self.lib._check(rc, allow=() if strict else (-5, -6, -7))
        return _decode_loci("smallsv", self.n_reads, res, contigs, aligns, seq, bits, cig)


def small_sv_text(d):
    """canonical text of ref_small_sv_locus (oracle/ref_driver.cpp) for one locus result dict"""
    t = assembly_text(d)
    for i, a in enumerate(d["aligns"]):
        t += "align %d lead=%d trail=%d " % (i, a["lead"], a["trail"]) + align_text(ALIGNER_LARGE_INDEL, a)
    return t


def assembly_text(d):
    """canonical text of oracle/ref_driver.cpp for one locus result dict (readInfo re-derived as the C++ adapter does)"""
    lines = ["contigs %d" % len(d["contigs"])]
    for i, c in enumerate(d["contigs"]):
        lines.append("contig %d seq=%s seed=%d cons=%d,%d support=%s reject=%s" % (
            i, c["seq"], c["seed"], c["cons"][0], c["cons"][1], ",".join(map(str, c["support"])), ",".join(map(str, c["reject"]))))
    n_total = d["n_reads"] + len(d["pseudo"])
    lines.append("reads %d normal %d" % (n_total, d["n_reads"]))
    for r in range(n_total):
        ids = [i for i, c in enumerate(d["contigs"]) if r in c["support"]]
        lines.append("read %d used=%d filtered=0 pseudo=%d ids=%s" % (r, 1 if ids else 0, 1 if r >= d["n_reads"] else 0,
                                                                       ",".join(map(str, ids))))
    for i, p in enumerate(d["pseudo"]):
        lines.append("pseudo %d seq=%s" % (d["n_reads"] + i, p))
    return "\n".join(lines) + "\n"


def align_text(kind, r):
    """canonical text of oracle/ref_driver.cpp for one alignment result dict"""
    if kind == ALIGNER_JUMP:
        return "score=%d jumpInsertSize=%d jumpRange=%d begin1=%d cigar1=%s begin2=%d cigar2=%s\n" % (
            r["score"], r["jump_insert_size"], r["jump_range"], r["begin1"], r["cigar1"], r["begin2"], r["cigar2"])
    return "score=%d jumped=%d begin=%d cigar=%s\n" % (r["score"], r["is_jumped"], r["begin1"], r["cigar1"])


class JumpCuts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("align1_leading_cut", "align1_trailing_cut", "align2_leading_cut", "align2_trailing_cut")]


class SpanningAlignment(ctypes.Structure):
    _fields_ = [("is_uncut", ctypes.c_int32), ("reserved", ctypes.c_int32), ("align", AlignResult)]


class SpanningBatch:
    """Staged fused spanning pipeline (manta_spanning_*): assemble -> jump-align on cut references -> re-align rule."""

    def __init__(self, lib, opts, scores, jump_score):
        self.lib = lib
        self.h = ctypes.c_void_p()
        o, sc = AsmOptions(*opts), AlignScores(*scores)
        self.max_asm = o.max_assembly_count
        L = lib.lib
        L.manta_spanning_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
        L.manta_spanning_destroy.argtypes = [ctypes.c_void_p]
        L.manta_spanning_run.argtypes = [ctypes.c_void_p]
        L.manta_spanning_output_sizes.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_uint64)] * 4
        lib._check(L.manta_spanning_create(lib.ctx, ctypes.byref(o), ctypes.byref(sc), jump_score, ctypes.byref(self.h)))

    def close(self):
        if self.h:
            self.lib.lib.manta_spanning_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, loci_reads, refs1, refs2, cuts):
        """cuts: per locus (a1Lead, a1Trail, a2Lead, a2Trail)"""
        bases, read_off, begin = pack_loci(loci_reads)
        self.n_loci = len(begin) - 1
        self.n_reads = np.diff(begin)

        def pack(refs):
            rb = [_b(r) for r in refs]
            off = np.zeros(len(rb) + 1, dtype=np.uint64)
            np.cumsum([len(r) for r in rb], out=off[1:])
            return np.frombuffer(b"".join(rb) + b"\0", dtype=np.uint8), off
        r1, o1 = pack(refs1)
        r2, o2 = pack(refs2)
        c = np.ascontiguousarray(np.array(cuts, dtype=np.int32).reshape(self.n_loci, 4))
        self._keep = (bases, read_off, begin, r1, o1, r2, o2, c)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self.lib._check(self.lib.lib.manta_spanning_upload(self.h, self.n_loci, p(bases), p(read_off), p(begin), p(r1), p(o1), p(r2), p(o2), p(c)))

    def run(self):
        self.lib._check(self.lib.lib.manta_spanning_run(self.h))

    def stats(self):
        st = SmallSvStats()
        self.lib._check(self.lib.lib.manta_spanning_stats(self.h, ctypes.byref(st)))
        return {f[0]: getattr(st, f[0]) for f in SmallSvStats._fields_}

    def download(self, strict=True):
        n = self.n_loci
        res = (AsmLocusResult * n)()
        sizes = [ctypes.c_uint64(0) for _ in range(4)]
        self.lib._check(self.lib.lib.manta_spanning_output_sizes(self.h, *[ctypes.byref(x) for x in sizes]))
        ccap, seq_cap, bits_cap, cig_cap = [int(x.value) for x in sizes]
        contigs = (AsmContig * ccap)()
        aligns = (SpanningAlignment * ccap)()
        seq = np.zeros(seq_cap, dtype=np.uint8)
        bits = np.zeros(bits_cap, dtype=np.uint64)
        cig = np.zeros(cig_cap, dtype=np.uint32)
        su, bu, cu = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = self.lib.lib.manta_spanning_download(
            self.h, res, contigs, aligns, ctypes.c_uint64(ccap), seq.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(seq_cap),
            ctypes.byref(su), bits.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(bits_cap), ctypes.byref(bu),
            cig.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(cig_cap), ctypes.byref(cu))
        self.lib._check(rc, allow=() if strict else (-4, -5, -6, -7))
        return _decode_loci("spanning", self.n_reads, res, contigs, aligns, seq, bits, cig)


# ---------------------------------------------------------------------------------------------------------------
# whole-batch calls (manta_smallsv_batch / manta_spanning_batch) and pinned host memory
# ---------------------------------------------------------------------------------------------------------------
class BatchPlan(ctypes.Structure):
    _fields_ = [("block_loci", ctypes.c_uint32), ("n_workers", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("shared_queue", ctypes.c_void_p)]


class BatchStats(ctypes.Structure):
    _fields_ = [("wall_ms", ctypes.c_double), ("h2d_ms", ctypes.c_double), ("kernel_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double),
                ("assemble_ms", ctypes.c_float), ("schedule_ms", ctypes.c_float), ("align_ms", ctypes.c_float),
                ("n_blocks", ctypes.c_uint32), ("n_workers", ctypes.c_uint32), ("n_alignments", ctypes.c_uint64),
                ("n_align_launches", ctypes.c_uint64), ("dp_cells", ctypes.c_uint64), ("ptr_matrix_bytes", ctypes.c_uint64), ("h2d_bytes", ctypes.c_uint64),
                ("d2h_bytes", ctypes.c_uint64), ("n_loci_lds_small", ctypes.c_uint64), ("n_loci_lds_big", ctypes.c_uint64),
                ("n_loci_handed_back", ctypes.c_uint64), ("n_loci_general", ctypes.c_uint64)]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def pinned_empty(lib, shape, dtype):
    """numpy array over page-locked host memory (manta_host_alloc).  The memory lives until pinned_free(array) or
    process exit (bench/test processes allocate a handful of these once)."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    n = max(count * dtype.itemsize, 1)
    ptr = ctypes.c_void_p()
    lib.lib.manta_host_alloc.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
    lib.lib.manta_host_free.argtypes = [ctypes.c_void_p]
    rc = lib.lib.manta_host_alloc(n, ctypes.byref(ptr))
    if rc != 0:
        raise MantaError(rc, "manta_host_alloc failed")
    buf = (ctypes.c_uint8 * n).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)
    _PINNED[arr.ctypes.data] = (lib, ptr)
    return arr


_PINNED = {}


def pinned_free(arr):
    lib, ptr = _PINNED.pop(arr.ctypes.data)
    lib.lib.manta_host_free(ptr)


def pinned_copy(lib, a):
    out = pinned_empty(lib, a.shape, a.dtype)
    out[...] = a
    return out


class BatchOutput:
    """caller-side result records and arenas of a whole-batch call (allocated once, reusable across calls)"""

    def __init__(self, lib, kind, n_loci, max_asm, seq_cap, bits_cap, cig_cap, pinned=False):
        self.kind, self.n_loci = kind, n_loci
        self.ccap = n_loci * max_asm + 1
        self.res = (AsmLocusResult * n_loci)()
        self.contigs = (AsmContig * self.ccap)()
        self.aligns = ((SmallSvAlignment if kind == "smallsv" else SpanningAlignment) * self.ccap)()
        mk = (lambda n, dt: pinned_empty(lib, (n,), dt)) if pinned else (lambda n, dt: np.zeros(n, dtype=dt))
        self.seq, self.bits, self.cig = mk(seq_cap, np.uint8), mk(bits_cap, np.uint64), mk(cig_cap, np.uint32)
        self.used = [ctypes.c_uint64(0) for _ in range(3)]
        self.stats = BatchStats()

    def decode(self, n_reads):
        return _decode_loci(self.kind, n_reads, self.res, self.contigs, self.aligns, self.seq, self.bits, self.cig)

    def stats_dict(self):
        return {f[0]: getattr(self.stats, f[0]) for f in BatchStats._fields_}


def _smallsv_batch(self, opts, scores, large_indel_score, batch, out, min_wl=None, max_wl=None, block_loci=0, n_workers=0, strict=True, serial_kernels=False, streamed_upload=True, shared_queue=None):
    """batch = (bases, read_off, begin, refs, ref_off, cuts) numpy arrays as synth.config2_batch returns them"""
    bases, read_off, begin, refs, ref_off, cuts = batch
    o, sc, plan = AsmOptions(*opts), AlignScores(*scores), BatchPlan(block_loci, n_workers, (1 if serial_kernels else 0) | (0 if streamed_upload else 2), 0, shared_queue)
    n = len(begin) - 1
    f = self.lib.manta_smallsv_batch
    f.restype = ctypes.c_int
    rc = f(self.ctx, ctypes.byref(o), ctypes.byref(sc), ctypes.c_int32(large_indel_score), ctypes.c_uint32(n), _p(bases), _p(read_off),
           _p(begin), _p(refs), _p(ref_off), _p(cuts), _p(min_wl), _p(max_wl), out.res, out.contigs, out.aligns, ctypes.c_uint64(out.ccap),
           _p(out.seq), ctypes.c_uint64(len(out.seq)), ctypes.byref(out.used[0]), _p(out.bits), ctypes.c_uint64(len(out.bits)),
           ctypes.byref(out.used[1]), _p(out.cig), ctypes.c_uint64(len(out.cig)), ctypes.byref(out.used[2]), ctypes.byref(plan),
           ctypes.byref(out.stats))
    self._check(rc, allow=(() if strict else (-4, -5, -7)) + ((-10,) if shared_queue else ()))
    return rc


def _spanning_batch(self, opts, scores, jump_score, batch, out, min_wl=None, max_wl=None, block_loci=0, n_workers=0, strict=True, serial_kernels=False, streamed_upload=True, shared_queue=None):
    """batch = (bases, read_off, begin, refs1, ref1_off, refs2, ref2_off, cuts)"""
    bases, read_off, begin, r1, o1, r2, o2, cuts = batch
    o, sc, plan = AsmOptions(*opts), AlignScores(*scores), BatchPlan(block_loci, n_workers, (1 if serial_kernels else 0) | (0 if streamed_upload else 2), 0, shared_queue)
    n = len(begin) - 1
    f = self.lib.manta_spanning_batch
    f.restype = ctypes.c_int
    rc = f(self.ctx, ctypes.byref(o), ctypes.byref(sc), ctypes.c_int32(jump_score), ctypes.c_uint32(n), _p(bases), _p(read_off), _p(begin),
           _p(r1), _p(o1), _p(r2), _p(o2), _p(cuts), _p(min_wl), _p(max_wl), out.res, out.contigs, out.aligns, ctypes.c_uint64(out.ccap),
           _p(out.seq), ctypes.c_uint64(len(out.seq)), ctypes.byref(out.used[0]), _p(out.bits), ctypes.c_uint64(len(out.bits)),
           ctypes.byref(out.used[1]), _p(out.cig), ctypes.c_uint64(len(out.cig)), ctypes.byref(out.used[2]), ctypes.byref(plan),
           ctypes.byref(out.stats))
    self._check(rc, allow=(() if strict else (-4, -5, -7)) + ((-10,) if shared_queue else ()))
    return rc


Lib.smallsv_batch = _smallsv_batch
Lib.spanning_batch = _spanning_batch


class Node:
    """manta_node_t: the GPUs of one node behind one block queue (one context per device id)"""

    def __init__(self, path=None, devices=(0,)):
        self.path = path or default_library_path()
        self.lib = ctypes.CDLL(self.path)
        L = self.lib
        L.manta_node_last_error.restype = ctypes.c_char_p
        L.manta_node_last_error.argtypes = [ctypes.c_void_p]
        L.manta_node_destroy.argtypes = [ctypes.c_void_p]
        self.n_devices = len(devices)
        self.node = ctypes.c_void_p()
        ids = (ctypes.c_int32 * len(devices))(*devices)
        rc = L.manta_node_create(ids, ctypes.c_uint32(len(devices)), ctypes.byref(self.node))
        if rc != 0:
            raise MantaError(rc, L.manta_node_last_error(None).decode())

    def close(self):
        if self.node:
            self.lib.manta_node_destroy(self.node)
            self.node = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise MantaError(rc, self.lib.manta_node_last_error(self.node).decode())

    def smallsv_batch(self, opts, scores, large_indel_score, batch, out, min_wl=None, max_wl=None, block_loci=0, n_workers=0, strict=True,
                      streamed_upload=True):
        """manta_node_smallsv_batch; returns the loci each device processed"""
        bases, read_off, begin, refs, ref_off, cuts = batch
        o, sc = AsmOptions(*opts), AlignScores(*scores)
        plan = BatchPlan(block_loci, n_workers, 0 if streamed_upload else 2, 0, None)
        per_dev = (ctypes.c_uint32 * self.n_devices)()
        n = len(begin) - 1
        f = self.lib.manta_node_smallsv_batch
        f.restype = ctypes.c_int
        rc = f(self.node, ctypes.byref(o), ctypes.byref(sc), ctypes.c_int32(large_indel_score), ctypes.c_uint32(n), _p(bases), _p(read_off),
               _p(begin), _p(refs), _p(ref_off), _p(cuts), _p(min_wl), _p(max_wl), out.res, out.contigs, out.aligns, ctypes.c_uint64(out.ccap),
               _p(out.seq), ctypes.c_uint64(len(out.seq)), ctypes.byref(out.used[0]), _p(out.bits), ctypes.c_uint64(len(out.bits)),
               ctypes.byref(out.used[1]), _p(out.cig), ctypes.c_uint64(len(out.cig)), ctypes.byref(out.used[2]), ctypes.byref(plan),
               ctypes.byref(out.stats), per_dev)
        self._check(rc, allow=() if strict else (-4, -5, -7))
        return list(per_dev)

    def spanning_batch(self, opts, scores, jump_score, batch, out, min_wl=None, max_wl=None, block_loci=0, n_workers=0, strict=True,
                       streamed_upload=True):
        bases, read_off, begin, r1, o1, r2, o2, cuts = batch
        o, sc = AsmOptions(*opts), AlignScores(*scores)
        plan = BatchPlan(block_loci, n_workers, 0 if streamed_upload else 2, 0, None)
        per_dev = (ctypes.c_uint32 * self.n_devices)()
        n = len(begin) - 1
        f = self.lib.manta_node_spanning_batch
        f.restype = ctypes.c_int
        rc = f(self.node, ctypes.byref(o), ctypes.byref(sc), ctypes.c_int32(jump_score), ctypes.c_uint32(n), _p(bases), _p(read_off), _p(begin),
               _p(r1), _p(o1), _p(r2), _p(o2), _p(cuts), _p(min_wl), _p(max_wl), out.res, out.contigs, out.aligns, ctypes.c_uint64(out.ccap),
               _p(out.seq), ctypes.c_uint64(len(out.seq)), ctypes.byref(out.used[0]), _p(out.bits), ctypes.c_uint64(len(out.bits)),
               ctypes.byref(out.used[1]), _p(out.cig), ctypes.c_uint64(len(out.cig)), ctypes.byref(out.used[2]), ctypes.byref(plan),
               ctypes.byref(out.stats), per_dev)
        self._check(rc, allow=() if strict else (-4, -5, -7))
        return list(per_dev)


def pack_spanning(loci_reads, refs1, refs2, cuts):
    """python lists -> the packed arrays manta_spanning_upload / manta_spanning_batch take"""
    bases, read_off, begin = pack_loci(loci_reads)

    def pack(refs):
        rb = [_b(r) for r in refs]
        off = np.zeros(len(rb) + 1, dtype=np.uint64)
        np.cumsum([len(r) for r in rb], out=off[1:])
        return np.frombuffer(b"".join(rb) + b"\0", dtype=np.uint8), off
    r1, o1 = pack(refs1)
    r2, o2 = pack(refs2)
    c = np.ascontiguousarray(np.array(cuts, dtype=np.int32).reshape(len(begin) - 1, 4))
    return bases, read_off, begin, r1, o1, r2, o2, c


# ---------------------------------------------------------------------------------------------------------------
# packed read piles (manta_packed_piles_t)
# ---------------------------------------------------------------------------------------------------------------
class PackedPilesStruct(ctypes.Structure):
    _fields_ = [("codes", ctypes.c_void_p), ("nmask", ctypes.c_void_p), ("read_len", ctypes.c_void_p), ("read_code_off", ctypes.c_void_p),
                ("read_mask_off", ctypes.c_void_p), ("locus_read_begin", ctypes.c_void_p)]


class PackedPiles:
    """numpy arrays in the layout of manta_packed_piles_t"""

    def __init__(self, codes, nmask, read_len, code_off, mask_off, begin):
        self.codes, self.nmask, self.read_len, self.code_off, self.mask_off, self.begin = codes, nmask, read_len, code_off, mask_off, begin

    def struct(self):
        return PackedPilesStruct(*[a.ctypes.data for a in (self.codes, self.nmask, self.read_len, self.code_off, self.mask_off, self.begin)])

    def nbytes(self):
        return sum(a.nbytes for a in (self.codes, self.nmask, self.read_len, self.code_off, self.mask_off, self.begin))

    def pinned(self, lib):
        return PackedPiles(*[pinned_copy(lib, a) for a in (self.codes, self.nmask, self.read_len, self.code_off, self.mask_off, self.begin)])


def pack_piles(bases, read_off, begin):
    """1-byte-per-base piles (bases uint8 over {A,C,G,T,N}, read_off uint64, locus_read_begin uint32) -> PackedPiles.
    numpy restatement of manta_amd/host/read_pile.hpp::ReadPileBuilder::addRead for tests and bench.py."""
    read_off = np.asarray(read_off, dtype=np.uint64)
    lens = np.diff(read_off).astype(np.int64)
    n_reads = len(lens)
    cw, mw = (lens + 15) // 16, (lens + 31) // 32
    code_off = np.zeros(n_reads + 1, dtype=np.uint64)
    mask_off = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(cw, out=code_off[1:])
    np.cumsum(mw, out=mask_off[1:])
    total = int(read_off[-1])
    b = np.asarray(bases[:total], dtype=np.uint8)
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    lut[ord("N")] = 4
    c = lut[b]
    if (c == 255).any():
        raise ValueError("pack_piles: byte outside {A,C,G,T,N}")
    read_of_base = np.repeat(np.arange(n_reads), lens)
    pos = np.arange(total, dtype=np.int64) - np.repeat(read_off[:-1].astype(np.int64), lens)
    codes = np.zeros(int(code_off[-1]), dtype=np.uint32)
    nmask = np.zeros(int(mask_off[-1]), dtype=np.uint32)
    is_n = c == 4
    cidx = code_off[:-1].astype(np.int64)[read_of_base] + pos // 16
    val = (np.where(is_n, 0, c).astype(np.uint32)) << (30 - 2 * (pos % 16)).astype(np.uint32)
    np.bitwise_or.at(codes, cidx, val)
    if is_n.any():
        midx = mask_off[:-1].astype(np.int64)[read_of_base[is_n]] + pos[is_n] // 32
        np.bitwise_or.at(nmask, midx, (np.uint32(1) << (pos[is_n] % 32).astype(np.uint32)))
    return PackedPiles(codes, nmask, lens.astype(np.uint32), code_off, mask_off, np.ascontiguousarray(begin, dtype=np.uint32))


def _smallsv_upload_piles(self, piles, refs, ref_off, cuts):
    self.n_loci = len(piles.begin) - 1
    self.n_reads = np.diff(piles.begin)
    self._keep = (piles, refs, ref_off, cuts)
    st = piles.struct()
    self.lib._check(self.lib.lib.manta_smallsv_upload_piles(self.h, ctypes.c_uint32(self.n_loci), ctypes.byref(st), _p(refs), _p(ref_off), _p(cuts)))


def _spanning_upload_piles(self, piles, r1, o1, r2, o2, cuts):
    self.n_loci = len(piles.begin) - 1
    self.n_reads = np.diff(piles.begin)
    self._keep = (piles, r1, o1, r2, o2, cuts)
    st = piles.struct()
    self.lib._check(self.lib.lib.manta_spanning_upload_piles(self.h, ctypes.c_uint32(self.n_loci), ctypes.byref(st), _p(r1), _p(o1), _p(r2),
                                                             _p(o2), _p(cuts)))


SmallSvBatch.upload_piles = _smallsv_upload_piles
SpanningBatch.upload_piles = _spanning_upload_piles


def _smallsv_batch_piles(self, opts, scores, large_indel_score, piles, refs, ref_off, cuts, out, min_wl=None, max_wl=None, block_loci=0,
                         n_workers=0, strict=True, serial_kernels=False, streamed_upload=True):
    o, sc, plan = AsmOptions(*opts), AlignScores(*scores), BatchPlan(block_loci, n_workers, (1 if serial_kernels else 0) | (0 if streamed_upload else 2), 0, None)
    n = len(piles.begin) - 1
    st = piles.struct()
    f = self.lib.manta_smallsv_batch_piles
    f.restype = ctypes.c_int
    rc = f(self.ctx, ctypes.byref(o), ctypes.byref(sc), ctypes.c_int32(large_indel_score), ctypes.c_uint32(n), ctypes.byref(st), _p(refs),
           _p(ref_off), _p(cuts), _p(min_wl), _p(max_wl), out.res, out.contigs, out.aligns, ctypes.c_uint64(out.ccap), _p(out.seq),
           ctypes.c_uint64(len(out.seq)), ctypes.byref(out.used[0]), _p(out.bits), ctypes.c_uint64(len(out.bits)), ctypes.byref(out.used[1]),
           _p(out.cig), ctypes.c_uint64(len(out.cig)), ctypes.byref(out.used[2]), ctypes.byref(plan), ctypes.byref(out.stats))
    self._check(rc, allow=() if strict else (-4, -5, -7))
    return rc


Lib.smallsv_batch_piles = _smallsv_batch_piles


def _spanning_batch_piles(self, opts, scores, jump_score, piles, r1, o1, r2, o2, cuts, out, min_wl=None, max_wl=None, block_loci=0, n_workers=0,
                          strict=True, serial_kernels=False, streamed_upload=True, shared_queue=None):
    """manta_spanning_batch_piles: the whole-batch spanning call on packed read piles (PackedPiles)"""
    o, sc, plan = AsmOptions(*opts), AlignScores(*scores), BatchPlan(block_loci, n_workers, (1 if serial_kernels else 0) | (0 if streamed_upload else 2), 0, shared_queue)
    n = len(piles.begin) - 1
    st = piles.struct()
    f = self.lib.manta_spanning_batch_piles
    f.restype = ctypes.c_int
    rc = f(self.ctx, ctypes.byref(o), ctypes.byref(sc), ctypes.c_int32(jump_score), ctypes.c_uint32(n), ctypes.byref(st), _p(r1), _p(o1), _p(r2), _p(o2),
           _p(cuts), _p(min_wl), _p(max_wl), out.res, out.contigs, out.aligns, ctypes.c_uint64(out.ccap), _p(out.seq), ctypes.c_uint64(len(out.seq)),
           ctypes.byref(out.used[0]), _p(out.bits), ctypes.c_uint64(len(out.bits)), ctypes.byref(out.used[1]), _p(out.cig),
           ctypes.c_uint64(len(out.cig)), ctypes.byref(out.used[2]), ctypes.byref(plan), ctypes.byref(out.stats))
    self._check(rc, allow=(() if strict else (-4, -5, -7)) + ((-10,) if shared_queue else ()))
    return rc


Lib.spanning_batch_piles = _spanning_batch_piles
