"""ctypes bindings for the two CPU checkers (TEST INFRASTRUCTURE):

  * ``RefLib``    -- oracle/_ref/libmanta_ref.so : the unmodified reference sources behind oracle/ref_driver.cpp
  * ``OracleLib`` -- oracle/libmanta_oracle.so   : the CPU restatement (oracle/manta_oracle.cpp)

Both render results as the same canonical text so parity is plain string equality.
"""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# option vector order shared by every entry point
ASM_OPT_FIELDS = ("minWordLength", "maxWordLength", "wordStepSize", "minContigLength", "minCoverage",
                  "minConservativeCoverage", "minUnusedReads", "minSupportReads", "maxAssemblyCount")
ASM_DEFAULTS = dict(minWordLength=41, maxWordLength=76, wordStepSize=5, minContigLength=15, minCoverage=1,
                    minConservativeCoverage=2, minUnusedReads=3, minSupportReads=2, maxAssemblyCount=10)


def asm_opts(**kw):
    d = dict(ASM_DEFAULTS)
    d.update(kw)
    return [d[f] for f in ASM_OPT_FIELDS]


def _b(s):
    return s if isinstance(s, bytes) else s.encode("latin-1")


class _TextLib:
    prefix = None

    def __init__(self, path):
        self.path = path
        self.lib = ctypes.CDLL(path)
        p = self.prefix
        self._assemble = getattr(self.lib, p + "assemble")
        self._align = getattr(self.lib, p + "align")
        self._small = getattr(self.lib, p + "small_sv_locus")
        self._bench = getattr(self.lib, p + "bench_small_sv")
        self._bench.restype = ctypes.c_double

    @staticmethod
    def _reads(reads):
        rb = [_b(r) for r in reads]
        arr = (ctypes.c_char_p * len(rb))(*rb)
        lens = (ctypes.c_uint32 * len(rb))(*[len(r) for r in rb])
        return rb, arr, lens

    def assemble(self, opts, reads):
        rb, arr, lens = self._reads(reads)
        o = (ctypes.c_uint32 * 9)(*opts)
        cap = 1 << 22
        buf = ctypes.create_string_buffer(cap)
        n = self._assemble(o, len(rb), arr, lens, buf, cap)
        txt = buf.value.decode("latin-1")
        if n < 0:
            raise RuntimeError(txt)
        assert n < cap
        return txt

    def small_assemble(self, opts, reads):
        """runSmallAssembler; opts = [minWordLength, maxWordLength, wordStepSize, minCoverage, minConservativeCoverage,
        minSeedReads, maxAssemblyIterations]"""
        fn = getattr(self.lib, self.prefix + "small_assemble")
        rb, arr, lens = self._reads(reads)
        o = (ctypes.c_uint32 * 7)(*opts)
        cap = 1 << 22
        buf = ctypes.create_string_buffer(cap)
        n = fn(o, len(rb), arr, lens, buf, cap)
        txt = buf.value.decode("latin-1")
        if n < 0:
            raise RuntimeError(txt)
        assert n < cap
        return txt

    def align(self, kind, scores, extra, query, ref1, ref2=None):
        s = (ctypes.c_int32 * 6)(*scores)
        cap = 1 << 20
        buf = ctypes.create_string_buffer(cap)
        q, r1 = _b(query), _b(ref1)
        r2 = _b(ref2) if ref2 is not None else None
        n = self._align(kind, s, extra, q, len(q), r1, len(r1), r2, len(r2) if r2 is not None else 0, buf, cap)
        txt = buf.value.decode("latin-1")
        if n < 0:
            raise RuntimeError(txt)
        return txt

    def small_sv_locus(self, opts, scores, large_indel_score, reads, ref, cuts):
        rb, arr, lens = self._reads(reads)
        o = (ctypes.c_uint32 * 9)(*opts)
        s = (ctypes.c_int32 * 6)(*scores)
        cap = 1 << 22
        buf = ctypes.create_string_buffer(cap)
        r = _b(ref)
        n = self._small(o, s, large_indel_score, len(rb), arr, lens, r, len(r), cuts[0], cuts[1], cuts[2], cuts[3],
                        buf, cap)
        txt = buf.value.decode("latin-1")
        if n < 0:
            raise RuntimeError(txt)
        return txt

    def bench_small_sv(self, opts, scores, large_indel_score, bases, read_off, locus_read_begin, refs, ref_off, cuts,
                       n_threads):
        """numpy arrays: bases(uint8), read_off(uint64), locus_read_begin(uint32), refs(uint8), ref_off(uint64)"""
        o = (ctypes.c_uint32 * 9)(*opts)
        s = (ctypes.c_int32 * 6)(*scores)
        n_loci = len(locus_read_begin) - 1
        done = ctypes.c_uint64(0)
        secs = self._bench(o, s, large_indel_score, n_loci, bases.ctypes.data_as(ctypes.c_char_p),
                           read_off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                           locus_read_begin.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                           refs.ctypes.data_as(ctypes.c_char_p),
                           ref_off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cuts[0], cuts[1], cuts[2], cuts[3],
                           n_threads, ctypes.byref(done))
        assert done.value == n_loci
        return secs

    def bench_small_sv_timed(self, opts, scores, large_indel_score, bases, read_off, locus_read_begin, refs, ref_off, cuts,
                             n_threads, loci_total, max_seconds, pin=True):
        """the same pipeline with the thread harness of oracle/bench_harness.hpp: threads started (and pinned) before the clock,
        `loci_total` loci taken round robin from the batch, at most `max_seconds`.  Returns (wall seconds, loci done)."""
        fn = getattr(self.lib, self.prefix + "bench_small_sv_timed")
        fn.restype = ctypes.c_double
        o = (ctypes.c_uint32 * 9)(*opts)
        s = (ctypes.c_int32 * 6)(*scores)
        n_loci = len(locus_read_begin) - 1
        done = ctypes.c_uint64(0)
        secs = fn(o, s, large_indel_score, n_loci, bases.ctypes.data_as(ctypes.c_char_p),
                  read_off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                  locus_read_begin.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                  refs.ctypes.data_as(ctypes.c_char_p),
                  ref_off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cuts[0], cuts[1], cuts[2], cuts[3],
                  n_threads, ctypes.c_uint64(loci_total), ctypes.c_double(max_seconds), 1 if pin else 0, ctypes.byref(done))
        return secs, done.value


class RefLib(_TextLib):
    prefix = "ref_"

    def __init__(self):
        super().__init__(os.path.join(ORACLE_DIR, "_ref", "libmanta_ref.so"))


class OracleLib(_TextLib):
    prefix = "orc_"

    def __init__(self):
        path = os.path.join(ORACLE_DIR, "libmanta_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"])
        super().__init__(path)
        self.lib.orc_hash_bytes.restype = ctypes.c_uint64
        self.lib.orc_hash_bytes_real.restype = ctypes.c_uint64

    def unordered_order(self, keys, real=False):
        kb = [_b(k) for k in keys]
        arr = (ctypes.c_char_p * len(kb))(*kb)
        out = (ctypes.c_uint32 * len(kb))()
        (self.lib.orc_unordered_order_real if real else self.lib.orc_unordered_order)(len(kb), arr, out)
        return list(out)

    def hash_bytes(self, s, real=False):
        b = _b(s)
        f = self.lib.orc_hash_bytes_real if real else self.lib.orc_hash_bytes
        return f(b, ctypes.c_uint64(len(b)))


def have_ref():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libmanta_ref.so"))
