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


class MantaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("manta_amd error %d: %s" % (code, msg))
        self.code = code


def default_library_path():
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


def cigar_string(packed):
    return "".join("%d%s" % (int(v) >> 4, CIGAR_OPS[int(v) & 15]) for v in packed)


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode("latin-1")


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


def align_text(kind, r):
    """canonical text of oracle/ref_driver.cpp for one alignment result dict"""
    if kind == ALIGNER_JUMP:
        return "score=%d jumpInsertSize=%d jumpRange=%d begin1=%d cigar1=%s begin2=%d cigar2=%s\n" % (
            r["score"], r["jump_insert_size"], r["jump_range"], r["begin1"], r["cigar1"], r["begin2"], r["cigar2"])
    return "score=%d jumped=%d begin=%d cigar=%s\n" % (r["score"], r["is_jumped"], r["begin1"], r["cigar1"])
