"""developer tool (GPU box): throughput of manta_read_piles_batch on random record batches (tests/read_class_util.random_batch).
Inputs and outputs live in page-locked host memory (manta_host_alloc), as a feeder that decodes BAM blocks would keep them; the clock
is around the ABI call alone (H2D + five kernels + D2H).  The restatement runs beside it on one core and on all host threads
(candidates dealt out to threads).  Last line: one JSON object (profiles/r04_read_class_line.json)."""
import ctypes, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import read_class_util as u
from manta_amd._capi import Lib, ReadLocusResult, pinned_empty, read_class_options

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 150
lib = Lib(path=os.environ["BENCH_RC_LIB"]) if os.environ.get("BENCH_RC_LIB") else Lib()  # (BENCH_RC_LIB: dry run of this script on the emulator build)
t0 = time.time()
b = u.random_batch(7, n_loci=n_loci, reads_per_scan=(200, 900))
loci, scans, reads, cigars, names, seqs, quals, refs = b.arrays()
n_scans, n_reads = len(b.scans), len(b.reads)
print("batch: %d candidates, %d queries, %d records (%.1f s to generate)" % (n_loci, n_scans, n_reads, time.time() - t0), flush=True)
opt = read_class_options()


def pinned_copy(src_bytes, dtype=np.uint8):
    a = pinned_empty(lib, (len(src_bytes) // np.dtype(dtype).itemsize,), dtype)
    a.view(np.uint8)[:] = np.frombuffer(src_bytes, dtype=np.uint8)
    return a


P = dict(loci=pinned_copy(bytes(loci)[:ctypes.sizeof(loci._type_) * n_loci]), scans=pinned_copy(bytes(scans)[:ctypes.sizeof(scans._type_) * n_scans]),
         reads=pinned_copy(bytes(reads)[:ctypes.sizeof(reads._type_) * n_reads]), cigars=pinned_copy(cigars.tobytes(), np.uint32),
         names=pinned_copy(names.tobytes()), seqs=pinned_copy(seqs.tobytes()), quals=pinned_copy(quals.tobytes()), refs=pinned_copy(refs.tobytes()))
total_len = sum(int(r.read_len) for r in b.reads)
O = dict(decision=pinned_empty(lib, (n_reads + 1,), np.uint8), pile_index=pinned_empty(lib, (n_reads + 1,), np.uint32),
         results=pinned_empty(lib, (ctypes.sizeof(ReadLocusResult) * (n_loci + 1),), np.uint8),
         codes=pinned_empty(lib, (total_len // 16 + n_reads + 4,), np.uint32), nmask=pinned_empty(lib, (total_len // 32 + n_reads + 4,), np.uint32),
         read_len=pinned_empty(lib, (n_reads + 1,), np.uint32), code_off=pinned_empty(lib, (n_reads + 2,), np.uint64),
         mask_off=pinned_empty(lib, (n_reads + 2,), np.uint64), pile_read=pinned_empty(lib, (n_reads + 1,), np.uint32),
         begin=pinned_empty(lib, (n_loci + 1,), np.uint32))
used = (ctypes.c_uint64 * 3)()
f = lib.lib.manta_read_piles_batch
f.restype = ctypes.c_int
V, U32, U64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
f.argtypes = [V, V, U32, V, U32, V, U32, V, V, U64, V, U64, V, U64, V, U64, V, U64, V, V, V, V, U64, V, V, U64, V, V, V, V, V, U64, V, V]
d = lambda a: a.ctypes.data  # noqa: E731


def call():
    return f(lib.ctx, ctypes.addressof(opt), n_loci, d(P["loci"]), n_scans, d(P["scans"]), n_reads, d(P["reads"]), d(P["cigars"]), len(P["cigars"]),
             d(P["names"]), len(P["names"]), d(P["seqs"]), len(P["seqs"]), d(P["quals"]), len(P["quals"]), d(P["refs"]), len(P["refs"]),
             d(O["decision"]), d(O["pile_index"]), d(O["results"]), d(O["codes"]), len(O["codes"]), ctypes.addressof(used), d(O["nmask"]),
             len(O["nmask"]), ctypes.addressof(used) + 8, d(O["read_len"]), d(O["code_off"]), d(O["mask_off"]), d(O["pile_read"]), n_reads,
             ctypes.addressof(used) + 16, d(O["begin"]))


best = 1e9
for it in range(6):
    t = time.perf_counter()
    rc = call()
    best = min(best, time.perf_counter() - t)
    assert rc in (0, -5), rc
bases = total_len
h2d = sum(a.nbytes for a in P.values())
print("manta_read_piles_batch (page-locked host memory): best of 6 calls %.2f ms wall (H2D %.1f MB + 5 kernels + D2H) = %.2f M records/s, "
      "%.1f M input bases/s; pile reads %d" % (best * 1e3, h2d / 1e6, n_reads / best / 1e6, bases / best / 1e6, int(used[2])), flush=True)

# the checker: the CPU restatement on the same batch -- equality, and its time on one core and on all host threads
out = u._call(lib, opt, b, loci, scans, reads, cigars, names, seqs, quals, refs, strict=False)
o = u.run_oracle(b, opt)
t = time.time(); o = u.run_oracle(b, opt); one = time.time() - t
out["piles_text"] = u.piles_text(out["piles"], n_loci)
u.same(out, o, n_loci)
print("restatement (1 core, same batch): %.1f ms; equal to the product's output" % (one * 1e3), flush=True)
threads = os.cpu_count() or 1
parts = [u.random_batch(7000 + k, n_loci=1, reads_per_scan=(200, 900)) for k in range(min(n_loci, 2 * threads))]
olib = u._oracle_lib()
olib.oracle_read_piles.argtypes = [V, U32] + [V] * 11 + [V, U64, V]


def prepared(part):
    """the restatement's call on one candidate with every argument built beforehand: the timed region is the C call alone (ctypes
    releases the interpreter lock around it)"""
    A = part.arrays()
    n = len(part.reads)
    dec, pix = np.zeros(n + 1, dtype=np.uint8), np.zeros(n + 1, dtype=np.uint32)
    res = (ReadLocusResult * 1)()
    cap = sum(int(r.read_len) + 1 for r in part.reads) + 16
    text, usedp = ctypes.create_string_buffer(cap), ctypes.c_uint64()
    keep = (A, dec, pix, res, text, usedp)
    return lambda: (olib.oracle_read_piles(ctypes.addressof(opt), 1, ctypes.addressof(A[0]), ctypes.addressof(A[1]), ctypes.addressof(A[2]), A[3].ctypes.data,
                                           A[4].ctypes.data, A[5].ctypes.data, A[6].ctypes.data, A[7].ctypes.data, dec.ctypes.data, pix.ctypes.data,
                                           ctypes.addressof(res), ctypes.addressof(text), cap, ctypes.addressof(usedp)), keep)[0]


calls = [prepared(p) for p in parts]
with ThreadPoolExecutor(max_workers=threads) as ex:
    list(ex.map(lambda c: c(), calls[:threads]))  # warm
    t = time.time()
    list(ex.map(lambda c: c(), calls))
    many = time.time() - t
part_records = sum(len(p.reads) for p in parts)
print("restatement on %d host threads: %d records in %.1f ms = %.2f M records/s (one candidate per task)"
      % (threads, part_records, many * 1e3, part_records / many / 1e6))
print(json.dumps({"call": "manta_read_piles_batch", "candidates": n_loci, "queries": n_scans, "records": n_reads, "input_bases": bases,
                  "host_memory": "page-locked", "wall_ms": round(best * 1e3, 3), "records_per_s": round(n_reads / best, 1),
                  "h2d_MB": round(h2d / 1e6, 1), "pile_reads": int(used[2]), "restatement_one_core_ms": round(one * 1e3, 1),
                  "restatement_all_threads": {"threads": threads, "records": part_records, "ms": round(many * 1e3, 1),
                                              "records_per_s": round(part_records / many, 1)}}))
