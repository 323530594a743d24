"""developer tool (GPU box): throughput of manta_read_piles_batch on random record batches (tests/read_class_util.random_batch)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import read_class_util as u
from manta_amd._capi import Lib, read_class_options
n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 150
lib = Lib()
t0 = time.time()
b = u.random_batch(7, n_loci=n_loci, reads_per_scan=(200, 900))
arrays = b.arrays()
print("batch: %d candidates, %d queries, %d records (%.1f s to generate)" % (len(b.loci), len(b.scans), len(b.reads), time.time() - t0), flush=True)
opt = read_class_options()
best = 1e9
for it in range(4):
    t = time.time()
    out = u._call(lib, opt, b, *arrays, strict=False)
    best = min(best, time.time() - t)
bases = sum(int(r.read_len) for r in b.reads)
print("manta_read_piles_batch: best of 4 calls %.2f ms wall (H2D + 3 kernels + D2H) = %.2f M records/s, %.1f M input bases/s; pile reads %d"
      % (best * 1e3, len(b.reads) / best / 1e6, bases / best / 1e6, len(out["pile_read"])))
o = u.run_oracle(b, opt)
t = time.time(); o = u.run_oracle(b, opt); dt = time.time() - t
print("restatement (1 core, same batch): %.1f ms" % (dt * 1e3))
out["piles_text"] = u.piles_text(out["piles"], len(b.loci))
u.same(out, o, len(b.loci))
print("equal to the restatement")
