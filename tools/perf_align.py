"""Quick aligner throughput probe (GPU box): config-2 shaped LargeIndel alignments, contig ~270 bp vs ~1.5 kb."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from manta_amd._capi import Lib
from synth import rand_seq, mutate

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(5)
probs = []
base = []
for i in range(64):
    ref = rand_seq(rng, 1500)
    alt = np.concatenate([ref[:750], ref[790:]])
    q = mutate(rng, alt[620:890], 0.003).tobytes()
    base.append((q, ref.tobytes(), rand_seq(rng, 700).tobytes()) if kind == 2 else (q, ref.tobytes()))
probs = [base[i % 64] for i in range(n)]
lib = Lib()
print(lib.device_name())
sc = [2, -8, -24, -1, -1, 0] if kind != 2 else [2, -8, -12, -1, -1, 0]
lib.align_batch(kind, sc, -100, probs[:256])
for rep in range(3):
    t0 = time.time()
    res = lib.align_batch(kind, sc, -100, probs)
    dt = time.time() - t0
    cells = sum(len(p[0]) * (len(p[1]) + (len(p[2]) if kind == 2 else 0)) for p in probs)
    print("n=%d  %.3f s  %.0f aln/s  %.1f GCUPS (host-timed incl. staging)" % (n, dt, n / dt, cells / dt / 1e9))
print(res[0])
