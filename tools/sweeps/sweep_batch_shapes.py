"""developer sweep (GPU box or emulator): the whole-batch small-SV call on random shapes (loci, block size, workers) against the staged API on the
same batch -- the streamed upload started before the batch is sized, the compaction beside the aligners, stage gates and host ranges all
take their edge paths.  usage: sweep_batch_shapes.py <first seed> <count> [emu]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from manta_amd._capi import Lib
from test_batch_calls import stress_shapes

seed0, n = int(sys.argv[1]), int(sys.argv[2])
lib = Lib(path=os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so")) if len(sys.argv) > 3 else Lib()
t0 = time.time()
for s in range(seed0, seed0 + n):
    rng = random.Random(s)
    nl = rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 64, 100, 257, 700, 1500])
    block = rng.choice([1, 2, 5, 16, 64, 300, nl, nl + 1, 4096])
    workers = rng.choice([1, 1, 2, 3, 4])
    if (nl + block - 1) // block > 64:
        block = max(block, nl // 32 + 1)
    os.environ["MANTA_AMD_HOST_PARTS"] = str(rng.choice([1, 2, 3, 8]))
    stress_shapes(lib, [(nl, block, workers)], 100000 + 7 * s)
print("%d shapes, whole-batch call == staged API on every locus, %.0f s" % (n, time.time() - t0))
