"""Random-parameter sweep against the UNMODIFIED reference (oracle/_ref) on the wave emulator -- developer tool, authoring container
only (needs /root/reference built into oracle/_ref).  usage: python tools/sweeps/sweep_iter.py [cases] [seed]
Findings of round 2: loci that overflow the typical-case workspace, jumpRange on a path that overruns ref1."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, random, time
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_lib import RefLib, asm_opts
from manta_amd._capi import Lib, assembly_text
from small_asm_cases import random_case
ref = RefLib(); emu = Lib(path=os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so"))
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
bad = 0; unsup = 0; t0 = time.time()
for it in range(N):
    _, reads = random_case(100000 + rng.randint(0, 10**6))
    kmin = rng.choice([6, 11, 15, 21, 25, 31, 41]); step = rng.choice([1, 3, 5, 10]); kmax = kmin + step * rng.randint(0, 5)
    o = asm_opts(minWordLength=kmin, maxWordLength=kmax, wordStepSize=step, minCoverage=rng.choice([1, 1, 2, 3]),
                 minConservativeCoverage=rng.choice([1, 2, 3]), minUnusedReads=rng.choice([1, 2, 3]), minSupportReads=rng.choice([1, 2]),
                 maxAssemblyCount=rng.choice([1, 3, 10]))
    r = emu.assemble_batch(o, [reads], strict=False)[0]
    if r["status"] != 0:
        unsup += 1; continue
    want = ref.assemble(o, reads)
    if assembly_text(r) != want:
        bad += 1
        if bad <= 2: print("MISMATCH", it, o, len(reads)); print(want[:600]); print(assembly_text(r)[:600])
print("iterative sweep", N, "mismatches", bad, "unsupported", unsup, "%.0fs" % (time.time() - t0))
