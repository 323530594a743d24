"""developer sweep (GPU box or emulator): random repeat-rich piles of 129..236 reads under random assembler options through the big class'
word-length rounds, each against the CPU restatement.  usage: sweep_rounds.py <first seed> <count> [emu]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MANTA_AMD_ASM_PATH", "fast")
import numpy as np, synth
from manta_amd._capi import Lib, assembly_text
from oracle_lib import asm_opts, OracleLib

seed0, n = int(sys.argv[1]), int(sys.argv[2])
lib = Lib(path=os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so")) if len(sys.argv) > 3 else Lib()
orc = OracleLib()
cases = []
for s in range(seed0, seed0 + n):
    rng = np.random.default_rng(9000 + s)
    nr = int(rng.integers(129, 237))
    if s % 3 == 0:
        reads = synth.repeat_rich_pile(s, n_reads=nr, read_len=int(rng.integers(40, 90)))
    elif s % 3 == 1:
        reads = synth.small_indel_locus(s, n_reads=nr, read_len=int(rng.integers(60, 120)), ref_len=500, sub_rate=0.01, n_rate=0.005, tandem=True)[0]
    else:
        reads = synth.breakend_locus(s, n_reads=min(nr, 200), read_len=int(rng.integers(80, 160)), ref_len=600, tandem_frac=1.0)[0]
    mac = int(rng.integers(2, 11))
    k0 = int(rng.integers(8, 33))
    o = asm_opts(minWordLength=k0, maxWordLength=k0 + int(rng.integers(0, 40)), wordStepSize=int(rng.integers(1, 8)), minCoverage=int(rng.integers(1, 4)),
                 minConservativeCoverage=int(rng.integers(1, 4)), maxAssemblyCount=mac, minContigLength=15,
                 minUnusedReads=int(rng.integers(1, 5)), minSupportReads=int(rng.integers(1, 4)))
    if len(reads) + 2 * mac <= 256:
        cases.append((s, o, reads))
# piles with the same options go to the device in one call (here: every pile has its own options -> one call each; the batch form is
# covered by the digest tests)
bad, iters, t0 = [], 0, time.time()
for s, o, reads in cases:
    r = lib.assemble_batch(o, [reads])[0]
    iters += r["n_iterations"]
    if r["status"] != 0 or assembly_text(r) != orc.assemble(o, reads):
        bad.append(s)
print("%d piles, %d word lengths in all, %d mismatches %s, %.0f s" % (len(cases), iters, len(bad), bad[:10], time.time() - t0))
sys.exit(1 if bad else 0)
