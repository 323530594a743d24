"""developer sweep (GPU box or emulator): random repeat-rich piles of 129..236 reads under random assembler options through the big class'
word-length rounds, each against the CPU restatement.  usage: sweep_rounds.py <first seed> <count> [emu]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MANTA_AMD_ASM_PATH", "fast")
from manta_amd._capi import Lib, assembly_text
from oracle_lib import OracleLib

seed0, n = int(sys.argv[1]), int(sys.argv[2])
lib = Lib(path=os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so")) if len(sys.argv) > 3 else Lib()
orc = OracleLib()
from rounds_cases import rounds_case
cases = []
for s in range(seed0, seed0 + n):
    c = rounds_case(s)
    if c:
        cases.append((s,) + c)
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
