"""Assembler throughput probe (GPU box): config-2 shaped loci (80 reads x 150 bp, k=31..76)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from manta_amd._capi import Lib, assembly_text
from oracle_lib import OracleLib, asm_opts
from synth import small_indel_locus

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
base = [small_indel_locus(s)[0] for s in range(64)]
loci = [base[i % 64] for i in range(n)]
lib = Lib()
o = asm_opts(minWordLength=31)
r = lib.assemble_batch(o, loci[:64])
orc = OracleLib()
bad = sum(assembly_text(a) != orc.assemble(o, reads) for a, reads in zip(r, base))
print("parity mismatches on 64 loci:", bad)
for rep in range(2):
    t0 = time.time()
    res = lib.assemble_batch(o, loci)
    dt = time.time() - t0
    print("n=%d %.3f s  %.0f loci/s (host-timed incl. staging)  contigs/locus=%.2f" % (n, dt, n / dt, sum(len(x["contigs"]) for x in res) / n))
