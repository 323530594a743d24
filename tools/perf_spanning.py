"""Spanning-path throughput probe (GPU box): config-5 shaped breakend loci (200 reads x 250 bp, N-masked, k=41..76),
assemble_batch + jump align_batch, host-timed incl. staging, parity-sampled against the oracle."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from manta_amd._capi import Lib, assembly_text, align_text
from oracle_lib import OracleLib, asm_opts
from synth import breakend_locus

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200
base = [breakend_locus(s, n_reads=n_reads) for s in range(32)]
loci = [base[i % 32] for i in range(n)]
lib = Lib()
o = asm_opts(minWordLength=41, minContigLength=75)
orc = OracleLib()
r = lib.assemble_batch(o, [l[0] for l in base[:8]])
bad = sum(assembly_text(a) != orc.assemble(o, l[0]) for a, l in zip(r, base))
print("assembler parity mismatches on 8 loci:", bad, flush=True)
for rep in range(2):
    t0 = time.time()
    res = lib.assemble_batch(o, [l[0] for l in loci])
    dt = time.time() - t0
    nc = sum(len(x["contigs"]) for x in res)
    print("assemble n=%d reads=%d %.3f s  %.0f loci/s  contigs/locus=%.2f  k=%s" % (n, n_reads, dt, n / dt, nc / n, sorted(set(x["final_word_length"] for x in res))), flush=True)
SC = [2, -8, -12, -1, -1, 0]
problems = []
for x, l in zip(res, loci):
    for c in x["contigs"]:
        problems.append((c["seq"].encode(), l[1][100:800], l[2][100:800]))
for rep in range(2):
    t0 = time.time()
    ar = lib.align_batch(2, SC, -100, problems)
    dt = time.time() - t0
    print("jump align %d tasks %.3f s  %.0f tasks/s" % (len(problems), dt, len(problems) / dt), flush=True)
bad = 0
for p, a in list(zip(problems, ar))[:16]:
    bad += align_text(2, a) != orc.align(2, SC, -100, p[0], p[1], p[2])
print("jump parity mismatches on 16 tasks:", bad)
