"""Large single-word-length batch of config-5 shaped loci with the default 10 % tandem-repeat mix: fused spanning pipeline,
HIP-event times (developer tool; GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from manta_amd._capi import Lib, SpanningBatch
from oracle_lib import asm_opts
from synth import breakend_locus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
base = [breakend_locus(5000 + s) for s in range(128)]
loci = [base[i % 128] for i in range(n)]
lib = Lib()
b = SpanningBatch(lib, asm_opts(minWordLength=41, minContigLength=75), [2, -8, -12, -1, -1, 0], -100)
b.upload([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [(100, 100, 100, 100)] * n)
for rep in range(2):
    b.run()
    st = b.stats()
    print("n=%d total %.1f ms (assemble %.1f, align+realign %.1f) -> %.0f loci/s" % (n, st["total_ms"], st["assemble_ms"], st["align_ms"], n / st["total_ms"] * 1e3), flush=True)
