"""Phase shares of assemble_kernel on config-5 shaped loci (profile build), with and without tandem-repeat loci."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from manta_amd._capi import Lib
from oracle_lib import asm_opts
from synth import breakend_locus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = Lib(path=os.path.join(ROOT, "manta_amd", "libmanta_amd_prof.so"))
o = asm_opts(minWordLength=41, minContigLength=75)
for tf in (0.0, 1.0):
    base = [breakend_locus(s, tandem_frac=tf)[0] for s in range(32)]
    loci = [base[i % 32] for i in range(n)]
    t0 = time.time()
    res = lib.assemble_batch(o, loci)
    print("tandem_frac=%.1f n=%d host %.2fs contigs/locus=%.2f k=%s cyclic_iters/locus=%.2f" % (
        tf, n, time.time() - t0, sum(len(x["contigs"]) for x in res) / n, sorted(set(x["final_word_length"] for x in res)),
        sum(x["cyclic_iterations"] for x in res) / n), flush=True)
