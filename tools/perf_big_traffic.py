"""One assemble_batch call on config-5 shaped loci without tandem repeats (so that graph_big_kernel / contig_big_kernel take every locus):
run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` for the HBM-side bytes per locus of the pipeline's big class."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from manta_amd._capi import Lib
from oracle_lib import asm_opts
from synth import breakend_locus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = Lib()
o = asm_opts(minWordLength=41, maxWordLength=76, minContigLength=75)
base = [breakend_locus(s, tandem_frac=0.0)[0] for s in range(64)]
loci = [base[i % 64] for i in range(n)]
t0 = time.time()
res = lib.assemble_batch(o, loci)
print("n=%d %.3f s contigs/locus=%.2f final k=%s" % (n, time.time() - t0, sum(len(x["contigs"]) for x in res) / n, sorted(set(x["final_word_length"] for x in res))))
