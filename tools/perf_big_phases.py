"""Phase shares of the LDS pipeline's big class (graph_big_kernel / contig_big_kernel) on config-5 shaped loci without tandem repeats
(profile builds: libmanta_amd_prof.so = both kernels, coarse; libmanta_amd_profg.so = graph_big_kernel alone, fine)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MANTA_AMD_PROFILE"] = "1"
from manta_amd._capi import Lib
from oracle_lib import asm_opts
from synth import breakend_locus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name in ("libmanta_amd_prof.so", "libmanta_amd_profg.so"):
    path = os.path.join(ROOT, "manta_amd", name)
    if not os.path.exists(path):
        continue
    lib = Lib(path=path)
    for k in (25, 41, 75):
        o = asm_opts(minWordLength=k, maxWordLength=max(76, k), minContigLength=75)
        base = [breakend_locus(s, tandem_frac=0.0)[0] for s in range(32)]
        loci = [base[i % 32] for i in range(n)]
        lib.assemble_batch(o, loci[:256])
        t0 = time.time()
        res = lib.assemble_batch(o, loci)
        print("%s k=%d n=%d host %.3fs contigs/locus=%.2f" % (name, k, n, time.time() - t0, sum(len(x["contigs"]) for x in res) / n), flush=True)
