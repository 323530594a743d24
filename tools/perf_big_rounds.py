"""Phase shares of graph_big_kernel on config-5 shaped piles WITH tandem repeats (every word length of the rounds) against piles without:
where the later word lengths' ~2 ms per locus go.  Needs manta_amd/libmanta_amd_profg.so:
  python -c "from manta_amd import build as b; b.build(extra_flags=['-DMANTA_ASM_PROFILE','-DMANTA_LG_PROFILE_GRAPH','-DMANTA_DEV_NO_GENERIC'], out='manta_amd/libmanta_amd_profg.so', obj_dir='manta_amd/build_profg')"
Fine slots (asm_lds_big.hpp: tick): 0 pack, 1 table, 2 offsets, 3 sort, 4 links, 5 preds+sibs, 6 peel + lexicographic ranks + speculation list, 7 slab write."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MANTA_AMD_PROFILE"] = "1"
os.environ.setdefault("MANTA_AMD_ASM_PATH", "fast")
from manta_amd._capi import Lib
from oracle_lib import asm_opts
from synth import breakend_locus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lib = Lib(path=os.path.join(ROOT, "manta_amd", "libmanta_amd_profg.so" if len(sys.argv) < 3 else sys.argv[2]))
for frac in (0.0, 1.0):
    base = [breakend_locus(s, tandem_frac=frac)[0] for s in range(64)]
    loci = [base[i % 64] for i in range(n)]
    for k in (25, 41):
        o = asm_opts(minWordLength=k, maxWordLength=max(76, k), minContigLength=75)
        lib.assemble_batch(o, loci[:128])
        t0 = time.time()
        res = lib.assemble_batch(o, loci)
        print("tandem_frac=%.1f k=%d n=%d host %.3fs word lengths/locus=%.2f" % (frac, k, n, time.time() - t0, sum(x["n_iterations"] for x in res) / n), flush=True)
