"""developer tool (GPU box): where a tandem-repeat locus of the config-5 generator spends its time in assemble_kernel
(needs manta_amd/libmanta_amd_prof.so: python manta_amd/build.py --profile)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MANTA_AMD_PROFILE"] = "1"
import numpy as np
import synth
from manta_amd._capi import Lib
from oracle_lib import asm_opts

def is_tandem(i, seed0=555000):
    rng = np.random.default_rng(seed0 + i)
    synth.rand_seq(rng, 900); synth.rand_seq(rng, 900)
    return rng.random() < 0.1

lib = Lib(path=os.path.join(ROOT, "manta_amd", os.environ.get("MANTA_PROF_LIB", "libmanta_amd_prof.so")))
tand = [i for i in range(4000) if is_tandem(i)][:int(sys.argv[1]) if len(sys.argv) > 1 else 64]
plain = [i for i in range(4000) if not is_tandem(i)][:len(tand)]
for name, ids in (("plain", plain), ("tandem", tand), ("one tandem", tand[:1]), ("another", tand[5:6])):
    loci = [synth.config5_locus(i) for i in ids]
    # per-locus word lengths as in the spanning batch
    t = []
    for rep in range(2):
        t0 = time.time()
        res = [lib.assemble_batch(asm_opts(minWordLength=l[3], maxWordLength=l[4], minContigLength=75), [l[0]])[0] for l in loci] if len(ids) == 1 else None
        if res is None:
            # same k for a group: run the groups by k
            by = {}
            for l in loci:
                by.setdefault((l[3], l[4]), []).append(l[0])
            for (k, kmax), piles in by.items():
                lib.assemble_batch(asm_opts(minWordLength=k, maxWordLength=kmax, minContigLength=75), piles)
        t.append(time.time() - t0)
    print("== %s: %d loci, wall %.1f ms (second pass)" % (name, len(ids), t[1] * 1e3), flush=True)
