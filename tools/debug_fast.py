"""developer tool (GPU box): the fast assembler's team sizes against the reference digests, with a status histogram"""
import collections, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from manta_amd._capi import BatchOutput, Lib, small_sv_text
from oracle_lib import asm_opts
from synth import config2_batch
from test_digests import C2_OPTS, C2_SCORES, digests
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
want = digests("config2_digests.bin")
batch_all = config2_batch(10000, seed=12345)
lib = Lib(path=os.environ.get("DBG_LIB"))
for blk, workers in ((n, 1),):
    out = BatchOutput(lib, "smallsv", 10000, 10, 64 << 20, 8 << 20, 16 << 20)
    try:
        lib.smallsv_batch(C2_OPTS, C2_SCORES, -100, batch_all, out, block_loci=2500 if blk != n else 10000, n_workers=workers)
    except Exception as e:
        print("batch raised:", e)
    res = out.decode(np.diff(batch_all[2]))
    bad = [l for l, r in enumerate(res) if hashlib.sha256(small_sv_text(r).encode("latin-1")).digest() != want[l]]
    st = collections.Counter(r["status"] for r in res)
    print("team", os.environ.get("MANTA_AMD_FAST_TEAM"), "workers", workers, "bad", len(bad), "first", bad[:8], "status", dict(st), out.stats_dict())
    import difflib, json
    exp = json.load(open(os.path.join(ROOT, "tools", "_dbg_expected.json")))
    for l in bad[:3]:
        if str(l) in exp and res[l]["status"] == 0:
            a, b = exp[str(l)].splitlines(), small_sv_text(res[l]).splitlines()
            d = [x[:230] for x in difflib.unified_diff(a, b, "want", "got", lineterm="", n=0)]
            print("  locus", l, "\n   " + "\n   ".join(d[:14]))
