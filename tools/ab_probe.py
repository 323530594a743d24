"""A/B of library builds (developer tool): config-2 pipeline, HIP-event times.  usage: ab_probe.py lib1.so lib2.so ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from manta_amd._capi import Lib, SmallSvBatch
from oracle_lib import asm_opts
from synth import config2_batch
batch = config2_batch(10000, seed=12345)
for path in [None] + sys.argv[1:]:
    lib = Lib(path=os.path.join(ROOT, path) if path else None)
    b = SmallSvBatch(lib, asm_opts(minWordLength=31), [2, -8, -24, -1, -1, 0], -100)
    b.upload_packed(*batch)
    b.run(); b.run(); b.run()
    print(path, b.stats()["assemble_ms"], flush=True)
    b.close()
