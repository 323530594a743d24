"""Per-phase shader-clock shares of assemble_kernel (developer tool; GPU box).
Needs manta_amd/libmanta_amd_prof.so (python manta_amd/build.py --profile)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MANTA_AMD_PROFILE"] = "1"
from manta_amd._capi import Lib, SmallSvBatch
from oracle_lib import asm_opts
from synth import config2_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
lib = Lib(path=os.environ.get("MANTA_AMD_LIB") or os.path.join(ROOT, "manta_amd", "libmanta_amd_prof.so"))
b = SmallSvBatch(lib, asm_opts(minWordLength=31), [2, -8, -24, -1, -1, 0], -100)
b.upload_packed(*config2_batch(n, seed=12345))
b.run(); b.run()
print(b.stats())
b.download()
