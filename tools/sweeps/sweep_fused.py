"""Random-parameter sweep against the UNMODIFIED reference (oracle/_ref) on the wave emulator -- developer tool, authoring container
only (needs /root/reference built into oracle/_ref).  usage: python tools/sweeps/sweep_fused.py [cases] [seed]
Findings of round 2: loci that overflow the typical-case workspace, jumpRange on a path that overruns ref1."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, random, time
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_lib import RefLib, asm_opts
from manta_amd._capi import Lib, BatchOutput, small_sv_text, pack_loci
from synth import small_indel_locus
ref = RefLib(); emu = Lib(path=os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so"))
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0; t0 = time.time(); stat = {}
for it in range(0, N, 4):
    # four loci per call, common options
    kmin = rng.choice([15, 21, 25, 31, 41]); step = rng.choice([5, 10]); kmax = kmin + step * rng.randint(0, 4)
    o = asm_opts(minWordLength=kmin, maxWordLength=kmax, wordStepSize=step, minCoverage=rng.choice([1, 2, 3]),
                 minConservativeCoverage=rng.choice([1, 2]), minUnusedReads=rng.choice([1, 3]), minSupportReads=rng.choice([1, 2]),
                 maxAssemblyCount=rng.choice([1, 2, 10]))
    sc = [rng.choice([1, 2]), rng.choice([-4, -8]), rng.choice([-12, -24]), rng.choice([-1, -2]), rng.choice([-1, -2]), 0]
    li = rng.choice([-100, -50, -24])
    loci = []
    for j in range(4):
        rl = rng.choice([50, 80, 100, 150]); nr = rng.randint(4, 60); refl = rng.choice([300, 500, 900])
        reads, rf = small_indel_locus(rng.randint(0, 10**7), n_reads=nr, read_len=rl, ref_len=refl, sub_rate=rng.choice([0.0, 0.003, 0.01]),
                                      n_rate=rng.choice([0.0, 0.005]), tandem=rng.random() < 0.25)[:2]
        lead = rng.randint(0, 60); trail = rng.randint(0, 60)
        cuts = (lead, trail, lead + rng.randint(0, 150), trail + rng.randint(0, 150))
        loci.append((reads, rf, cuts))
    bases, read_off, begin = pack_loci([l[0] for l in loci])
    refs = np.frombuffer(b"".join((l[1] if isinstance(l[1], bytes) else l[1].encode()) for l in loci) + b"\0" * 64, dtype=np.uint8)
    ref_off = np.zeros(5, dtype=np.uint64); np.cumsum([len(l[1]) for l in loci], out=ref_off[1:])
    cuts = np.array([l[2] for l in loci], dtype=np.int32)
    out = BatchOutput(emu, "smallsv", 4, o[8], 1 << 21, 1 << 16, 1 << 19)
    try:
        emu.smallsv_batch(o, sc, li, (bases, read_off, begin, refs, ref_off, cuts), out, strict=False)
    except Exception as e:
        print("CALL FAILED", it, e); bad += 1; continue
    res = out.decode(np.diff(begin))
    for j, r in enumerate(res):
        rd = [x if isinstance(x, str) else x.decode() for x in loci[j][0]]
        rf = loci[j][1] if isinstance(loci[j][1], str) else loci[j][1].decode()
        want = ref.small_sv_locus(o, sc, li, rd, rf, loci[j][2])
        got = small_sv_text(r)
        stat[r["status"]] = stat.get(r["status"], 0) + 1
        if got != want:
            bad += 1
            if bad <= 3: print("MISMATCH", it + j, o, sc, li, loci[j][2]); print(want[:500]); print(got[:500])
print("fused sweep", N, "mismatches", bad, "status histogram", stat, "%.0fs" % (time.time() - t0))
