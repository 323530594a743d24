"""Random-parameter sweep against the UNMODIFIED reference (oracle/_ref) on the wave emulator -- developer tool, authoring container
only (needs /root/reference built into oracle/_ref).  usage: python tools/sweeps/sweep_span.py [cases] [seed]
Findings of round 2: loci that overflow the typical-case workspace, jumpRange on a path that overruns ref1."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, random, time, re
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_lib import RefLib, asm_opts
from manta_amd._capi import Lib, SpanningBatch, assembly_text
from synth import breakend_locus
from test_spanning_pipeline import oracle_locus, SC
ref = RefLib(); emu = Lib(path=os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so"))
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0; t0 = time.time(); nuncut = 0; nal = 0
for it in range(0, N, 3):
    kmin = rng.choice([21, 25, 31, 41]); step = rng.choice([5, 10]); kmax = kmin + step * rng.randint(0, 4)
    o = asm_opts(minWordLength=kmin, maxWordLength=kmax, wordStepSize=step, minCoverage=rng.choice([1, 2]), minContigLength=rng.choice([15, 40, 75]),
                 minUnusedReads=rng.choice([1, 3]), minSupportReads=rng.choice([1, 2]), maxAssemblyCount=rng.choice([1, 3, 10]))
    loci, cuts = [], []
    for j in range(3):
        rl = rng.choice([60, 80, 120]); refl = rng.choice([260, 320, 500])
        loci.append(breakend_locus(rng.randint(0, 10**7), n_reads=rng.randint(6, 40), read_len=rl, ref_len=refl, sub_rate=rng.choice([0.0, 0.005]),
                                   n_rate=rng.choice([0.0, 0.01]), tandem_frac=0.0))
        cuts.append(tuple(rng.choice([0, 30, 60, 90, 110]) for _ in range(4)))
    b = SpanningBatch(emu, o, SC, -100)
    try:
        b.upload([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], cuts)
        b.run(); res = b.download(strict=False) if 'strict' in SpanningBatch.download.__code__.co_varnames else b.download()
    except Exception as e:
        print("CALL FAILED", it, str(e)[:200]); bad += 1; continue
    for (reads, r1, r2), c, r in zip(loci, cuts, res):
        text, want = oracle_locus(ref, o, reads, r1, r2, c)
        got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"]) for a in r["aligns"]]
        seqs = re.findall(r"^contig \d+ seq=(\S+)", text, flags=re.M)
        nal += len(got); nuncut += sum(g[7] for g in got)
        if [cc["seq"] for cc in r["contigs"]] != seqs or got != want:
            bad += 1
            if bad <= 3: print("MISMATCH", it, o, c); print(want[:3]); print(got[:3])
print("spanning sweep", N, "mismatches", bad, "alignments", nal, "uncut re-aligns", nuncut, "%.0fs" % (time.time() - t0))
