"""Every locus of a full-size batch against the oracle (GPU box; the oracle runs on the host threads).
config 2: N loci through the fused small-SV pipeline; config-5 shape: M breakend loci through the fused spanning pipeline."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from manta_amd._capi import Lib, SmallSvBatch, SpanningBatch, small_sv_text
from oracle_lib import OracleLib, asm_opts
from synth import config2_batch, unpack_locus, breakend_locus
from test_spanning_pipeline import oracle_locus, SC as SPAN_SC

n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n5 = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
threads = min(128, os.cpu_count() or 8)
lib, orc = Lib(), OracleLib()

o = asm_opts(minWordLength=31)
batch = config2_batch(n2, seed=424242)
b = SmallSvBatch(lib, o, [2, -8, -24, -1, -1, 0], -100)
b.upload_packed(*batch)
b.run()
res = b.download()
t0 = time.time()
def chk2(l):
    reads, ref, cuts = unpack_locus(batch, l)
    return small_sv_text(res[l]) == orc.small_sv_locus(o, [2, -8, -24, -1, -1, 0], -100, reads, ref, cuts)
with ThreadPoolExecutor(threads) as ex:
    ok = list(ex.map(chk2, range(n2)))
print("config 2: %d loci, %d contigs, mismatches vs oracle: %d  (oracle side %.1f s on %d threads)" % (
    n2, sum(len(r["contigs"]) for r in res), ok.count(False), time.time() - t0, threads), flush=True)

o5 = asm_opts(minWordLength=41, minContigLength=75)
loci = [breakend_locus(777000 + s) for s in range(n5)]
sb = SpanningBatch(lib, o5, SPAN_SC, -100)
sb.upload([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [(100, 100, 100, 100)] * n5)
sb.run()
res5 = sb.download()
t0 = time.time()
def chk5(i):
    reads, ref1, ref2 = loci[i]
    _, want = oracle_locus(orc, o5, reads, ref1, ref2, (100, 100, 100, 100))
    got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"]) for a in res5[i]["aligns"]]
    return got == want
with ThreadPoolExecutor(threads) as ex:
    ok5 = list(ex.map(chk5, range(n5)))
print("config-5 shape: %d loci, %d contigs, cyclic loci %d, re-aligned contigs %d, mismatches vs oracle: %d  (oracle side %.1f s)" % (
    n5, sum(len(r["contigs"]) for r in res5), sum(r["cyclic_iterations"] > 0 for r in res5),
    sum(a["is_uncut"] for r in res5 for a in r["aligns"]), ok5.count(False), time.time() - t0), flush=True)
