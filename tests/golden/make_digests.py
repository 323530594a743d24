#!/usr/bin/env python3
"""Writes the full-size parity digests (run in the authoring container, where oracle/_ref -- the UNMODIFIED reference
sources -- is built):

  tests/golden/config2_digests.bin   SHA-256 of the reference's canonical text (oracle/FORMAT.md) for every locus of the
                                     metric's workload: config2_batch(10000, seed=12345), k = 31..76 step 5,
                                     GlobalLargeIndelAligner(2,-8,-24,-1,-1;-100), cuts 100/100/800/800
  tests/golden/config5_digests.bin   same for 2048 config-5 shaped breakend loci (synth.config5_locus, mixed k), assembled by
                                     the reference's runIterativeAssembler and aligned call by call with its GlobalJumpAligner
                                     the way alignJumpContigs does (tests/test_spanning_pipeline.oracle_locus on RefLib)

  tests/golden/mixed_digests.bin     same as config 2 for the 2 048 loci of bench.py's `mixed_shape` leg: synth.mixed_shape_batch(2048,
                                     seed=777), read counts log-uniform 3..1000, k = 41..76 step 5 -- the one batch that takes every
                                     route of the assembler stage at once (both LDS classes, hand-backs, the general kernel)

`make_digests.py [config2] [config5] [mixed]` writes the named files (no argument: all three).  32 bytes per locus, locus order.  The GPU tier (tests/test_digests.py) recomputes the same text from the device results
and compares digests: /root/reference does not exist on the GPU box, the digests travel.
"""
import hashlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_lib import RefLib, asm_opts  # noqa: E402
from synth import config2_batch, config5_locus, mixed_shape_batch, unpack_locus  # noqa: E402
from test_spanning_pipeline import oracle_locus  # noqa: E402

C2_N, C2_SEED = 10000, 12345
C5_N = 2048
C2_OPTS = dict(minWordLength=31, maxWordLength=76, wordStepSize=5)
C2_SCORES, C2_LARGE_INDEL, C2_CUTS = [2, -8, -24, -1, -1, 0], -100, (100, 100, 800, 800)
C5_CUTS = (100, 100, 100, 100)


def c5_text(asm_text, aligns):
    """canonical text of a spanning locus: the assembler text + one line per contig alignment"""
    return asm_text + "".join("span %d score=%d ins=%d range=%d begin1=%d cigar1=%s begin2=%d cigar2=%s uncut=%d\n" % ((i,) + tuple(a))
                              for i, a in enumerate(aligns))


MX_N, MX_SEED = 2048, 777
MX_OPTS = dict(minWordLength=41, maxWordLength=76, wordStepSize=5)


def mixed(ref, threads):
    t0 = time.time()
    batch = mixed_shape_batch(MX_N, seed=MX_SEED)
    o = asm_opts(**MX_OPTS)

    def d(l):
        reads, r, cuts = unpack_locus(batch, l)
        return hashlib.sha256(ref.small_sv_locus(o, C2_SCORES, C2_LARGE_INDEL, reads, r, cuts).encode("latin-1")).digest()
    with ThreadPoolExecutor(threads) as ex:
        dig = list(ex.map(d, range(MX_N)))
    open(os.path.join(HERE, "mixed_digests.bin"), "wb").write(b"".join(dig))
    print("mixed shape: %d loci in %.0f s" % (MX_N, time.time() - t0), flush=True)


def main():
    ref = RefLib()
    threads = os.cpu_count() or 8
    which = set(sys.argv[1:]) or {"config2", "config5", "mixed"}
    if "mixed" in which:
        mixed(ref, threads)
    if "config2" in which:
        config2(ref, threads)
    if "config5" in which:
        config5(ref, threads)


def config2(ref, threads):
    t0 = time.time()
    batch = config2_batch(C2_N, seed=C2_SEED)
    o2 = asm_opts(**C2_OPTS)

    def d2(l):
        reads, r, cuts = unpack_locus(batch, l)
        return hashlib.sha256(ref.small_sv_locus(o2, C2_SCORES, C2_LARGE_INDEL, reads, r, cuts).encode("latin-1")).digest()
    with ThreadPoolExecutor(threads) as ex:
        dig = list(ex.map(d2, range(C2_N)))
    open(os.path.join(HERE, "config2_digests.bin"), "wb").write(b"".join(dig))
    print("config 2: %d loci in %.0f s" % (C2_N, time.time() - t0), flush=True)


def config5(ref, threads):
    t0 = time.time()

    def d5(i):
        reads, ref1, ref2, k, kmax = config5_locus(i)
        o = asm_opts(minWordLength=k, maxWordLength=kmax, minContigLength=75)
        text, aligns = oracle_locus(ref, o, reads, ref1, ref2, C5_CUTS)
        return hashlib.sha256(c5_text(text, aligns).encode("latin-1")).digest()
    with ThreadPoolExecutor(threads) as ex:
        dig = list(ex.map(d5, range(C5_N)))
    open(os.path.join(HERE, "config5_digests.bin"), "wb").write(b"".join(dig))
    print("config 5: %d loci in %.0f s" % (C5_N, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
