#!/usr/bin/env python3
"""Replays a pile dump (manta_amd/host/pile_dump.hpp, "manta-pile-dump v1": per candidate the read pile, the reference window(s), the
cuts, the assembler options and aligner scores -- what reaches the assembler + aligner in a GenerateSVCandidates run) through the
whole-batch ABI calls on packed piles (manta_smallsv_batch_piles / manta_spanning_batch_piles) and prints

  * the canonical text of every candidate (oracle/FORMAT.md; --out FILE),
  * with --check ref|oracle: the comparison of every candidate with the unmodified reference (oracle/_ref/libmanta_ref.so) or the CPU
    restatement run on the same dump here -- test infrastructure, never part of the replay itself,
  * a histogram of reads per candidate and how the assembler stage routed the candidates (LDS pipeline small / big class, handed back,
    general kernel: manta_batch_stats_t::n_loci_*).

SURVEY section 8(d), configs C1/C3/C4: dump a production run (INTEGRATION.md shows the lines an instrumented reference needs; this
repository's refiner writes the dump itself, SVCandidateAssemblyRefiner::setPileDump), replay, diff.

  python tools/replay_piles.py tests/golden/demo_pile_dump.txt.gz --check ref          # on an MI355X
  python tools/replay_piles.py dump.txt.gz --device emu --out replay.txt               # wave emulator (tests/emu), CPU only
"""
import argparse
import collections
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse_dump(path):
    """-> list of dicts: kind 'S' / 'J', id, opt (9 ints), scores (6 ints), extra, cuts (4 ints), refs (1 or 2 str), reads (list of str)"""
    op = gzip.open if path.endswith(".gz") else open
    recs, cur = [], None
    with op(path, "rt") as f:
        first = f.readline().strip()
        if first != "#manta-pile-dump v1":
            raise SystemExit("%s: not a manta-pile-dump v1 file (first line: %r)" % (path, first[:60]))
        for line in f:
            line = line.rstrip("\n")
            if not line or line[0] == "#":
                continue
            tag, _, rest = line.partition(" ")
            if tag in ("S", "J"):
                kv = dict(x.split("=", 1) for x in rest.split())
                cur = dict(kind=tag, id=int(kv["id"]), n_reads=int(kv["reads"]), opt=[int(x) for x in kv["opt"].split(",")],
                           scores=[int(x) for x in kv["scores"].split(",")], extra=int(kv["extra"]), cuts=[int(x) for x in kv["cuts"].split(",")],
                           refs=[], reads=[])
                recs.append(cur)
            elif tag == "W":
                cur["refs"].append(rest)
            elif tag == "r":
                cur["reads"].append(rest)
            else:
                raise SystemExit("%s: unknown line tag %r" % (path, tag))
    for r in recs:
        if len(r["reads"]) != r["n_reads"] or len(r["refs"]) != (1 if r["kind"] == "S" else 2):
            raise SystemExit("%s: record %d is incomplete" % (path, r["id"]))
    return recs


def span_text(r):
    from test_digests import c5_text
    from manta_amd._capi import assembly_text
    got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"]) for a in r["aligns"]]
    return c5_text(assembly_text(r), got)


def replay(lib, recs):
    """-> ({record id: canonical text}, routing counters summed over the calls)"""
    from manta_amd._capi import BatchOutput, pack_loci, pack_piles, small_sv_text
    texts, routing = {}, collections.Counter()
    groups = collections.defaultdict(list)
    for r in recs:
        groups[(r["kind"], tuple(r["opt"]), tuple(r["scores"]), r["extra"])].append(r)
    for (kind, opt, scores, extra), rs in sorted(groups.items()):
        n = len(rs)
        bases, read_off, begin = pack_loci([[x.encode("latin-1") for x in r["reads"]] for r in rs])
        piles = pack_piles(bases, read_off, begin)
        n_reads = np.diff(begin)
        tot = int(read_off[-1])
        max_asm = opt[8]

        def pack_refs(which):
            rb = [r["refs"][which].encode("latin-1") for r in rs]
            off = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum([len(x) for x in rb], out=off[1:])
            return np.frombuffer(b"".join(rb) + b"\0", dtype=np.uint8), off
        cuts = np.ascontiguousarray(np.array([r["cuts"] for r in rs], dtype=np.int32).reshape(n, 4))
        out = BatchOutput(lib, "smallsv" if kind == "S" else "spanning", n, max_asm, 4 * tot + 65536 * n + (1 << 20),
                          40 * int(n_reads.sum()) // 64 + 256 * n + 4096, 4096 * n + 65536)
        # opt in the ABI's order: (minWordLength, maxWordLength, wordStepSize, minContigLength, minCoverage, minConservativeCoverage,
        # minUnusedReads, minSupportReads, maxAssemblyCount) -- the dump's own order
        if kind == "S":
            refs, ref_off = pack_refs(0)
            lib.smallsv_batch_piles(list(opt), list(scores), extra, piles, refs, ref_off, cuts, out, strict=False)
            res = out.decode(n_reads)
            for r, d in zip(rs, res):
                texts[r["id"]] = small_sv_text(d) if d["status"] == 0 else "STATUS %d\n" % d["status"]
        else:
            r1, o1 = pack_refs(0)
            r2, o2 = pack_refs(1)
            lib.spanning_batch_piles(list(opt), list(scores), extra, piles, r1, o1, r2, o2, cuts, out, strict=False)
            res = out.decode(n_reads)
            for r, d in zip(rs, res):
                texts[r["id"]] = span_text(d) if d["status"] == 0 else "STATUS %d\n" % d["status"]
        st = out.stats_dict()
        for k in ("n_loci_lds_small", "n_loci_lds_big", "n_loci_handed_back", "n_loci_general"):
            routing[k] += int(st[k])
    return texts, routing


def checker_text(cpu, r):
    """the same candidate through a CPU checker (RefLib: the unmodified reference; OracleLib: the restatement)"""
    reads = [x.encode("latin-1") for x in r["reads"]]
    if r["kind"] == "S":
        return cpu.small_sv_locus(r["opt"], r["scores"], r["extra"], reads, r["refs"][0].encode("latin-1"), r["cuts"])
    from test_digests import c5_text
    from test_spanning_pipeline import oracle_locus
    assert r["scores"][:5] == [2, -8, -12, -1, -1] and r["extra"] == -100, "the spanning checker is written for Manta's spanningAlignScores"
    text, aligns = oracle_locus(cpu, r["opt"], reads, r["refs"][0].encode("latin-1"), r["refs"][1].encode("latin-1"), tuple(r["cuts"]))
    return c5_text(text, aligns)


def histogram(recs):
    edges = [0, 1, 3, 8, 16, 32, 64, 128, 256, 512, 1001]
    names = ["0", "1-2", "3-7", "8-15", "16-31", "32-63", "64-127", "128-255", "256-511", "512-1000"]
    h = collections.Counter()
    for r in recs:
        n = len(r["reads"])
        for i in range(len(names)):
            if edges[i] <= n < edges[i + 1]:
                h[names[i]] += 1
                break
        else:
            h[">1000"] += 1
    return [(k, h[k]) for k in names + [">1000"] if h[k]]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dump")
    ap.add_argument("--device", choices=("gpu", "emu"), default="gpu", help="gpu: manta_amd/libmanta_amd.so on an MI355X; emu: the wave emulator of tests/emu (test infrastructure)")
    ap.add_argument("--out", help="write the canonical text of every candidate here")
    ap.add_argument("--check", choices=("ref", "oracle"), help="compare every candidate with the unmodified reference (oracle/_ref) / the CPU restatement")
    args = ap.parse_args()
    from manta_amd._capi import Lib
    recs = parse_dump(args.dump)
    lib = Lib(os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so")) if args.device == "emu" else Lib()
    texts, routing = replay(lib, recs)
    if args.out:
        with open(args.out, "w") as f:
            for r in recs:
                f.write("== candidate %d (%s, %d reads)\n%s" % (r["id"], "small" if r["kind"] == "S" else "spanning", len(r["reads"]), texts[r["id"]]))
    n_small = sum(1 for r in recs if r["kind"] == "S")
    counts = sorted(len(r["reads"]) for r in recs)
    print("dump: %s -- %d candidates (%d small-SV, %d spanning), reads per candidate: min %d, median %d, mean %.1f, max %d; device: %s"
          % (os.path.relpath(args.dump, ROOT) if os.path.abspath(args.dump).startswith(ROOT) else args.dump, len(recs), n_small, len(recs) - n_small,
             counts[0], counts[len(counts) // 2], sum(counts) / len(counts), counts[-1], lib.device_name()))
    print("reads per candidate:")
    for k, v in histogram(recs):
        print("  %-9s %6d  %s" % (k, v, "#" * max(1, (60 * v) // len(recs))))
    print("assembler routing: LDS pipeline small class %d, big class %d, handed back to the general kernel %d, outside both classes (general kernel) %d"
          % (routing["n_loci_lds_small"], routing["n_loci_lds_big"], routing["n_loci_handed_back"], routing["n_loci_general"]))
    failed = sum(1 for r in recs if texts[r["id"]].startswith("STATUS"))
    print("candidates with a device status other than 0: %d" % failed)
    bad = []
    if args.check:
        from oracle_lib import OracleLib, RefLib
        cpu = RefLib() if args.check == "ref" else OracleLib()
        for r in recs:
            if checker_text(cpu, r) != texts[r["id"]]:
                bad.append(r["id"])
        print("parity vs %s: %d of %d candidates differ%s" % ("the unmodified reference (oracle/_ref/libmanta_ref.so)" if args.check == "ref" else "the CPU restatement",
                                                             len(bad), len(recs), (": " + str(bad[:20])) if bad else ""))
    return 1 if (bad or failed) else 0


if __name__ == "__main__":
    sys.exit(main())
