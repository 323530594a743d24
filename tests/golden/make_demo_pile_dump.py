#!/usr/bin/env python3
"""Pile dump of the reference's bundled demo (src/demo/data, BASELINE config 1) for tools/replay_piles.py.
Run in the authoring container (needs /root/reference; `make -C oracle bam ref`, tests/emu built).

The GenerateSVCandidates binary cannot be built here (boost), so the demo's own candidate list (the SV locus graph) is out of reach;
what stands in for "every candidate of the demo" is a sweep of its covered regions:
  * a complex candidate (+-80 bases) every 100 bases of the three covered stretches (chr8 107.6525-107.6545 M, chr11 94.975-94.9768 M and
    94.987-94.9888 M), and
  * spanning candidates at the three junction pairs of src/demo/expectedResults/somaticSV.vcf.gz with breakend regions of four widths, plus
    a few unrelated region pairs (the noise candidates a locus graph also holds).
For each, oracle/_ref/libmanta_ref_bam.so -- the reference's UNMODIFIED SVCandidateAssembler.cpp + htsapi + htslib -- gathers the assembly
read pile from the two BAMs (getBreakendReads); pile and cropped chromosomes then go through this repository's refiner (host code on the
wave emulator) with its pile dump switched on (manta_amd/host/pile_dump.hpp), which writes each candidate as it reaches the assembler +
aligner: reads, oriented reference windows, cuts, options.  The dump holds inputs only.

Writes tests/golden/demo_pile_dump.txt.gz."""
import ctypes
import gzip
import os
import shutil
import subprocess
import sys
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from refiner_loci import RefinerLib, LEFT_OPEN, RIGHT_OPEN, UNKNOWN, COMPLEX  # noqa: E402

DEMO = "/root/reference/src/demo/data/"
BAMS = [DEMO + "HCC1954.NORMAL.30x.compare.COST16011_region.bam", DEMO + "G15512.HCC1954.1.COST16011_region.bam"]
TMP = "/tmp/manta_demo_ref"
FA = os.path.join(TMP, "Homo_sapiens_assembly19.COST16011_region.fa")
PAD = 3000
STRETCHES = [(0, 107652500, 107654500), (1, 94975000, 94976800), (1, 94987000, 94988800)]
JUNCTIONS = [((0, 107653518, RIGHT_OPEN), (1, 94975747, RIGHT_OPEN)), ((0, 107653411, LEFT_OPEN), (1, 94987872, RIGHT_OPEN)),
             ((1, 94975753, RIGHT_OPEN), (1, 94987865, RIGHT_OPEN))]
NOISE_PAIRS = [((0, 107652900, RIGHT_OPEN), (1, 94975300, LEFT_OPEN)), ((0, 107653900, LEFT_OPEN), (1, 94987400, RIGHT_OPEN)),
               ((1, 94975500, RIGHT_OPEN), (1, 94988300, LEFT_OPEN)), ((0, 107653200, RIGHT_OPEN), (0, 107654100, LEFT_OPEN))]


def candidates():
    out = []
    for tid, lo, hi in STRETCHES:
        for c in range(lo, hi, 100):
            out.append(dict(tid=[tid, tid], centre=[c, c], state=[COMPLEX, UNKNOWN], half=[80, 80]))
    for (t1, p1, s1), (t2, p2, s2) in JUNCTIONS:
        for half in (40, 60, 100, 150):
            out.append(dict(tid=[t1, t2], centre=[p1, p2], state=[s1, s2], half=[half, half]))
    for (t1, p1, s1), (t2, p2, s2) in NOISE_PAIRS:
        out.append(dict(tid=[t1, t2], centre=[p1, p2], state=[s1, s2], half=[80, 80]))
    return out


def main():
    os.makedirs(TMP, exist_ok=True)
    if not os.path.exists(FA):
        tarfile.open(DEMO + "Homo_sapiens_assembly19.COST16011_region.fa.tar.bz2").extractall(TMP)
        subprocess.check_call(["cp", DEMO + "Homo_sapiens_assembly19.COST16011_region.fa.fai", TMP])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "bam", "ref"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    bam = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_bam.so"))
    from test_refiner import build_mine
    mine = build_mine(os.path.join(ROOT, "tests", "emu"), "manta_amd_emu", "emu")
    mine.lib.mine_set_pile_dump.restype = ctypes.c_uint64
    chrom = {}
    with open(FA) as f:
        name, parts = None, []
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    chrom[name] = "".join(parts)
                name, parts = line[1:].split()[0], []
            else:
                parts.append(line.strip())
        chrom[name] = "".join(parts)
    tid_name = ["8", "11"]
    buf = ctypes.create_string_buffer(1 << 26)
    raw = os.path.join(TMP, "demo_pile_dump.txt")
    mine.lib.mine_set_pile_dump(raw.encode())
    n_reads = []
    for c in candidates():
        b = [c["centre"][i] - c["half"][i] for i in range(2)]
        e = [c["centre"][i] + c["half"][i] for i in range(2)]
        bam.ref_demo_pile(2, (ctypes.c_char_p * 2)(*[x.encode() for x in BAMS]), (ctypes.c_int * 2)(0, 1), FA.encode(), c["tid"][0], b[0], e[0],
                          c["state"][0], c["tid"][1], b[1], e[1], c["state"][1], buf, len(buf))
        lines = buf.value.decode().splitlines()
        assert lines[0].startswith("reads "), lines[0][:200]
        n = int(lines[0].split()[1])
        reads = lines[1:1 + n]
        w0 = [c["centre"][i] - PAD for i in range(2)]
        if c["state"][1] == UNKNOWN:
            chroms, tid = [chrom[tid_name[c["tid"][0]]][w0[0]:w0[0] + 2 * PAD].upper()], [0, 0]
            w0[1] = w0[0]
        else:
            chroms, tid = [chrom[tid_name[c["tid"][i]]][w0[i]:w0[i] + 2 * PAD].upper() for i in range(2)], [0, 1]
        case = dict(chroms=chroms, reads=reads, state=c["state"], tid=tid, begin=[b[i] - w0[i] for i in range(2)],
                    end=[e[i] - w0[i] for i in range(2)], large=1 if c["state"][0] == COMPLEX else 0)
        text = mine.run(case)
        n_reads.append(n)
        print("%s %d:%d%s  reads %4d  %s" % ("complex " if c["state"][0] == COMPLEX else "spanning", c["tid"][0], c["centre"][0],
                                             "" if c["state"][1] == UNKNOWN else " <-> %d:%d" % (c["tid"][1], c["centre"][1]), n,
                                             text[:60].replace("\n", " | ")), flush=True)
    written = mine.lib.mine_set_pile_dump(None)
    with open(raw, "rb") as f, gzip.GzipFile(os.path.join(HERE, "demo_pile_dump.txt.gz"), "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    print("candidates %d, reached the assembler %d, reads per gathered pile: min %d median %d max %d" %
          (len(n_reads), written, min(n_reads), sorted(n_reads)[len(n_reads) // 2], max(n_reads)))


if __name__ == "__main__":
    main()
