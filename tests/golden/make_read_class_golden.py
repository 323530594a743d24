#!/usr/bin/env python3
"""Read-gathering goldens from the reference's bundled demo (src/demo/data, BASELINE config 1).  Run in the authoring container
(needs /root/reference; `make -C oracle bam`).

For every demo candidate: the records of each region query getBreakendReads makes (one per breakend and BAM; dumped field by field
through the reference's own BAM layer, oracle/ref_bam_driver.cpp::ref_region_records), the reference windows the refiner fetches,
and the pile the reference's UNMODIFIED SVCandidateAssembler::getBreakendReads builds from them (ref_breakend_pile).
Writes tests/golden/read_class_demo.json.gz; tests/test_read_class.py replays the records through the restatement and the kernel."""
import ctypes
import gzip
import json
import os
import subprocess
import sys
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import read_class_util as u  # noqa: E402

DEMO = "/root/reference/src/demo/data/"
BAMS = [DEMO + "HCC1954.NORMAL.30x.compare.COST16011_region.bam", DEMO + "G15512.HCC1954.1.COST16011_region.bam"]
TUMOR = [False, True]
TMP = "/tmp/manta_demo_ref"
FA = os.path.join(TMP, "Homo_sapiens_assembly19.COST16011_region.fa")
R, L, C, U = u.RIGHT_OPEN, u.LEFT_OPEN, u.COMPLEX, u.UNKNOWN

# (tid, centre, half, state) per breakend; the first five are the candidates of make_demo_golden.py, the rest move the windows and
# states over the same region (every state combination of a spanning candidate, narrow and wide intervals, a depth-filtered one)
CANDIDATES = [
    dict(name="BND 8:107653518 <-> 11:94975747", bp=[(0, 107653518, 60, R), (1, 94975747, 60, R)]),
    dict(name="BND 8:107653411 <-> 11:94987872", bp=[(0, 107653411, 70, L), (1, 94987872, 50, R)]),
    dict(name="wide regions", bp=[(0, 107653490, 150, R), (1, 94975790, 150, R)]),
    dict(name="complex chr8", bp=[(0, 107653460, 80, C)]),
    dict(name="complex chr11", bp=[(1, 94975800, 100, C)]),
    dict(name="left/left", bp=[(0, 107653411, 40, L), (1, 94975747, 30, L)]),
    dict(name="right/left", bp=[(1, 94987872, 90, R), (0, 107653518, 20, L)]),
    dict(name="complex, small indel size", bp=[(0, 107653500, 250, C)], min_variant=4),
    dict(name="complex with depth filter", bp=[(0, 107653460, 120, C)], chrom_depth=2.0),
    dict(name="spanning with depth filter", bp=[(0, 107653518, 60, R), (1, 94975747, 60, R)], chrom_depth=4.0),
    dict(name="unknown state", bp=[(1, 94987872, 200, U)]),
]


def main():
    os.makedirs(TMP, exist_ok=True)
    if not os.path.exists(FA):
        tarfile.open(DEMO + "Homo_sapiens_assembly19.COST16011_region.fa.tar.bz2").extractall(TMP)
        subprocess.check_call(["cp", DEMO + "Homo_sapiens_assembly19.COST16011_region.fa.fai", TMP])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "bam", "oracle"])
    rb = u.RefBam()
    orc = u._oracle_lib()
    cases, regions = [], {}
    for c in CANDIDATES:
        bps = [(t, ce - h, ce + h, st) for t, ce, h, st in c["bp"]]
        cd, depth = "", c.get("chrom_depth")
        if depth:
            cd = os.path.join(TMP, "chromdepth_%g.txt" % depth)
            open(cd, "w").write("8\t%g\n11\t%g\n" % (depth, depth))
        minvar = c.get("min_variant", 10)
        ref = rb.pile(BAMS, TUMOR, FA, cd, minvar, False, bps[0], bps[1] if len(bps) == 2 else None)
        rev = [False, False]
        if len(bps) == 2 and bps[0][3] == bps[1][3]:
            rev = [False, True] if bps[0][3] == R else [True, False]
        scans = []
        for k, bp in enumerate(bps):
            sb, se = ctypes.c_int32(), ctypes.c_int32()
            orc.oracle_read_search_range(bp[1], bp[2], ctypes.byref(sb), ctypes.byref(se))
            roff, rseq = ref["ref%d" % (k + 1)]
            for bi in range(2):
                key = "%d:%d:%d-%d" % (bi, bp[0], sb.value, se.value)  # (several candidates query the same region: stored once)
                if key not in regions:
                    regions[key] = rb.region_text(BAMS[bi], FA, bp[0], sb.value, se.value).splitlines()
                scans.append(dict(region=key, bam_index=bi, is_tumor=TUMOR[bi], is_locus_reversed=rev[k], first_of_breakend=(bi == 0),
                                  bp_begin=bp[1], bp_end=bp[2], bp_state=bp[3], ref_begin=roff, ref_seq=rseq))
        cases.append(dict(name=c["name"], scans=scans, is_max_depth=bool(depth), chrom_depth=depth or 0.0, min_variant=minvar, pile=ref["reads"]))
        print("%-36s pile %3d   records %s" % (c["name"], len(ref["reads"]), [len(regions[s["region"]]) for s in scans]), flush=True)
    blob = json.dumps(dict(source="reference demo BAMs through oracle/_ref/libmanta_ref_bam.so: ref_region_records (records), ref_breakend_pile "
                                  "(the unmodified getBreakendReads)", regions=regions, cases=cases)).encode()
    with gzip.GzipFile(os.path.join(HERE, "read_class_demo.json.gz"), "wb", mtime=0) as f:
        f.write(blob)
    print("bytes", len(blob), "->", os.path.getsize(os.path.join(HERE, "read_class_demo.json.gz")))


if __name__ == "__main__":
    main()
