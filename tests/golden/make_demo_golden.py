#!/usr/bin/env python3
"""Real-data goldens from the reference's bundled demo (src/demo/data: HCC1954 tumor/normal region BAMs, BASELINE config 1).
Run in the authoring container (needs /root/reference; `make -C oracle bam ref`).

 1. oracle/_ref/libmanta_ref_bam.so -- the reference's UNMODIFIED manta/SVCandidateAssembler.cpp + htsapi + htslib 1.9 -- gathers
    the assembly read pile of each demo junction from the two BAMs exactly as GenerateSVCandidates would (getBreakendReads,
    insertAssemblyRead: Q<minQval masking, reverse complement) and fetches the breakend reference windows.
 2. The pile and cropped chromosomes go through the reference's UNMODIFIED refiner in memory (oracle/_ref/libmanta_ref_refiner.so):
    canonical text of SVCandidateAssemblyData + the candidateSV.vcf records.
 3. Raw BAM records (4-bit sequence, qualities) with the reference's own text for them, both orientations: vectors for the
    read-pile builder.
Writes tests/golden/demo_cases.json and tests/golden/demo_bam_records.txt."""
import ctypes
import json
import os
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
PAD = 3000  # cropped chromosome = [centre - PAD, centre + PAD): far beyond the 250/700-base windows the refiner fetches

# the junctions of src/demo/expectedResults/somaticSV.vcf.gz (tid 0 = chromosome 8, tid 1 = chromosome 11), given to the refiner the
# way the SV locus graph would: imprecise breakend regions around them
CANDIDATES = [
    dict(name="BND 8:107653518 <-> 11:94975747 (HOMSEQ AA/TT)", tid=[0, 1], centre=[107653518, 94975747], state=[RIGHT_OPEN, RIGHT_OPEN],
         half=[60, 60], expect_pos=[107653518, 94975747]),
    dict(name="BND 8:107653411 <-> 11:94987872", tid=[0, 1], centre=[107653411, 94987872], state=[LEFT_OPEN, RIGHT_OPEN], half=[70, 50],
         expect_pos=[107653411, 94987872]),
    dict(name="BND 8:107653518 <-> 11:94975747, wide regions", tid=[0, 1], centre=[107653490, 94975790], state=[RIGHT_OPEN, RIGHT_OPEN],
         half=[150, 150], expect_pos=[107653518, 94975747]),
    dict(name="complex region at the chr8 junctions", tid=[0, 0], centre=[107653460, 107653460], state=[COMPLEX, UNKNOWN], half=[80, 80],
         expect_pos=None),
    dict(name="complex region on chr11", tid=[1, 1], centre=[94975800, 94975800], state=[COMPLEX, UNKNOWN], half=[100, 100], expect_pos=None),
]


def main():
    os.makedirs(TMP, exist_ok=True)
    if not os.path.exists(FA):
        tarfile.open(DEMO + "Homo_sapiens_assembly19.COST16011_region.fa.tar.bz2").extractall(TMP)
        subprocess.check_call(["cp", DEMO + "Homo_sapiens_assembly19.COST16011_region.fa.fai", TMP])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "bam", "ref"])
    bam = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_bam.so"))
    ref = RefinerLib(os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_refiner.so"), "ref")
    # chromosome text for cropping (the fasta holds two full-length chromosomes, 'N' outside the demo region)
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
    buf = ctypes.create_string_buffer(1 << 24)
    cases = []
    for c in CANDIDATES:
        b = [c["centre"][i] - c["half"][i] for i in range(2)]
        e = [c["centre"][i] + c["half"][i] for i in range(2)]
        bam.ref_demo_pile(2, (ctypes.c_char_p * 2)(*[x.encode() for x in BAMS]), (ctypes.c_int * 2)(0, 1), FA.encode(), c["tid"][0], b[0], e[0],
                          c["state"][0], c["tid"][1], b[1], e[1], c["state"][1], buf, len(buf))
        lines = buf.value.decode().splitlines()
        assert lines[0].startswith("reads "), lines[0][:200]
        n = int(lines[0].split()[1])
        reads = lines[1:1 + n]
        # cropped chromosomes: one per breakend (both from the same real chromosome for the complex candidates)
        w0 = [c["centre"][i] - PAD for i in range(2)]
        if c["state"][1] == UNKNOWN:
            chroms, tid = [chrom[tid_name[c["tid"][0]]][w0[0]:w0[0] + 2 * PAD].upper()], [0, 0]
            w0[1] = w0[0]
        else:
            chroms, tid = [chrom[tid_name[c["tid"][i]]][w0[i]:w0[i] + 2 * PAD].upper() for i in range(2)], [0, 1]
        case = dict(chroms=chroms, reads=reads, state=c["state"], tid=tid, begin=[b[i] - w0[i] for i in range(2)],
                    end=[e[i] - w0[i] for i in range(2)], large=1 if c["state"][0] == COMPLEX else 0)
        text = ref.run(case)
        vcf = ref.vcf(case)
        cases.append(dict(name=c["name"], window_begin=w0, real_tid=c["tid"], expect_pos=c["expect_pos"], case=case, ref_text=text, ref_vcf=vcf))
        print("%-55s reads %3d  refined candidates %d" % (c["name"], n, text.count("\nsv ")), flush=True)
        print("   " + "\n   ".join(l[:160] for l in vcf.splitlines()[:4]))
    json.dump(dict(source="reference demo BAMs through oracle/_ref/libmanta_ref_bam.so (read piles) and libmanta_ref_refiner.so (refiner, VCF)",
                   cases=cases), open(os.path.join(HERE, "demo_cases.json"), "w"), indent=0)
    # raw records for the read-pile builder
    n = bam.ref_bam_records(BAMS[1].encode(), FA.encode(), 0, 107653300, 107653700, 5, 250, buf, len(buf))
    open(os.path.join(HERE, "demo_bam_records.txt"), "w").write(buf.value.decode())
    print("bam records:", buf.value.decode().count("\n"))


if __name__ == "__main__":
    main()
