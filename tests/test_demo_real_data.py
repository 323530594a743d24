"""Real data (BASELINE config 1: the reference's bundled demo, src/demo/data).  tests/golden/make_demo_golden.py gathered the assembly
read piles of the demo's junctions with the reference's UNMODIFIED SVCandidateAssembler (real BAM scan through htslib), ran them
through the reference's UNMODIFIED refiner and candidate-VCF writer, and stored piles, cropped chromosomes and outputs.  Here the
product refiner (device assemble + align) replays them: same SVCandidateAssemblyData text, byte-identical candidateSV.vcf records,
and -- shifted back to genome coordinates -- exactly the breakends the reference's published demo result lists
(src/demo/expectedResults/somaticSV.vcf.gz: 8:107653518 CIPOS=0,2 HOMSEQ=AA <-> 11:94975747 HOMSEQ=TT, 8:107653411 <-> 11:94987872)."""
import json
import os

import numpy as np
import pytest

from manta_amd._capi import pack_piles
import read_class_util as rcu
from manta_amd._capi import read_class_options
from test_read_class import G as RC_DEMO, demo_batch
from test_read_pile import pack_records, pile_lib  # noqa: F401  (fixture)
from test_refiner import mine_emu, mine_gpu  # noqa: F401  (fixtures)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "demo_cases.json")))
CHROM_NAME = ["8", "11"]


def check_cases(lib):
    n_sv = 0
    for c in G["cases"]:
        assert lib.run(c["case"]) == c["ref_text"], c["name"]
        vcf = lib.vcf(c["case"])
        assert vcf == c["ref_vcf"], c["name"]
        if c["expect_pos"]:
            recs = [l.split("\t") for l in vcf.splitlines()]
            assert len(recs) == 2, c["name"]
            for i, r in enumerate(recs):  # record i lives on cropped chromosome i
                assert int(r[1]) + c["window_begin"][i] == c["expect_pos"][i], (c["name"], r[:5])
            n_sv += 1
    return n_sv


def check_chain(lib, refiner):
    """Product only, end to end: the decoded records of the demo BAMs' region queries (tests/golden/read_class_demo.json.gz) ->
    manta_read_piles_batch (read gathering on the device) -> those piles into the product refiner (device assemble + align through
    manta_smallsv_* / manta_spanning_*) -> candidateSV.vcf records; against the unmodified reference's refiner text and VCF records
    for the same candidates (tests/golden/demo_cases.json) and the published demo breakends.  Nothing of oracle/ in between."""
    n_sv = 0
    for i, c in enumerate(G["cases"]):
        rc = RC_DEMO["cases"][i]
        assert rc["pile"] == c["case"]["reads"]  # (the two golden files describe the same candidate)
        p = rcu.run_product(lib, demo_batch([rc]), read_class_options(min_candidate_variant_size=rc["min_variant"]), strict=True)
        case = dict(c["case"], reads=p["piles_text"][0])
        assert refiner.run(case) == c["ref_text"], c["name"]
        vcf = refiner.vcf(case)
        assert vcf == c["ref_vcf"], c["name"]
        if c["expect_pos"]:
            for k, r in enumerate(l.split("\t") for l in vcf.splitlines()):
                assert int(r[1]) + c["window_begin"][k] == c["expect_pos"][k], (c["name"], r[:5])
            n_sv += 1
    return n_sv


def test_emulated_chain_from_bam_records_to_vcf_records(emu, mine_emu):
    assert check_chain(emu, mine_emu) == 3


@pytest.mark.gpu
def test_gpu_chain_from_bam_records_to_vcf_records(gpu, mine_gpu):
    assert check_chain(gpu, mine_gpu) == 3


def test_published_demo_records_are_reproduced_by_the_stored_reference_output():
    c = G["cases"][0]
    r0, r1 = [l.split("\t") for l in c["ref_vcf"].splitlines()]
    assert "CIPOS=0,2;HOMLEN=2;HOMSEQ=AA" in r0[7] and "CIPOS=0,2;HOMLEN=2;HOMSEQ=TT" in r1[7]
    # ALT of the chr8 record names the mate position on chr11: G]11:94975749]
    mate = int(r0[4].split(":")[1].rstrip("]")) + c["window_begin"][1]
    assert r0[4].startswith("G]") and mate == 94975749


def test_emulated_refiner_on_demo_piles(mine_emu):
    assert check_cases(mine_emu) == 3


@pytest.mark.gpu
def test_gpu_refiner_on_demo_piles(mine_gpu):
    assert check_cases(mine_gpu) == 3


def test_read_pile_builder_on_demo_bam_records(pile_lib):
    """250 records of the tumor BAM: raw 4-bit sequence + qualities -> packed pile, both orientations, against the text the
    reference's own bam_seq / reverseCompStr code produced for the same records (minQval 5)"""
    recs = []
    for line in open(os.path.join(ROOT, "tests", "golden", "demo_bam_records.txt")):
        s4, sq, fwd, rev = line.split()
        nib = [int(ch, 16) for ch in s4][:len(fwd)]
        qual = [int(sq[2 * i:2 * i + 2], 16) for i in range(len(fwd))]
        recs.append((nib, qual, fwd, rev))
    assert len(recs) == 250 and any("N" in r[2] for r in recs)
    for is_rev in (False, True):
        texts = [r[3] if is_rev else r[2] for r in recs]
        assert all(t != "-" for t in texts)
        k, acc, codes, mask, coff, moff, rlen = pack_records(pile_lib, [(r[0], r[1], is_rev) for r in recs], 5)
        assert k == len(recs) and all(acc)
        bases = np.frombuffer("".join(texts).encode() + b"\0", dtype=np.uint8)
        off = np.zeros(len(texts) + 1, dtype=np.uint64)
        np.cumsum([len(t) for t in texts], out=off[1:])
        p = pack_piles(bases, off, np.array([0, len(texts)], dtype=np.uint32))
        assert np.array_equal(p.codes, codes[:len(p.codes)]) and np.array_equal(p.nmask, mask[:len(p.nmask)])
        assert np.array_equal(p.read_len, rlen[:k])
