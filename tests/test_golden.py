"""Known-answer tests transcribed from the reference's own unit tests (tests/golden/make_golden.py)."""
import json
import os
import re

import pytest

from manta_amd._capi import align_text
from oracle_lib import asm_opts

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALIGN = json.load(open(os.path.join(GOLD, "aligner_reference_tests.json")))
ASM = json.load(open(os.path.join(GOLD, "assembler_reference_tests.json")))


def parse_fields(text):
    return dict(kv.split("=", 1) for kv in text.strip().split(" "))


def test_reference_output_satisfies_reference_assertions():
    """the stored output of the compiled reference meets every expectation the reference test asserts"""
    for c in ALIGN:
        f = parse_fields(c["ref_text"])
        for k, v in c["expect"].items():
            key = {"cigar1": "cigar1" if c["kind"] == 2 else "cigar", "begin1": "begin1" if c["kind"] == 2 else "begin",
                   "isJumped": "jumped"}.get(k, k)
            assert str(v) == f[key], (c["name"], k, v, f)
    for c in ASM:
        lines = c["ref_text"].splitlines()
        contigs = [l for l in lines if l.startswith("contig ")]
        reads = [l for l in lines if l.startswith("read ")]
        for k, v in c["expect"].items():
            if k == "contigs.size()":
                assert len(contigs) == v
                continue
            m = re.match(r"contigs\[(\d+)\]\.seq", k)
            if m:
                assert ("seq=%s " % v) in contigs[int(m.group(1))]
                continue
            m = re.match(r"readInfo\[(\d+)\]\.isUsed", k)
            if m:
                assert ("used=%d " % v) in reads[int(m.group(1))]
                continue
            m = re.match(r"readInfo\[(\d+)\]\.contigIds\[(\d+)\]", k)
            if m:
                ids = reads[int(m.group(1))].split("ids=")[1].split(",")
                assert int(ids[int(m.group(2))]) == v
                continue
            raise AssertionError("unhandled expectation " + k)


def test_oracle_reproduces_reference_golden(oracle):
    for c in ALIGN:
        assert oracle.align(c["kind"], c["scores"], c["extra"], c["query"], c["ref1"], c["ref2"]) == c["ref_text"], c["name"]
    for c in ASM:
        assert oracle.assemble(asm_opts(**c["opts"]), c["reads"]) == c["ref_text"], c["name"]


def _check_align(lib):
    for kind in (0, 1, 2):
        cases = [c for c in ALIGN if c["kind"] == kind]
        groups = {}
        for c in cases:
            groups.setdefault((tuple(c["scores"]), c["extra"]), []).append(c)
        for (scores, extra), cs in groups.items():
            res = lib.align_batch(kind, list(scores), extra, [(c["query"], c["ref1"], c["ref2"]) for c in cs])
            for c, r in zip(cs, res):
                assert align_text(kind, r) == c["ref_text"], c["name"]


def test_emulated_kernels_reproduce_aligner_golden(emu):
    _check_align(emu)


@pytest.mark.gpu
def test_gpu_reproduces_aligner_golden(gpu):
    _check_align(gpu)
