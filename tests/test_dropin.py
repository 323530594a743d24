"""The drop-in proof against the REAL Manta headers: oracle/_ref/libmanta_ref_dropin_{emu,gpu}.so is the
reference's own SVCandidateAssemblyRefiner.cpp -- unmodified, real SVCandidate / SVCandidateAssemblyData / GSCOptions types -- built
with manta_amd/host/dropin/ in front of the reference headers (INTEGRATION.md section A applied mechanically): its assembler and
its three aligners run in libmanta_amd.  It must produce, call for call, what the all-reference build produces."""
import json
import os
import subprocess

import pytest

from refiner_loci import RefinerLib
from test_refiner import GOLDEN, scenario_cases
from test_vcf_candidate import GOLDEN_VCF, vcf_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
DEMO = os.path.join(ROOT, "tests", "golden", "demo_cases.json")


def lib(name):
    path = os.path.join(REF_DIR, name)
    if not os.path.exists(path):
        pytest.skip("%s not built (reference sources unavailable)" % name)
    return RefinerLib(path, "ref")


@pytest.fixture(scope="module")
def dropin_emu(emu):
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref", "_ref/libmanta_ref_dropin_emu.so"])
    return lib("libmanta_ref_dropin_emu.so")


def test_dropin_refiner_equals_reference_refiner(dropin_emu):
    ref = lib("libmanta_ref_refiner.so")
    n = 0
    for seed in (41, 42):
        for name, c in scenario_cases(seed):
            assert dropin_emu.run(c) == ref.run(c), (seed, name)
            n += 1
    for name, c in vcf_cases(43):
        assert dropin_emu.vcf(c) == ref.vcf(c), name
    assert n >= 50


def test_dropin_batched_refiner_over_manta_types(dropin_emu):
    """The BATCHED call over Manta's own types (manta_amd/host/dropin/batch_refiner.hpp): std::vector<SVCandidate> in,
    std::vector<SVCandidateAssemblyData> out, one device batch inside.  Every scenario call through it equals the all-reference
    build's single call; a candidate list that exercises the cross-candidate interval filter (a close spanning pair transferred to the
    local assembler, then the same region as a complex candidate: isOverlapSkip) equals the reference refiner called twice."""
    import random
    from refiner_loci import spanning_case
    ref = lib("libmanta_ref_refiner.so")
    n = 0
    for seed in (41, 42):
        for name, c in scenario_cases(seed):
            assert dropin_emu.run_multi([c], True) == ref.run(c), (seed, name)
            n += 1
    assert n >= 50
    rng = random.Random(11)
    close = spanning_case(rng, "RL", same_chrom=True, far=False)
    cx = dict(close)
    lo, hi = min(close["begin"]), max(close["end"])
    cx.update(state=[3, 0], begin=[lo + 1, lo + 1], end=[hi - 1, hi - 1])
    both = dropin_emu.run_multi([close, cx], True)
    assert both.startswith(ref.run(close)) and "isOverlapSkip=1" in both
    twice = dict(close, calls=2)  # (the driver repeats the identical call on ONE reference refiner object)
    assert dropin_emu.run_multi([twice], True) == ref.run(twice)


def test_dropin_refiner_on_demo_piles(dropin_emu):
    for c in json.load(open(DEMO))["cases"]:
        assert dropin_emu.run(c["case"]) == c["ref_text"], c["name"]
        assert dropin_emu.vcf(c["case"]) == c["ref_vcf"], c["name"]


def small_assemble_text(rl, opts7, reads):
    """ref_refiner_small_assemble of a RefinerLib: runSmallAssembler with the reference's real types"""
    import ctypes
    rb = [r.encode("latin-1") for r in reads]
    arr = (ctypes.c_char_p * len(rb))(*rb)
    lens = (ctypes.c_uint32 * len(rb))(*[len(r) for r in rb])
    cap = 1 << 20
    buf = ctypes.create_string_buffer(cap)
    n = rl.lib.ref_refiner_small_assemble((ctypes.c_uint32 * 7)(*opts7), len(rb), arr, lens, buf, cap)
    assert 0 <= n < cap
    return buf.value.decode("latin-1")


def check_dropin_small_assembler(d, count):
    import hashlib
    from small_asm_cases import UNIT_CASES, UNIT_OPTS, random_case
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "small_assembler_cases.json")))
    for n, reads in UNIT_CASES.items():
        assert small_assemble_text(d, UNIT_OPTS, reads) == gold["unit"][n], n
    for s in range(count):
        o, reads = random_case(s)
        assert hashlib.sha256(small_assemble_text(d, o, reads).encode("latin-1")).hexdigest() == gold["random_sha256"][s], s


def test_dropin_small_assembler(dropin_emu):
    """dropin/runSmallAssembler.cpp (real AssemblyReadInfo / AssembledContig / SmallAssemblerOptions) == assembly/SmallAssembler.cpp"""
    ref = lib("libmanta_ref_refiner.so")
    from small_asm_cases import UNIT_CASES, UNIT_OPTS
    for n, reads in UNIT_CASES.items():
        assert small_assemble_text(ref, UNIT_OPTS, reads) == small_assemble_text(dropin_emu, UNIT_OPTS, reads), n
    check_dropin_small_assembler(dropin_emu, 120)


@pytest.mark.gpu
def test_gpu_dropin_refiner_golden(gpu):
    """the prebuilt device variant travels to the GPU box in oracle/_ref (it cannot be rebuilt there: no /root/reference)"""
    d = lib("libmanta_ref_dropin_gpu.so")
    g = json.load(open(GOLDEN))
    for (name, c), want in zip(scenario_cases(g["seed"]), g["texts"]):
        assert d.run(c) == want, name
    gv = json.load(open(GOLDEN_VCF))
    for (name, c), want in zip(vcf_cases(gv["seed"]), gv["records"]):
        assert d.vcf(c) == want, name
    for c in json.load(open(DEMO))["cases"]:
        assert d.run(c["case"]) == c["ref_text"], c["name"]
        assert d.vcf(c["case"]) == c["ref_vcf"], c["name"]
        assert d.run_multi([c["case"]], True) == c["ref_text"], c["name"]
    # the batched call over Manta's own types (batch_refiner.hpp) on the device
    for (name, c), want in zip(scenario_cases(g["seed"]), g["texts"]):
        assert d.run_multi([c], True) == want, name
    check_dropin_small_assembler(d, 200)
