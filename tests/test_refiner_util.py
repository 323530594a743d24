"""Host-side refiner glue (manta_amd/host/refiner_util.hpp) against the UNMODIFIED reference statics of
SVCandidateAssemblyRefiner.cpp / AlignmentScoringUtilImpl.hpp / AlignmentUtil.cpp (oracle/_ref/libmanta_ref_refiner.so).

Two tiers: committed golden lines (tests/golden/refiner_helpers.json, written by tests/golden/make_refiner_golden.py from
the reference build) always run; the live fuzz against the reference library runs wherever oracle/_ref can be built."""
import json
import os
import subprocess

import pytest

from refiner_cases import HelperLib, make_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "refiner_helpers.json")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_refiner.so")


@pytest.fixture(scope="module")
def mine():
    so = os.path.join(ROOT, "tests", "cpp", "libhost_refiner.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "manta_amd", "host"), os.path.join(ROOT, "tests", "cpp", "host_refiner_capi.cpp"),
                           "-o", so])
    return HelperLib(so, "mine")


@pytest.fixture(scope="module")
def ref():
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libmanta_ref_refiner.so not built (reference sources unavailable)")
    return HelperLib(REF_SO, "ref")


def test_golden_lines(mine):
    g = json.load(open(GOLDEN))
    cases = make_cases(g["seed"], g["n"])
    assert len(cases) == len(g["lines"])
    for c, want in zip(cases, g["lines"]):
        assert mine.evaluate(c) == want, c


def test_reference_unit_vectors(mine):
    """the vectors of the reference's own tests for these helpers"""
    # AlignmentScoringUtilTest / SVCandidateAssemblyRefinerTest style: a clean 35-base flank passes, a short one fails
    sc = [2, -8, -18, 0, -1, 0]
    assert mine.evaluate(dict(kind="score", scores=sc, cigar="10=2X5I10=")).startswith("6/")
    r = mine.evaluate(dict(kind="smallsv", scores=sc, span=100, cigar="35="))
    assert r.split()[0] == "0:35="
    r = mine.evaluate(dict(kind="smallsv", scores=sc, span=100, cigar="29="))
    assert r.split()[0].startswith("1:")


def test_live_fuzz_against_reference(mine, ref):
    cases = make_cases(20260925, 4500)
    kinds = {}
    for c in cases:
        a, b = mine.evaluate(c), ref.evaluate(c)
        assert a == b, c
        kinds.setdefault(c["kind"], set()).add(a[:6])
    # the fuzz must actually exercise both outcomes of the boolean helpers
    for k in ("spanning", "candidates", "large_insert", "jump", "matchcount"):
        assert len(kinds[k]) > 1, k
