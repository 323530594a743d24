#!/usr/bin/env python3
"""Writes tests/golden/refiner_helpers.json: outputs of the UNMODIFIED reference helper statics
(oracle/_ref/libmanta_ref_refiner.so, built by `make -C oracle ref` from /root/reference) on the seeded cases of
tests/refiner_cases.py.  Run in the build container only (needs /root/reference)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from refiner_cases import HelperLib, make_cases  # noqa: E402

SEED, N = 777, 900
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
ref = HelperLib(os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_refiner.so"), "ref")
lines = [ref.evaluate(c) for c in make_cases(SEED, N)]
json.dump({"seed": SEED, "n": N, "source": "oracle/_ref/libmanta_ref_refiner.so (reference statics, unmodified)", "lines": lines},
          open(os.path.join(ROOT, "tests", "golden", "refiner_helpers.json"), "w"), indent=0)
print("wrote", len(lines), "lines")

# ---- whole refiner calls (tests/test_refiner.py) ----
from refiner_loci import RefinerLib  # noqa: E402
from test_refiner import scenario_cases  # noqa: E402

rl = RefinerLib(os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_refiner.so"), "ref")
cases = scenario_cases(4242)
json.dump({"seed": 4242, "names": [n for n, _ in cases], "source": "reference SVCandidateAssemblyRefiner::getCandidateAssemblyData "
           "(unmodified) via oracle/ref_refiner_driver.cpp", "texts": [rl.run(c) for _, c in cases]},
          open(os.path.join(ROOT, "tests", "golden", "refiner_calls.json"), "w"), indent=0)
print("wrote", len(cases), "refiner calls")

# ---- candidateSV.vcf records (tests/test_vcf_candidate.py) ----
from test_vcf_candidate import vcf_cases  # noqa: E402

vc = vcf_cases(4242)
json.dump({"seed": 4242, "names": [n for n, _ in vc], "source": "reference VcfWriterCandidateSV/VcfWriterSV/JunctionIdGenerator (unmodified) over the "
           "reference refiner's output, oracle/ref_refiner_driver.cpp: ref_candidate_vcf_records", "records": [rl.vcf(c) for _, c in vc]},
          open(os.path.join(ROOT, "tests", "golden", "candidate_vcf_records.json"), "w"), indent=0)
print("wrote", sum(r.count("\n") for r in [rl.vcf(c) for _, c in vc]), "vcf records")
