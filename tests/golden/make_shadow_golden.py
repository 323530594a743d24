#!/usr/bin/env python3
"""Writes tests/golden/shadow_cases.json: the scenarios of the reference's test_alignShadowRead plus random candidates, with the
output of the UNMODIFIED SVScorePairAltProcessor::realignPairedRead (oracle/_ref).  Run in the authoring container."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from test_shadow_align import REF_SO, ShadowLib, random_cases, reference_unit_test_cases  # noqa: E402

ref = ShadowLib(REF_SO, "ref_")
cases = reference_unit_test_cases() + random_cases(4242, 60)
json.dump({"source": "oracle/_ref (unmodified applications/GenerateSVCandidates/SVScorePairAltProcessor.cpp)", "cases": cases,
           "ref_texts": [ref.run(c) for c in cases]}, open(os.path.join(HERE, "shadow_cases.json"), "w"), indent=0)
print(len(cases), "cases,", sum(ref.run(c).count("pass=1") for c in cases), "usable reads")
