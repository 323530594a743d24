#!/usr/bin/env python3
"""Writes tests/golden/split_read_cases.json: inputs of the reference's SplitReadAlignment unit tests plus random cases, with the
output of the UNMODIFIED reference splitReadAligner (oracle/_ref).  Run in the authoring container."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from test_split_read import REF_SO, SplitLib, random_cases, reference_test_cases  # noqa: E402

ref = SplitLib(REF_SO, "ref_")
cases = reference_test_cases() + random_cases(77, 60)
json.dump({"source": "oracle/_ref (unmodified applications/GenerateSVCandidates/SplitReadAlignment.cpp)", "cases": cases,
           "ref_texts": [ref.run(c) for c in cases]}, open(os.path.join(HERE, "split_read_cases.json"), "w"), indent=0)
print(len(cases), "cases")
