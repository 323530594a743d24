#!/usr/bin/env python3
"""Writes tests/golden/small_assembler_cases.json from the UNMODIFIED reference (oracle/_ref/libmanta_ref.so, which holds
assembly/SmallAssembler.cpp): the canonical text of runSmallAssembler for the reference's own unit-test scenarios, and
SHA-256 digests of it for N_RANDOM seeded random piles (tests/small_asm_cases.py regenerates the inputs).
Run in the build container (needs /root/reference): python tests/golden/make_small_asm_golden.py"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import RefLib
from small_asm_cases import EDGE_CASES, UNIT_CASES, UNIT_OPTS, random_case

N_RANDOM = 600
ref = RefLib()
out = {"source": "oracle/_ref/libmanta_ref.so: ref_small_assemble -> runSmallAssembler (assembly/SmallAssembler.cpp:622-685)",
       "unit": {}, "edge": {}, "random_sha256": [], "stats": {}}
for name, reads in UNIT_CASES.items():
    out["unit"][name] = ref.small_assemble(UNIT_OPTS, reads)
for name, (opts, reads) in EDGE_CASES.items():
    out["edge"][name] = ref.small_assemble(opts, reads)
multi = filt = empty = 0
for s in range(N_RANDOM):
    opts, reads = random_case(s)
    txt = ref.small_assemble(opts, reads)
    out["random_sha256"].append(hashlib.sha256(txt.encode("latin-1")).hexdigest())
    nc = int(txt.split("\n", 1)[0].split()[1])
    multi += nc > 1
    empty += nc == 0
    filt += " filtered=1 " in txt
out["stats"] = {"cases": N_RANDOM, "with_two_or_more_contigs": multi, "with_no_contig": empty, "with_filtered_reads": filt}
json.dump(out, open(os.path.join(HERE, "small_assembler_cases.json"), "w"), indent=1)
print(out["stats"])
