"""Scenario families of tests/test_refiner.py under many more seeds: the product refiner (emulator build) against the reference's own
refiner object (oracle/_ref/libmanta_ref_refiner.so) -- developer tool, authoring container only.
usage: python tools/sweeps/sweep_refiner.py [first_seed] [n_seeds]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from refiner_loci import RefinerLib
from test_refiner import scenario_cases

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ref = RefinerLib(os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_refiner.so"), "ref")
emu_path = os.path.join(ROOT, "tests", "cpp", "libhost_refiner_emu.so")
from test_refiner import build_mine
mine = build_mine(os.path.join(ROOT, "tests", "emu"), "manta_amd_emu", "emu")
bad = calls = 0
t0 = time.time()
for seed in range(first, first + n):
    for name, c in scenario_cases(seed):
        calls += 1
        if mine.run(c) != ref.run(c):
            bad += 1
            if bad <= 5:
                print("MISMATCH seed", seed, name)
print("refiner sweep: seeds %d..%d, %d calls, %d mismatches, %.0fs" % (first, first + n - 1, calls, bad, time.time() - t0))
