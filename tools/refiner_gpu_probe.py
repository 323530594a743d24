"""Runs the golden refiner calls one by one on the GPU build, logging the time of each (diagnostic)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from test_refiner import scenario_cases, build_mine, GOLDEN
t0 = time.time()
mine = build_mine(os.path.join(ROOT, "manta_amd"), "manta_amd", "gpu")
print("built %.1fs" % (time.time() - t0), flush=True)
g = json.load(open(GOLDEN))
cases = scenario_cases(g["seed"])
for (name, c), want in zip(cases, g["texts"]):
    t = time.time()
    print("start", name, flush=True)
    got = mine.run(c)
    print("%-22s %6.2fs %s" % (name, time.time() - t, "OK" if got == want else "DIFF"), flush=True)
    if got != want:
        for x, y in zip(want.split("\n"), got.split("\n")):
            if x != y:
                print("  want:", x[:300]); print("  got :", y[:300]); break
