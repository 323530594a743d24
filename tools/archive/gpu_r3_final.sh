#!/bin/bash
# end of round 3: whole GPU tier, smoke, default bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03final
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py > gpurun_out/r03final/bench.json 2> gpurun_out/r03final/bench.err; tail -1 gpurun_out/r03final/bench.json | cut -c1-250
