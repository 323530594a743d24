#!/bin/bash
# round 3: read-gathering kernels + the whole GPU tier + bench after the team-kernel / read-class changes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03p
timeout 900 python -m pytest tests/test_read_class.py tests/test_host_adapter.py tests/test_capi.py -m gpu -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_read_class.py 2>&1 | tail -5
timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r03p/bench.json 2> gpurun_out/r03p/bench.err; tail -1 gpurun_out/r03p/bench.json | cut -c1-400
