#!/bin/bash
# round 3: the first run of a pipeline launches like the later ones: no transition step; default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03s
timeout 600 python -m pytest tests/test_batch_calls.py tests/test_digests.py -m gpu -x -q 2>&1 | tail -2
for w in 1 3; do
  timeout 300 python bench.py --steps 5 --warmup $w --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('warmup $w: value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'])"
done
timeout 400 python bench.py > gpurun_out/r03s/bench.json 2> gpurun_out/r03s/bench.err; tail -1 gpurun_out/r03s/bench.json | cut -c1-300
