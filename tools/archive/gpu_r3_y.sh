#!/bin/bash
# round 3: the spanning workload after the dense read sets; whole GPU tier
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03y
timeout 900 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03y/spanning.json 2> gpurun_out/r03y/spanning.err
tail -1 gpurun_out/r03y/spanning.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spanning value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'])"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
