#!/bin/bash
# round 3: repeat-rich loci at the head of the work queue (plan(): tandem stretch in two sampled reads): spanning batches
cd "$(dirname "$0")/.."
for n in 16384 65536; do
  for mode in "MANTA_AMD_NO_REPEAT_COST=1" ""; do
    env $mode timeout 600 python bench.py --workload spanning --loci $n --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n [$mode] value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'])"
  done
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
