#!/bin/bash
# round 3: general kernel emits its results from registers (no dependent slab loads per contig): digests + step time
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_digests.py tests/test_assemble_kernels.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'])"
done
