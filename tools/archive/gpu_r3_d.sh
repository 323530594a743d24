#!/bin/bash
# quick loop: digests + bench (fast path) + phase profile
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03d
rm -rf $O && mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_digests.py -m gpu -x -q > $O/pytest.log 2>&1
tail -2 $O/pytest.log
MANTA_AMD_DEBUG_STATUS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_fast.json 2> $O/bench_fast.err
tail -1 $O/bench_fast.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fast: value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'], d['config'].get('parity'))"
timeout 300 python tools/profile_phases.py 10000 > $O/phases_fast.log 2>&1
grep "phase share" $O/phases_fast.log | tail -1
