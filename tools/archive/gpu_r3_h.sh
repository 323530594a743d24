#!/bin/bash
# general kernel with speculative contig rounds: parity (assembler tests, digests) + bench + phase profile + spanning bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03h
rm -rf $O && mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_assemble_kernels.py tests/test_digests.py -m gpu -x -q > $O/pytest.log 2>&1
tail -2 $O/pytest.log
MANTA_AMD_DEBUG_STATUS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('general: value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'], d['config'].get('parity')[-14:])"
timeout 300 python tools/profile_phases.py 10000 > $O/phases.log 2>&1
grep "phase share" $O/phases.log | tail -1
timeout 600 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_span.json 2> $O/bench_span.err
tail -1 $O/bench_span.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spanning: value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'], str(d['config'].get('parity'))[-40:])"
