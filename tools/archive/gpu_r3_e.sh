#!/bin/bash
# fast + general kernels side by side on one queue: digests in all three modes, bench A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03e
rm -rf $O && mkdir -p $O
cd $R
for path in both fast general; do
  export MANTA_AMD_ASM_PATH=$path
  timeout 600 python -m pytest tests/test_digests.py tests/test_assemble_kernels.py -m gpu -x -q > $O/pytest_$path.log 2>&1
  echo "$path: $(tail -1 $O/pytest_$path.log)"
  MANTA_AMD_DEBUG=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$path.json 2> $O/bench_$path.err
  tail -1 $O/bench_$path.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$path: value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'], d['config'].get('parity'))"
  grep -m1 "assemble_fast_kernel:" $O/bench_$path.err
done
