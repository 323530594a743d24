#!/bin/bash
# round 3, first GPU visit: assemble_fast_kernel parity (assembler tests + full-size digests), bench A/B fast vs general
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03a
rm -rf $O && mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_assemble_kernels.py tests/test_digests.py tests/test_golden.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for path in fast general; do
  export MANTA_AMD_ASM_PATH=$path
  MANTA_AMD_DEBUG_STATUS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$path.json 2> $O/bench_$path.err
  tail -1 $O/bench_$path.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$path: value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'], d['config'].get('parity'))"
done
unset MANTA_AMD_ASM_PATH
MANTA_AMD_DEBUG=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -m3 "assemble_fast_kernel:"
