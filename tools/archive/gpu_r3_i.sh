#!/bin/bash
# quick: digests + bench of the default path
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03i
rm -rf $O && mkdir -p $O
cd $R
MANTA_AMD_DEBUG_STATUS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default: value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'], d['config'].get('parity')[-14:])"
