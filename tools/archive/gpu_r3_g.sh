#!/bin/bash
# side-by-side assembler kernels: how many loci of the list's end to leave to the fast kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03g
rm -rf $O && mkdir -p $O
cd $R
for reserve in 0 1024 2048 3072 4096 6144; do
  MANTA_AMD_ASM_RESERVE=$reserve timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$reserve.json 2> $O/bench_$reserve.err
  tail -1 $O/bench_$reserve.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('reserve $reserve: value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step']['assemble_kernel'], d['config'].get('parity')[-14:])"
done
