#!/bin/bash
# chunk counter bumped by hipStreamWriteValue32: single-rank bench (digest parity inside) and two ranks sharing one GPU (gloo)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02t
rm -rf $O && mkdir -p $O
cd $R
MANTA_AMD_DEBUG_STATUS=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/b1.json 2> $O/b1.err
tail -1 $O/b1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('1 rank: value', d['value'], 'ms', d['ms_per_step'], d['pcie']['host_ms_per_step'], d['kernels_ms_per_step']['assemble_kernel'], d['config'].get('parity'))"
grep -c status $O/b1.err
MANTA_BENCH_BACKEND=gloo timeout 200 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/b2.json 2> $O/b2.err
tail -1 $O/b2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2 ranks on one GPU (gloo): value', d['value'], 'ms', d['ms_per_step'], d['pcie']['host_ms_per_step'])"
