#!/bin/bash
# experiment: smaller typical-case workspace slabs (node capacity = bases / div; overflows run again)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for div in 1 2 4 6 8; do
  MANTA_AMD_ASM_NODE_DIV=$div MANTA_AMD_DEBUG=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/b.json 2> /tmp/b.err
  tail -1 /tmp/b.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('div $div: value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step']['assemble_kernel'], d['config'].get('parity')[-14:])"
  grep -m2 "ran again" /tmp/b.err
done
