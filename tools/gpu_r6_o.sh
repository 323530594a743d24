#!/bin/bash
# round 6, late: the block-size table again on the build with the host overlaps (early stream / early staging) and 16 free slots
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06o
rm -rf $O && mkdir -p $O
cd $R


one() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value=%.0f ms_per_step=%.3f' % (d['value'], d['ms_per_step']), {k:v for k,v in d['kernels_ms_per_step'].items() if k!='note'}, d['pcie']['host_ms_per_step'], d['config']['parity'][:30])"; }
B="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
for i in 1 2 3; do timeout 300 $B 2>/dev/null | one one_block_$i; done
timeout 300 $B --block-loci 5000 2>/dev/null | one b5000_w1
timeout 300 $B --block-loci 2500 2>/dev/null | one b2500_w1
timeout 300 $B --block-loci 5000 --workers 2 2>/dev/null | one b5000_w2
timeout 300 $B --block-loci 2500 --workers 2 2>/dev/null | one b2500_w2
MANTA_AMD_NO_STAGE_GATES=1 timeout 300 $B --block-loci 2500 --workers 2 2>/dev/null | one b2500_w2_free
MANTA_AMD_NO_STAGE_GATES=1 timeout 300 $B --block-loci 5000 --workers 2 2>/dev/null | one b5000_w2_free
