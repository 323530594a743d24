#!/bin/bash
# round 6: later rounds with both contig LDS classes (MANTA_AMD_BIG_ONE_CLASS=0) against one
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
B="python $R/bench.py --workload spanning --no-cpu-baseline --no-extras"
one() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value=%.0f ms_per_step=%.1f' % (d['value'], d['ms_per_step']), {k:v for k,v in d['kernels_ms_per_step'].items() if k!='note'}, d['config']['parity'][:30])"; }
for l in 16384 65536; do
  timeout 400 $B --loci $l --steps 3 --warmup 1 2>/dev/null | one oneclass_$l
  MANTA_AMD_BIG_ONE_CLASS=0 timeout 400 $B --loci $l --steps 3 --warmup 1 2>/dev/null | one twoclasses_$l
done
MANTA_AMD_DEBUG=1 timeout 300 $B --loci 16384 --steps 1 --warmup 0 2>&1 | grep "rounds (graphs\|big class:\|LDS assembler pipeline" | tail -4 | cut -c1-500
