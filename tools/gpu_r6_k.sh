#!/bin/bash
# round 6: repeat_big_kernel at four waves per SIMD (-DMANTA_RPB_OCC=4, manta_amd/libmanta_amd_occ4.so) against two, spanning workload
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
B="python $R/bench.py --workload spanning --no-cpu-baseline --no-extras"
one() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value=%.0f ms_per_step=%.1f' % (d['value'], d['ms_per_step']), {k:v for k,v in d['kernels_ms_per_step'].items() if k!='note'}, d['config']['parity'][:30])"; }
for l in 16384 65536; do
  timeout 400 $B --loci $l --steps 3 --warmup 1 2>/dev/null | one occ2_$l
  MANTA_AMD_LIB=$R/manta_amd/libmanta_amd_occ4.so timeout 400 $B --loci $l --steps 3 --warmup 1 2>/dev/null | one occ4_$l
done
