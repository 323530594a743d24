#!/bin/bash
# round 6: contig_kernel LDS size classes (MANTA_AMD_LG_CLASSES) on the metric's workload
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
one() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value=%.0f ms_per_step=%.3f' % (d['value'], d['ms_per_step']), {k:v for k,v in d['kernels_ms_per_step'].items() if k!='note'})"; }
B="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
timeout 300 $B 2>/dev/null | one default_20480_54272
MANTA_AMD_LG_CLASSES=16384,20480,54272 MANTA_AMD_CONTIG_WG_CAP=10 timeout 300 $B 2>/dev/null | one c16384_20480_54272
MANTA_AMD_LG_CLASSES=18432,54272 MANTA_AMD_CONTIG_WG_CAP=8 timeout 300 $B 2>/dev/null | one c18432_54272
MANTA_AMD_LG_CLASSES=17920,20480,54272 MANTA_AMD_CONTIG_WG_CAP=9 timeout 300 $B 2>/dev/null | one c17920_20480_54272
MANTA_AMD_LG_CLASSES=15360,20480,54272 MANTA_AMD_CONTIG_WG_CAP=10 timeout 300 $B 2>/dev/null | one c15360_20480_54272
timeout 300 $B 2>/dev/null | one default_again
