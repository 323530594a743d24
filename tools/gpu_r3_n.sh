#!/bin/bash
# round 3: the fast assembler as cooperative workgroups (teams of 1 / 2 / 4 wavefronts per locus): parity, then step and kernel times
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export MANTA_AMD_ASM_PATH=fast
timeout 600 python -m pytest tests/test_digests.py tests/test_assemble_kernels.py -m gpu -x -q -k "fast or digest or side" 2>&1 | tail -3
for team in 4 2 1; do
  echo "== team $team"
  MANTA_AMD_FAST_TEAM=$team timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['avg_launch_ms'], d.get('parity'))"
done
echo "== both, team 4"
MANTA_AMD_ASM_PATH=both timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-200
echo "== phases team 4"
timeout 300 python tools/profile_phases.py 2>&1 | tail -14
