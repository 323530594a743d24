#!/bin/bash
# round 6: the GPU tier on the library built from the split host sources; host-side timing of the metric's step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06g
rm -rf $O && mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
MANTA_AMD_DEBUG_TIMING=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_timing.json 2> $O/bench_timing.err
grep "manta_amd:" $O/bench_timing.err | tail -12
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run: value=%.0f ms_per_step=%.3f' % (d['value'], d['ms_per_step']), {k:v for k,v in d['kernels_ms_per_step'].items() if k!='note'}, d['pcie']['host_ms_per_step'])"; done
