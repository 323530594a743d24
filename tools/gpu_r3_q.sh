#!/bin/bash
# round 3: fewer host wake-ups per step (staging behind the run, aligner buckets from history): parity + step time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03q
timeout 900 python -m pytest tests/test_batch_calls.py tests/test_digests.py tests/test_node.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -3
for mode in "" "MANTA_AMD_SYNC_BUCKETS=1"; do
  echo "== $mode"
  env $mode timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms', d['ms_per_step'], d['kernels_ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r03q/trace -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r03q/trace/*/ | head
