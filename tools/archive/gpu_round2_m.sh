#!/bin/bash
# host-side split of one batch call (MANTA_AMD_DEBUG_TIMING) + the plain bench + batch-call tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02m
rm -rf $O && mkdir -p $O
cd $R
MANTA_AMD_DEBUG_TIMING=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_t.json 2> $O/bench_t.err
grep "manta_amd:" $O/bench_t.err | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err
python - $O/bench$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("value", d["value"], "ms", d["ms_per_step"], d["pcie"]["host_ms_per_step"], d["config"].get("parity"))
PY
done
timeout 600 python -m pytest tests/test_batch_calls.py tests/test_digests.py -m gpu -x -q 2>&1 | tail -2
