#!/bin/bash
# round 6: GPU tier after the bucket sort / refiner changes, the refiner probe at several host thread counts, the default line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06f
rm -rf $O && mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
for t in 16 32 64; do timeout 300 tools/cpp/perf_refiner 10000 0 $t > $O/perf_refiner_small_$t.txt 2>&1; tail -1 $O/perf_refiner_small_$t.txt | cut -c1-900; done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06f/bench_default.json")).read().strip().splitlines()[-1])
print("default: value=%.0f ms_per_step=%.2f" % (d["value"], d["ms_per_step"]), d["kernels_ms_per_step"])
print("refiner_batch:", json.dumps(d.get("refiner_batch")))
print("spanning:", d.get("spanning", {}).get("value"))
PY
