#!/bin/bash
# round 5: the whole GPU tier, then the default bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05i
rm -rf $O && mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -6 $O/pytest.log
MANTA_BENCH_SPANNING_LOCI=16384 MANTA_AMD_DEBUG_TIMING=1 timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["pcie"]["host_ms_per_step"], d["config"]["parity"])
for k in ("kernel_only","packed_input","mixed_shape","refiner_batch"):
    print(k, d.get(k))
sp=d.get("spanning",{}); print("spanning", {k: sp.get(k) for k in ("value","ms_per_step","loci","parity","kernels_ms_per_step","error")})
PY
grep "plan:" $O/bench.err | tail -3
