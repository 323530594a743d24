#!/bin/bash
# round 5: phase shares of the big class (profile builds), the spanning line with the wider punt grid
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05d
rm -rf $O && mkdir -p $O
cd $R
timeout 300 python tools/perf_big_phases.py 4096 > $O/phases.log 2>&1
grep -v "^manta_amd: " $O/phases.log | tail -20; grep "phase share" $O/phases.log
MANTA_AMD_DEBUG=1 timeout 300 python bench.py --workload spanning --loci 16384 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_spanning_16k.json 2> $O/bench_spanning_16k.err
python - <<PY
import json
d=json.loads(open("$O/bench_spanning_16k.json").read().strip().splitlines()[-1])
print("16k:", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["config"]["parity"])
PY
timeout 600 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_spanning_64k.json 2> $O/bench_spanning_64k.err
python - <<PY
import json
d=json.loads(open("$O/bench_spanning_64k.json").read().strip().splitlines()[-1])
print("64k:", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["config"]["parity"])
PY
