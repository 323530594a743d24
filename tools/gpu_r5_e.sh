#!/bin/bash
# round 5: block pipelining for the spanning workload (aligners of one block under the assembler of the next), default line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05e
rm -rf $O && mkdir -p $O
cd $R
for cfg in "1 65536" "2 16384" "2 8192" "3 16384" "4 8192"; do
set -- $cfg
timeout 300 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline --workers $1 --block-loci $2 > $O/span_w$1_b$2.json 2> $O/span_w$1_b$2.err
python - <<PY
import json
try:
    d=json.loads(open("$O/span_w$1_b$2.json").read().strip().splitlines()[-1])
    print("workers $1 block $2:", d["value"], d["ms_per_step"], d["kernels_ms_per_step"])
except Exception as e:
    print("workers $1 block $2: failed", e)
PY
done
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d.get("spanning"))
PY
