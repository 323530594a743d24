#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02q
rm -rf $O && mkdir -p $O
cd $R
for i in 1 2 3; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err
python - $O/bench$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("value", d["value"], "ms", d["ms_per_step"], d["pcie"]["host_ms_per_step"], d["kernels_ms_per_step"]["assemble_kernel"], d["config"].get("parity"))
PY
done
timeout 300 python -m pytest tests/test_assemble_kernels.py -m gpu -x -q 2>&1 | tail -2
