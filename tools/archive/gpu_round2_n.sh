#!/bin/bash
# kernel change check: bench (digest parity inside) + assembler tests + spanning bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02n
rm -rf $O && mkdir -p $O
cd $R
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err
python - $O/bench$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("value", d["value"], "ms", d["ms_per_step"], d["pcie"]["host_ms_per_step"], d["kernels_ms_per_step"]["assemble_kernel"], d["config"].get("parity"))
PY
done
timeout 600 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline > $O/span.json 2> $O/span.err
python - $O/span.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("spanning value", d["value"], "ms", d["ms_per_step"], d["kernels_ms_per_step"]["assemble_kernel"], d["config"].get("parity"))
PY
timeout 900 python -m pytest tests/test_assemble_kernels.py tests/test_digests.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
