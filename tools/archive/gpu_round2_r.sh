#!/bin/bash
# streamed packed piles: bench with the packed_input leg + pile tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02r
rm -rf $O && mkdir -p $O
cd $R
export MANTA_AMD_DEBUG_STATUS=1
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench$i.json 2> $O/bench$i.err
python - $O/bench$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("value", d["value"], "ms", d["ms_per_step"], d["pcie"]["host_ms_per_step"], d["kernels_ms_per_step"]["assemble_kernel"], "| packed", d["packed_input"]["value"], d["packed_input"]["ms_per_step"], "| kernel_only", d["kernel_only"]["value"])
PY
done
grep -c status $O/*.err
unset MANTA_AMD_DEBUG_STATUS
timeout 600 python -m pytest tests/test_read_pile.py tests/test_batch_calls.py tests/test_demo_real_data.py -m gpu -x -q 2>&1 | tail -2
