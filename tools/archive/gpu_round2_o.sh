#!/bin/bash
# streamed upload in the spanning batch call
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02o
rm -rf $O && mkdir -p $O
cd $R
export MANTA_AMD_DEBUG_STATUS=1
for tag in span1 span2; do
timeout 600 python bench.py --workload spanning --steps 3 --warmup 1 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
python - $O/$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("spanning value", d["value"], "ms", d["ms_per_step"], d["pcie"]["host_ms_per_step"], d["kernels_ms_per_step"]["assemble_kernel"], d["config"].get("parity"))
PY
done
MANTA_AMD_NO_STREAM_UPLOAD=1 timeout 600 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline > $O/nostream.json 2> $O/nostream.err
python - $O/nostream.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("no-stream value", d["value"], "ms", d["ms_per_step"], d["pcie"]["host_ms_per_step"], d["kernels_ms_per_step"]["assemble_kernel"])
PY
grep -c status $O/*.err
unset MANTA_AMD_DEBUG_STATUS
timeout 900 python -m pytest tests/test_batch_calls.py tests/test_spanning_pipeline.py tests/test_digests.py -m gpu -x -q 2>&1 | tail -2
