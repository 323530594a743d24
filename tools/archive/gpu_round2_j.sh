#!/bin/bash
# after the gate/auto-block changes: soak of multi-worker calls, the GPU test tier, default bench lines
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02j
rm -rf $O && mkdir -p $O
cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["pcie"]["host_ms_per_step"], "asm", d["kernels_ms_per_step"]["assemble_kernel"], "align", d["kernels_ms_per_step"]["align_kernels"], "parity", d.get("parity"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/$tag.json 2> $O/$tag.err; show $O/$tag.json $tag; }
export MANTA_AMD_DEBUG_STATUS=1
S="python bench.py --warmup 2 --no-cpu-baseline --no-extras"
run soak2x5000 $S --steps 300 --block-loci 5000 --workers 2
run soak4x2500 $S --steps 300 --block-loci 2500 --workers 4
grep -c "status" $O/soak2x5000.err $O/soak4x2500.err
unset MANTA_AMD_DEBUG_STATUS
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
run bench_default python bench.py
run bench_spanning python bench.py --workload spanning --steps 2 --warmup 1
