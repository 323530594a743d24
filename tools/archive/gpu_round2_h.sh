#!/bin/bash
# stage-gated block pipeline (MANTA_BATCH_PIPELINE_STAGES) on the spanning workload + soak of concurrent blocks
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02h
rm -rf $O && mkdir -p $O
cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["pcie"]["host_ms_per_step"], "asm", d["kernels_ms_per_step"]["assemble_kernel"], "align", d["kernels_ms_per_step"]["align_kernels"])
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/$tag.json 2> $O/$tag.err; show $O/$tag.json $tag; }
export MANTA_AMD_DEBUG_STATUS=1
S="python bench.py --warmup 2 --no-cpu-baseline --no-extras"
run soak2x5000 $S --steps 300 --block-loci 5000 --workers 2
run soak3x3334 $S --steps 200 --block-loci 3334 --workers 3
grep -c "status" $O/soak2x5000.err $O/soak3x3334.err
C="python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline"
run span_p2_16k $C --block-loci 16384 --workers 2 --pipeline-stages
run span_p2_32k $C --block-loci 32768 --workers 2 --pipeline-stages
MANTA_AMD_PIPELINED_ASM_WAVES=14 run span_p2_16k_w14 $C --block-loci 16384 --workers 2 --pipeline-stages
MANTA_AMD_PIPELINED_ASM_WAVES=8 run span_p2_16k_w8 $C --block-loci 16384 --workers 2 --pipeline-stages
