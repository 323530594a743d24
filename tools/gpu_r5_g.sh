#!/bin/bash
# round 5: knobs of the general kernel's workspace on the spanning workload (16 384 loci): node capacity divisor, waves per CU
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05g
rm -rf $O && mkdir -p $O
cd $R
run() {
  env "$@" timeout 300 python bench.py --workload spanning --loci 16384 --steps 2 --warmup 1 --no-cpu-baseline > $O/out.json 2> $O/err.txt
  python - "$*" <<PY
import json,sys
try:
    d=json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "->", d["value"], d["ms_per_step"], d["kernels_ms_per_step"]["assembler_stage"])
except Exception as e:
    print(sys.argv[1], "failed", e, open("$O/err.txt").read()[-300:])
PY
}
run X=1
run MANTA_AMD_ASM_NODE_DIV=4
run MANTA_AMD_ASM_NODE_DIV=8
run MANTA_AMD_ASM_WAVES_PER_CU=8
run MANTA_AMD_ASM_NODE_DIV=4 MANTA_AMD_ASM_WAVES_PER_CU=8
run MANTA_AMD_LG_BIG=0
