#!/bin/bash
# round 5: repeat_big_kernel at two waves per SIMD (256 VGPRs) instead of four
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05v
rm -rf $O && mkdir -p $O
cd $R
run() {
  n=$1; shift
  env "$@" MANTA_AMD_DEBUG=1 timeout 400 python bench.py --workload spanning --loci $n --steps 2 --warmup 1 --no-cpu-baseline > $O/out.json 2> $O/err.txt
  grep "repeat_big_kernel clocks" $O/err.txt | tail -1 | cut -c1-260
  python - "$n $*" <<PY
import json,sys
try:
    d=json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "->", d["value"], d["ms_per_step"], d["kernels_ms_per_step"]["assembler_stage"], d["kernels_ms_per_step"]["align_kernels"], d["config"]["parity"][-14:])
except Exception as e:
    print(sys.argv[1], "failed", e, open("$O/err.txt").read()[-600:])
PY
}
run 16384 X=1
run 65536 X=1
