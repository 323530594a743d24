#!/bin/bash
# round 5: tandem piles on a second stream beside the big class' pipeline -- spanning at 16 384 / 65 536 loci with and without, graph workgroup size
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05h
rm -rf $O && mkdir -p $O
cd $R
run() {
  n=$1; shift
  env "$@" MANTA_AMD_DEBUG=1 timeout 400 python bench.py --workload spanning --loci $n --steps 2 --warmup 1 --no-cpu-baseline > $O/out.json 2> $O/err.txt
  python - "$n $*" <<PY
import json,sys
try:
    d=json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "->", d["value"], d["ms_per_step"], d["kernels_ms_per_step"]["assembler_stage"], d["kernels_ms_per_step"]["align_kernels"], d["config"]["parity"][-14:])
except Exception as e:
    print(sys.argv[1], "failed", e, open("$O/err.txt").read()[-400:])
PY
  grep "LDS assembler pipeline" $O/err.txt | tail -1 | cut -c1-420
}
run 16384 X=1
run 16384 MANTA_AMD_NO_TANDEM_OVERLAP=1
run 16384 MANTA_AMD_NO_TANDEM_OVERLAP=1 MANTA_AMD_LGL_WAVES=8
run 65536 X=1
run 65536 MANTA_AMD_NO_TANDEM_OVERLAP=1
