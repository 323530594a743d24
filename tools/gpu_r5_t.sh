#!/bin/bash
# round 5: the rounds at 65 536 loci with two contig classes in the later rounds (throughput regime), and rounds off for reference
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05t
rm -rf $O && mkdir -p $O
cd $R
run() {
  n=$1; shift
  env "$@" timeout 400 python bench.py --workload spanning --loci $n --steps 2 --warmup 1 --no-cpu-baseline > $O/out.json 2> $O/err.txt
  python - "$n $*" <<PY
import json,sys
try:
    d=json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "->", d["value"], d["ms_per_step"], d["kernels_ms_per_step"]["assembler_stage"], d["kernels_ms_per_step"]["align_kernels"], d["config"]["parity"][-14:])
except Exception as e:
    print(sys.argv[1], "failed", e, open("$O/err.txt").read()[-600:])
PY
}
run 65536 MANTA_AMD_BIG_ROUNDS=1 MANTA_AMD_BIG_ONE_CLASS=0
run 65536 MANTA_AMD_BIG_ROUNDS=0
run 65536 MANTA_AMD_BIG_ROUNDS=1
