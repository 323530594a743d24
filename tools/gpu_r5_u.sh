#!/bin/bash
# round 5: stretch lists in contig_big_kernel / wider speculation in graph_big_kernel -- parity, then the spanning workload by block size, rounds on / off
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05u
rm -rf $O && mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_assemble_kernels.py -x -q -m gpu -k "big_class" > $O/pytest_big.txt 2>&1
tail -2 $O/pytest_big.txt
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
run 65536 MANTA_AMD_BIG_ROUNDS=0
run 65536 MANTA_AMD_BIG_ROUNDS=1
run 16384 MANTA_AMD_BIG_ROUNDS=1
run 16384 MANTA_AMD_BIG_ROUNDS=0
run 32768 MANTA_AMD_BIG_ROUNDS=1
