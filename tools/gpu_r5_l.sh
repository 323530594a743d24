#!/bin/bash
# round 5: packed jump aligner pairs on hardware -- parity (pair tests, config-5 digests), then the spanning workload with and without
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05l
rm -rf $O && mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_align_kernels.py tests/test_digests.py tests/test_spanning_pipeline.py -m gpu -x -q -k "jump_pairs or config5 or spanning or align_random or align_long" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
run() {
  n=$1; shift
  env "$@" timeout 400 python bench.py --workload spanning --loci $n --steps 2 --warmup 1 --no-cpu-baseline > $O/out.json 2> $O/err.txt
  python - "$n $*" <<PY
import json,sys
try:
    d=json.loads(open("$O/out.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "->", d["value"], d["ms_per_step"], d["kernels_ms_per_step"]["assembler_stage"], d["kernels_ms_per_step"]["align_kernels"], d["dp_gcups"], d["config"]["parity"][-14:])
except Exception as e:
    print(sys.argv[1], "failed", e, open("$O/err.txt").read()[-400:])
PY
}
run 16384 X=1
run 16384 MANTA_AMD_NO_JUMP_PAIRS=1
run 65536 X=1
