#!/bin/bash
# round 5: the peel in graph_big_kernel (a graph it empties skips repeat_big_kernel; the component scan stays inside the core) -- parity, then timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05y
rm -rf $O && mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_assemble_kernels.py tests/test_digests.py -x -q -m gpu -k "big_class or config5" > $O/pytest.txt 2>&1
tail -2 $O/pytest.txt
timeout 300 python tools/sweeps/sweep_rounds.py 20000 600 2>&1 | tail -1
run() {
  n=$1; shift
  env "$@" MANTA_AMD_DEBUG=1 timeout 400 python bench.py --workload spanning --loci $n --steps 2 --warmup 1 --no-cpu-baseline > $O/out.json 2> $O/err.txt
  grep "repeat_big_kernel clocks\|rounds (" $O/err.txt | tail -2 | cut -c1-300
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
