#!/bin/bash
# round 5: jump-aligner buckets sharing the wave slots by work (vs one full grid each), proof rate of the small class, big-class traffic
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05k
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
    print(sys.argv[1], "failed", e, open("$O/err.txt").read()[-400:])
PY
}
run 16384 X=1
run 16384 MANTA_AMD_SPAN_GRID_BY_WORK=0
run 65536 X=1
MANTA_AMD_DEBUG=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "LDS assembler pipeline" | tail -2 | cut -c1-330
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/big_fetch -o p -- python $R/tools/perf_big_traffic.py 4096 > $O/big_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/big_write -o p -- python $R/tools/perf_big_traffic.py 4096 > $O/big_write.log 2>&1
find $O -name "*.rocpd" -delete
tail -2 $O/big_fetch.log
