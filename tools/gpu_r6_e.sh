#!/bin/bash
# round 6: concurrent workers on the spanning workload (blocks of 16 384 / 32 768, with and without the stage gates), the whole-refiner probe
# on the whole-batch calls, the LDS ceilings once more (16-wave workgroups fixed)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06e
rm -rf $O && mkdir -p $O
cd $R
B="python $R/bench.py --workload spanning --no-cpu-baseline --no-extras"
run() {  # name, args..., env via leading VAR=val
  local name=$1; shift
  local envs=()
  while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
  env "${envs[@]}" timeout 400 $B --loci 65536 --steps 3 --warmup 1 "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value=%.0f" % d["value"], "ms_per_step=%.1f" % d["ms_per_step"], "kernels:", {k: v for k, v in d.get("kernels_ms_per_step").items() if k != "note"}, d["config"]["parity"][:40])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run one_block
run gates_w2_b32k --workers 2 --block-loci 32768
run gates_w2_b16k --workers 2 --block-loci 16384
run free_w2_b32k MANTA_AMD_NO_STAGE_GATES=1 --workers 2 --block-loci 32768
run free_w2_b16k MANTA_AMD_NO_STAGE_GATES=1 --workers 2 --block-loci 16384
run free_w4_b16k MANTA_AMD_NO_STAGE_GATES=1 --workers 4 --block-loci 16384
run free_w3_b8k MANTA_AMD_NO_STAGE_GATES=1 --workers 3 --block-loci 8192
run free_w2_b16k_noearly MANTA_AMD_NO_STAGE_GATES=1 MANTA_AMD_EARLY_ALIGN=0 --workers 2 --block-loci 16384
run free_w4_b16k_noearly MANTA_AMD_NO_STAGE_GATES=1 MANTA_AMD_EARLY_ALIGN=0 --workers 4 --block-loci 16384
timeout 300 tools/cpp/perf_refiner 10000 0 > $O/perf_refiner_small.txt 2>&1; tail -1 $O/perf_refiner_small.txt
timeout 300 tools/cpp/perf_refiner 2000 1 > $O/perf_refiner_span.txt 2>&1; tail -1 $O/perf_refiner_span.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/microbench/lds_ceiling tools/microbench/lds_ceiling.hip 2> $O/lds_build.err
timeout 200 tools/microbench/lds_ceiling > $O/lds_ceiling.txt 2>&1
cat $O/lds_ceiling.txt
