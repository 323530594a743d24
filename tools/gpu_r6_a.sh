#!/bin/bash
# round 6, first pass on the GPU box: the GPU test tier on the library built from separate translation units, then the early alignment
# pass of the spanning pipeline (MANTA_AMD_EARLY_ALIGN, CUs reserved for the word-length rounds, aligner waves per CU) at 16 384 / 65 536 loci
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06a
rm -rf $O && mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
B="python $R/bench.py --workload spanning --no-cpu-baseline --no-extras"
run() {  # name, loci, env...
  local name=$1 loci=$2; shift 2
  env "$@" timeout 400 $B --loci $loci --steps 3 --warmup 1 > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value=%.0f" % d["value"], "ms_per_step=%.1f" % d["ms_per_step"], "kernels:", d.get("kernels_ms_per_step"), "parity:", d.get("parity"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run off_16k 16384 MANTA_AMD_EARLY_ALIGN=0
run on_16k_r64 16384 MANTA_AMD_EARLY_RESERVE_CUS=64
run on_16k_r0 16384 MANTA_AMD_EARLY_RESERVE_CUS=0
run on_16k_r32 16384 MANTA_AMD_EARLY_RESERVE_CUS=32
run on_16k_r64_w8 16384 MANTA_AMD_EARLY_RESERVE_CUS=64 MANTA_AMD_EARLY_WAVES_PER_CU=8
run off_64k 65536 MANTA_AMD_EARLY_ALIGN=0
run on_64k_r64 65536 MANTA_AMD_EARLY_RESERVE_CUS=64
run on_64k_r32 65536 MANTA_AMD_EARLY_RESERVE_CUS=32
run on_64k_r96 65536 MANTA_AMD_EARLY_RESERVE_CUS=96
run on_64k_r32_w8 65536 MANTA_AMD_EARLY_RESERVE_CUS=32 MANTA_AMD_EARLY_WAVES_PER_CU=8
cd /tmp && export TMPDIR=/tmp
MANTA_AMD_EARLY_RESERVE_CUS=64 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_16k -o t -- $B --loci 16384 --steps 2 --warmup 1 > /dev/null 2>&1
find $O -name "*.rocpd" -delete
ls $O/trace_16k/* | head
