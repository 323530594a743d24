#!/bin/bash
# round 6: the early alignment pass on its own hardware queue -- spanning workload, 16 384 / 65 536 loci, CUs reserved for the rounds; host CPU topology
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06b
rm -rf $O && mkdir -p $O
cd $R
(nproc; lscpu | head -25; cat /sys/fs/cgroup/cpu.max; ldconfig -p | grep -i "tcmalloc\|jemalloc") > $O/host.txt 2>&1
B="python $R/bench.py --workload spanning --no-cpu-baseline --no-extras"
run() {  # name, loci, env...
  local name=$1 loci=$2; shift 2
  env "$@" timeout 400 $B --loci $loci --steps 3 --warmup 1 > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value=%.0f" % d["value"], "ms_per_step=%.1f" % d["ms_per_step"], "kernels:", {k: v for k, v in d.get("kernels_ms_per_step").items() if k != "note"}, d["config"]["parity"][:40])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run on_16k_r32 16384 MANTA_AMD_EARLY_RESERVE_CUS=32
run on_16k_r64 16384 MANTA_AMD_EARLY_RESERVE_CUS=64
run on_16k_r16 16384 MANTA_AMD_EARLY_RESERVE_CUS=16
run on_64k_r32 65536 MANTA_AMD_EARLY_RESERVE_CUS=32
run on_64k_r64 65536 MANTA_AMD_EARLY_RESERVE_CUS=64
run on_64k_r128 65536 MANTA_AMD_EARLY_RESERVE_CUS=128
run on_64k_r64_w8 65536 MANTA_AMD_EARLY_RESERVE_CUS=64 MANTA_AMD_EARLY_WAVES_PER_CU=8
cd /tmp && export TMPDIR=/tmp
MANTA_AMD_EARLY_RESERVE_CUS=64 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_16k -o t -- $B --loci 16384 --steps 2 --warmup 1 > /dev/null 2>&1
MANTA_AMD_EARLY_RESERVE_CUS=64 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_64k -o t -- $B --loci 65536 --steps 2 --warmup 1 > /dev/null 2>&1
find $O -name "*.rocpd" -delete
cat $O/host.txt | head -40
