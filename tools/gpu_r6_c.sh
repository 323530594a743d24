#!/bin/bash
# round 6: CUs reserved for the word-length rounds under the early alignment pass (4 096 / 16 384 loci), the CPU baseline's scaling table,
# the whole-refiner probe with both plans
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06c
rm -rf $O && mkdir -p $O
cd $R
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
for r in 96 128 160 192 224; do run on_16k_r$r 16384 MANTA_AMD_EARLY_RESERVE_CUS=$r; done
run off_4k 4096 MANTA_AMD_EARLY_ALIGN=0
for r in 64 128 192; do run on_4k_r$r 4096 MANTA_AMD_EARLY_RESERVE_CUS=$r; done
run on_64k_r160 65536 MANTA_AMD_EARLY_RESERVE_CUS=160
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06c/bench_default.json")).read().strip().splitlines()[-1])
print("default: value=%.0f ms_per_step=%.2f" % (d["value"], d["ms_per_step"]))
print("cpu_baseline:", json.dumps(d.get("cpu_baseline"), indent=1))
print("refiner_batch:", json.dumps(d.get("refiner_batch"), indent=1))
print("spanning:", d.get("spanning", {}).get("value"), d.get("spanning", {}).get("roofline"))
print("mixed:", d.get("mixed_shape"))
PY
