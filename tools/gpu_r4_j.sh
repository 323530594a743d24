#!/bin/bash
# round 4: aligner grid size (waves per CU per bucket) with the packed pair kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04j
rm -rf $O && mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2"
for W in 3 4 6 8 12 16; do
  MANTA_AMD_ALIGN_WAVES_PER_CU=$W timeout 80 $B > $O/bench_w$W.json 2> /dev/null
  MANTA_AMD_NO_ALIGN_PAIRS=1 MANTA_AMD_ALIGN_WAVES_PER_CU=$W timeout 80 $B > $O/bench_nopair_w$W.json 2> /dev/null
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04j")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j["value"], j["ms_per_step"], j["kernels_ms_per_step"]["align_kernels"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
