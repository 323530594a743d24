#!/bin/bash
# round 4: contig_kernel LDS classes / workgroups per CU (environment knobs, one build)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04r
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2"
timeout 80 $B > $O/bench_a_default.json 2> /dev/null
MANTA_AMD_CONTIG_WG_CAP=10 MANTA_AMD_LG_CLASSES=16384,20480,54272 timeout 80 $B > $O/bench_b_16k_cap10.json 2> /dev/null
MANTA_AMD_CONTIG_WG_CAP=11 MANTA_AMD_LG_CLASSES=14848,17408,20480,54272 timeout 80 $B > $O/bench_c_14k_17k_cap11.json 2> /dev/null
MANTA_AMD_CONTIG_WG_CAP=10 MANTA_AMD_LG_CLASSES=16384,54272 timeout 80 $B > $O/bench_d_16k_only_cap10.json 2> /dev/null
MANTA_AMD_CONTIG_WG_CAP=9 MANTA_AMD_LG_CLASSES=17920,54272 timeout 80 $B > $O/bench_e_17k5_cap9.json 2> /dev/null
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04r")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j["value"], j["ms_per_step"], j["kernels_ms_per_step"]["assemble_kernel"], j["config"]["parity"][-14:])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
