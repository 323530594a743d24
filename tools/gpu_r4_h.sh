#!/bin/bash
# round 4: fast walk steps: parity, cap variants, phases
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04h
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_assemble_kernels.py tests/test_digests.py -m gpu -x -q -k "fast or config2" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
timeout 80 $B > $O/bench_default.json 2> /dev/null
for V in none; do
  MANTA_AMD_LIB=$R/manta_amd/variants/libmanta_amd_$V.so timeout 80 $B > $O/bench_$V.json 2> /dev/null
done
MANTA_AMD_LG_CLASSES="16384,20480,54272" timeout 80 $B > $O/bench_cls3.json 2> /dev/null
timeout 100 python tools/profile_phases.py 10000 > $O/phases.log 2>&1
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04h")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j["value"], j["ms_per_step"], j["kernels_ms_per_step"]["assemble_kernel"], j["config"]["parity"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -2 $O/phases.log | cut -c1-700; MANTA_AMD_LIB=$R/manta_amd/variants/libmanta_amd_profg.so timeout 100 python tools/profile_phases.py 10000 2>&1 | tail -2 | cut -c1-700
