#!/bin/bash
# round 4: walk logs instead of visited-mask atomics, staged pack, 3 radix passes, lockstep lookups: parity, class sweep, phases
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04f
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_assemble_kernels.py tests/test_digests.py -m gpu -x -q -k "fast or config2" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
timeout 80 $B > $O/bench_default.json 2> /dev/null
for C in "16384,20480,54272" "16384,54272" "14336,20480,54272" "12288,16384,20480,54272"; do
  MANTA_AMD_LG_CLASSES=$C timeout 80 $B > $O/bench_cls_$C.json 2> /dev/null
done
MANTA_AMD_LG_NO_PROOF=1 timeout 80 $B > $O/bench_noproof.json 2> /dev/null
timeout 100 python tools/profile_phases.py 10000 > $O/phases.log 2>&1
MANTA_AMD_LIB=$R/manta_amd/variants/libmanta_amd_profg.so timeout 100 python tools/profile_phases.py 10000 > $O/phases_graph.log 2>&1
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > /dev/null 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.rocpd" -delete
cd $R
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04f")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j["value"], j["ms_per_step"], j["kernels_ms_per_step"]["assemble_kernel"], j["config"]["parity"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -2 $O/phases.log | cut -c1-700
tail -2 $O/phases_graph.log | cut -c1-700
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-160
