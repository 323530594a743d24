#!/bin/bash
# round 4, first GPU run of the LDS pipeline (graph_kernel -> contig_kernel): parity on hardware, bench, phases, kernel stats
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04a
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_assemble_kernels.py tests/test_digests.py -m gpu -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-extras"
timeout 300 $B > $O/bench_lds.json 2> $O/bench_lds.err
MANTA_AMD_ASM_PATH=general timeout 300 $B > $O/bench_general.json 2> $O/bench_general.err
for C in "23040,27136,32768,54272" "32768,54272" "27136,54272" "23040,54272"; do
  MANTA_AMD_LG_CLASSES=$C timeout 300 $B --steps 6 > $O/bench_cls_$C.json 2> /dev/null
done
timeout 300 python tools/profile_phases.py 10000 > $O/phases.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --steps 5 --warmup 1 > /dev/null 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.rocpd" -delete
cd $R
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04a")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j["value"], j["ms_per_step"], j["kernels_ms_per_step"], j["config"]["parity"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -3 $O/phases.log
head -12 $O/stats/*kernel_stats.csv 2>/dev/null | cut -c1-200
