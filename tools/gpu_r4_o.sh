#!/bin/bash
# round 4: merged packed-aligner launch (align_pair_multi_kernel): parity, bench A/B against per-bucket launches, kernel stats;
# kernel stats of the read-gathering probe
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04o
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_align_kernels.py tests/test_golden.py tests/test_digests.py tests/test_pipeline.py tests/test_batch_calls.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2"
timeout 80 $B > $O/bench_default.json 2> /dev/null
MANTA_AMD_NO_ALIGN_MERGE=1 timeout 80 $B > $O/bench_nomerge.json 2> /dev/null
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rc -o rc -- python $R/tools/bench_read_class.py 150 > $O/read_class.log 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.rocpd" -delete
cd $R
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04o")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j["value"], j["ms_per_step"], j["kernels_ms_per_step"], j["pcie"]["host_ms_per_step"], j["config"]["parity"][:24])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -r head -9 | cut -c1-150
find $O/stats_rc -name "*kernel_stats.csv" | head -1 | xargs -r head -9 | cut -c1-150
tail -4 $O/read_class.log
