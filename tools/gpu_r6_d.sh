#!/bin/bash
# round 6: LDS ceilings (tools/microbench/lds_ceiling.hip) with their LDS counters, the GPU test tier, the demo pile dump replayed on the
# device (tools/replay_piles.py), the default bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06d
rm -rf $O && mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/microbench/lds_ceiling tools/microbench/lds_ceiling.hip 2> $O/lds_build.err
timeout 200 tools/microbench/lds_ceiling > $O/lds_ceiling.txt 2>&1
cat $O/lds_ceiling.txt
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/lds_pmc -o p -- $R/tools/microbench/lds_ceiling gather > /dev/null 2>&1)
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 300 python tools/replay_piles.py tests/golden/demo_pile_dump.txt.gz --check ref > $O/demo_pile_histogram.txt 2>&1
cat $O/demo_pile_histogram.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06d/bench_default.json")).read().strip().splitlines()[-1])
print("default: value=%.0f ms_per_step=%.2f" % (d["value"], d["ms_per_step"]), d["kernels_ms_per_step"])
print("refiner_batch:", json.dumps(d.get("refiner_batch")))
print("spanning:", d.get("spanning", {}).get("value"))
print("mixed:", d.get("mixed_shape"))
PY
find $O -name "*.rocpd" -delete
