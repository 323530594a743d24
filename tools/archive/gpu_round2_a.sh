#!/bin/bash
# first GPU pass of round 2: the GPU test tier, the new PCIe-inclusive bench line, a block/worker sweep
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02a
rm -rf $O && mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
tail -5 $O/gputests.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
for cfg in "1250 4" "1250 8" "2500 4" "625 8" "5000 2" "10000 1"; do
  set -- $cfg
  timeout 200 python bench.py --no-cpu-baseline --steps 5 --block-loci $1 --workers $2 > $O/bench_b$1_w$2.json 2> $O/bench_b$1_w$2.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_b$1_w$2.json"))
    print("block $1 workers $2: value", d["value"], "ms/step", d["ms_per_step"], d["pcie"]["host_ms_per_step"], d["kernels_ms_per_step"])
except Exception as e:
    print("block $1 workers $2: FAILED", e)
PY
done
timeout 200 python bench.py --no-cpu-baseline --steps 5 --pageable > $O/bench_pageable.json 2> $O/bench_pageable.err
python -c "
import json; d=json.load(open('$O/bench_pageable.json')); print('pageable: value', d['value'], d['pcie'])"
timeout 400 python bench.py --workload spanning --steps 2 --warmup 1 > $O/bench_spanning.json 2> $O/bench_spanning.err
tail -c 1200 $O/bench_spanning.json; tail -3 $O/bench_spanning.err
