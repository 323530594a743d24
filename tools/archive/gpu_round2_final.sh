#!/bin/bash
# end-of-round check: GPU tier, the default bench line (copied to profiles/ by the caller), 2 ranks on one GPU over gloo
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02final
rm -rf $O && mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 400 $O/bench_line.json; echo
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
MANTA_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_2rank.json 2> $O/bench_2rank.err; python - $O/bench_2rank.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("2 ranks (one GPU, gloo): n_gpus", d["n_gpus"], "value", d["value"], "gather MB", d["pcie"]["gather_MB_per_step"], d["config"].get("parity"))
except Exception as e: print("2-rank FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-500:])
PY
