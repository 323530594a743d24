#!/bin/bash
# round 6, last pass: the GPU tier and the default bench line of the final build (the line goes to profiles/r06_bench_line.json)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06last
rm -rf $O && mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 400 $O/bench_line.json
