#!/bin/bash
# whole GPU tier + default bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03j
rm -rf $O && mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-600
