#!/bin/bash
# round 4: the whole GPU tier, then the profile round (tools/profile_round.sh r04, default workload) and the read-gathering probe
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04n
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1
tail -14 $O/pytest.log
timeout 1000 bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1
tail -5 $O/profile_round.log | cut -c1-400
timeout 200 python tools/bench_read_class.py 150 > $O/read_class.log 2>&1
tail -4 $O/read_class.log
