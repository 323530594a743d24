#!/bin/bash
# round 4: kernel stats of the split-read scorer and the shadow re-alignment (their GPU tests under rocprofv3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04s
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o sr -- python -m pytest $R/tests/test_split_read.py $R/tests/test_shadow_align.py -m gpu -x -q -p no:cacheprovider --rootdir $R > $O/pytest.log 2>&1
tail -3 $O/pytest.log
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.rocpd" -delete
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-150
