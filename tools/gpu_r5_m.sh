#!/bin/bash
# round 5: kernel stats of the spanning workload with the packed jump aligner (16 384 loci)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05m
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o span -- python $R/bench.py --workload spanning --loci 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/line.json 2> $O/err.txt
find $O -name "*.rocpd" -delete
find $O -name "*kernel_stats.csv" | head -1 | xargs -r head -14 | cut -c1-150
