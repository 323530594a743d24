#!/bin/bash
# round 5, final pass on the GPU box: the whole GPU test tier, the default bench line + kernel stats + step trace (tools/profile_round.sh),
# the spanning workload at 16 384 loci (line, kernel stats, per-dispatch trace of the word-length rounds) and its HBM traffic counters at 4 096
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05final
rm -rf $O && mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
PROFILE_LINES_ONLY=1 bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1
tail -c 400 $R/gpurun_out/prof_r05/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload spanning"
timeout 300 $B --loci 16384 --steps 2 --warmup 1 --no-cpu-baseline > $O/spanning_16k_line.json 2> $O/spanning_16k.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_spanning -o bench -- $B --loci 16384 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_fetch -o p -- $B --loci 4096 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_write -o p -- $B --loci 4096 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
find $O -name "*.rocpd" -delete
find $O -name "*_kernel_trace.csv" -size +12M -delete
tail -c 300 $O/spanning_16k_line.json; echo
ls $O $O/stats_spanning/* | head -30
