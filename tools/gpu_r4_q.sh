#!/bin/bash
# round 4, last pass over the final build: the bench line, kernel stats and step trace again (into gpurun_out/prof_r04, next to the
# counter passes of tools/profile_round.sh), with SPANNING_PMC=1 the spanning workload's counter passes on 8 192 loci (the 65 536-locus
# passes ran into their time limit under the profiler), then the whole GPU tier
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r04
mkdir -p $O
B="python $R/bench.py"
timeout 300 $B > $O/bench_line.json 2> $O/bench.err
tail -c 300 $O/bench_line.json | cut -c1-300
rm -rf $O/stats $O/trace_step
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> /dev/null
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_step -o t -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
if [ -n "$SPANNING_PMC" ]; then
S="--workload spanning --loci 8192 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
echo 8192 > $O/spanning_pmc_loci.txt
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_fetch -o p -- $B $S > /dev/null 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_write -o p -- $B $S > /dev/null 2>&1
fi
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.rocpd" -delete
ls $O/spanning_pmc_fetch $O/spanning_pmc_write 2>&1 | head -8
cd $R
timeout 500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r04q_pytest.log 2>&1
tail -3 $R/gpurun_out/r04q_pytest.log
