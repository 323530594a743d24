#!/bin/bash
# round 5: the round's profile (tools/profile_round.sh r05: bench line, kernel stats, counter passes, step trace), the spanning workload's
# line (65 536 loci), kernel stats (16 384 loci) and FETCH / WRITE counter passes (8 192 loci), the candidate-level probe
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $R/tools/profile_round.sh r05 > $R/gpurun_out/profile_round_r05.log 2>&1
O=$R/gpurun_out/prof_r05
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 600 $B --workload spanning --steps 2 --warmup 1 > $O/bench_spanning_line.json 2> $O/bench_spanning.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_spanning -o bench -- $B --workload spanning --loci 16384 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
S="--workload spanning --loci 8192 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
echo 8192 > $O/spanning_pmc_loci.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_fetch -o p -- $B $S > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_write -o p -- $B $S > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/spanning_pmc_sq -o p -- $B $S > /dev/null 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.rocpd" -delete
cd $R
timeout 300 tools/cpp/perf_refiner 10000 0 > $O/perf_refiner.log 2>&1
tail -4 $O/perf_refiner.log | cut -c1-300
tail -c 400 $O/bench_spanning_line.json
ls $O
