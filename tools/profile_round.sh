#!/bin/bash
# Round profile on the GPU box (run through gpurun): the default bench line, rocprofv3 kernel stats of the same command,
# and PMC passes (each counter group in its own run; never combined with sys/hip/hsa traces).  Every step has its own
# timeout.  Outputs land in gpurun_out/; tools/summarize_profiles.py turns them into the files under profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
B="python $R/bench.py"
timeout 400 $B > $O/bench_line.json 2> $O/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> /dev/null
P="--steps 1 --warmup 0 --no-cpu-baseline --no-extras"
if [ -z "$PROFILE_LINES_ONLY" ]; then  # PROFILE_LINES_ONLY=1: bench line, kernel stats and the step trace only (the counter passes are kept)
# counter passes use the blocking upload: under the profiler the copies of a streamed upload are delayed and the assembler's
# chunk-wait loop (s_sleep polling) would dominate every SQ_* counter; the kernel's work is the same
export MANTA_AMD_NO_STREAM_UPLOAD=1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $B $P > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $B $P > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_sq -o p -- $B $P > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- $B $P > /dev/null 2>&1
unset MANTA_AMD_NO_STREAM_UPLOAD
fi
# one plain kernel trace of the default step: every dispatch with its start / end (the gaps are the host turnarounds)
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_step -o t -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
if [ "$2" = "spanning" ]; then
  timeout 600 $B --workload spanning --steps 2 --warmup 1 > $O/bench_spanning_line.json 2> $O/bench_spanning.err
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_spanning -o bench -- $B --workload spanning --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  # (the config-5 shape's traffic counters: tools/gpu_r6_final.sh, on the 2 048 digest loci with the blocking upload -- a streamed upload under
  #  rocprofv3 --pmc never finishes: dispatches are serialised, the copy kernels wait for the assembler, which waits for its chunks)
fi
if [ "$2" = "spanning" ] || [ "$3" = "read_class" ] || [ "$2" = "read_class" ]; then
  # read gathering (manta_read_piles_batch) as a measured component: its line, kernel stats, counters in separate passes
  RC="python $R/tools/bench_read_class.py 150"
  timeout 200 $RC > $O/read_class.log 2>&1
  tail -1 $O/read_class.log > $O/read_class_line.json
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rc_stats -o rc -- $RC > /dev/null 2>&1
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/rc_pmc_fetch -o p -- $RC > /dev/null 2>&1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/rc_pmc_write -o p -- $RC > /dev/null 2>&1
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/rc_pmc_sq -o p -- $RC > /dev/null 2>&1
fi
# keep only the small files (gpurun_out merges back <= 64 MiB)
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.rocpd" -delete
ls -la $O $O/stats 2>/dev/null | head -40
tail -c 600 $O/bench_line.json
