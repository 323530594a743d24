#!/bin/bash
# round 6, final pass on the GPU box: the default bench line + kernel stats + counter passes + step trace and the spanning line / stats
# (tools/profile_round.sh r06 spanning), the spanning workload's traffic counters on the 2 048 digest loci, the per-dispatch trace of a
# 16 384-locus spanning block with the early alignment pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
bash tools/profile_round.sh r06 spanning > $R/gpurun_out/profile_round_r06.log 2>&1
tail -c 300 $R/gpurun_out/prof_r06/bench_line.json; echo
O=$R/gpurun_out/r06w
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload spanning --loci 2048 --steps 1 --warmup 0 --no-cpu-baseline"
MANTA_AMD_NO_STREAM_UPLOAD=1 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_fetch -o p -- $B > $O/pmc_fetch.log 2>&1
MANTA_AMD_NO_STREAM_UPLOAD=1 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_write -o p -- $B > $O/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_16k -o t -- python $R/bench.py --workload spanning --loci 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
find $O -name "*.rocpd" -delete
find $O -name "*_kernel_trace.csv" -size +12M -delete
ls $O/*
