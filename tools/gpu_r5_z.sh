#!/bin/bash
# round 5: the spanning workload's HBM traffic counters on 2 048 loci, final build (rocprofv3 --pmc, one counter per pass)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05w
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload spanning --loci 2048 --steps 1 --warmup 0 --no-cpu-baseline"
export MANTA_AMD_NO_STREAM_UPLOAD=1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_fetch -o p -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/spanning_pmc_write -o p -- $B > $O/pmc_write.log 2>&1
find $O -name "*.rocpd" -delete
find $O -name "*_kernel_trace.csv" -size +8M -delete
ls $O/spanning_pmc_fetch $O/spanning_pmc_write 2>&1 | head
