#!/bin/bash
# redo the counter passes of the config-2 profile with the blocking upload (see tools/profile_round.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r02
mkdir -p $O
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2
B="python $R/bench.py"
P="--steps 1 --warmup 0 --no-cpu-baseline --no-extras"
export MANTA_AMD_NO_STREAM_UPLOAD=1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $B $P > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $B $P > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_sq -o p -- $B $P > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- $B $P > /dev/null 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.rocpd" -delete
ls $O
