#!/bin/bash
# round 5: VALU issue ceiling microbenchmark (tools/microbench/valu_ceiling.hip) + the SQ counters of one configuration per instruction
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05valu
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
V=$R/tools/microbench/valu_ceiling
timeout 200 $V > $O/valu_ceiling.txt 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/valu_pmc -o p -- $V 1 > $O/valu_pmc.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/valu_pmc2 -o p -- $V 1 > $O/valu_pmc2.log 2>&1
find $O -name "*.rocpd" -delete
tail -60 $O/valu_ceiling.txt
