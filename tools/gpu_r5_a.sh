#!/bin/bash
# round 5, first pass: (1) VALU issue ceiling microbenchmark + its counters, (2) per-phase clocks of assemble_kernel on config-5
# shaped loci (profile build), (3) the spanning workload at 16 384 loci as the round's starting point
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05a
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
V=$R/tools/microbench/valu_ceiling
timeout 120 $V > $O/valu_ceiling.txt 2>&1
tail -5 $O/valu_ceiling.txt
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/valu_pmc -o p -- $V 1 > $O/valu_pmc.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/valu_pmc2 -o p -- $V 1 > $O/valu_pmc2.log 2>&1
find $O -name "*.rocpd" -delete
cd $R
MANTA_AMD_PROFILE=1 timeout 400 python tools/perf_spanning_phases.py 2048 > $O/phases_c5.log 2>&1
tail -4 $O/phases_c5.log
timeout 500 python bench.py --workload spanning --loci 16384 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_spanning_16k.json 2> $O/bench_spanning_16k.err
tail -c 1500 $O/bench_spanning_16k.json
