#!/bin/bash
# per-phase shader clocks of assemble_fast_kernel (profile build)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03b
rm -rf $O && mkdir -p $O
cd $R
timeout 300 python tools/profile_phases.py 10000 > $O/phases_fast.log 2>&1
grep "phase share" $O/phases_fast.log | tail -1
MANTA_AMD_ASM_PATH=general timeout 300 python tools/profile_phases.py 10000 > $O/phases_general.log 2>&1
grep "phase share" $O/phases_general.log | tail -1
