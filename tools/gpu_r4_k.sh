#!/bin/bash
# round 4: what the back-pointer stream costs the packed aligner (developer variant without the stores: timing only, results invalid)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04k
rm -rf $O && mkdir -p $O
cd /tmp
P="python $R/tools/profile_phases.py 10000"
MANTA_AMD_LIB=$R/manta_amd/variants/libmanta_amd_nostore.so timeout 100 $P 2>&1 | tail -1
MANTA_AMD_ALIGN_WAVES_PER_CU=8 MANTA_AMD_LIB=$R/manta_amd/variants/libmanta_amd_nostore.so timeout 100 $P 2>&1 | tail -1
MANTA_AMD_LIB=$R/manta_amd/libmanta_amd.so timeout 100 $P 2>&1 | tail -1
