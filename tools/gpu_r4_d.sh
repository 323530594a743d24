#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04d
rm -rf $O && mkdir -p $O
MANTA_AMD_LIB=$R/manta_amd/variants/libmanta_amd_profg.so timeout 100 python tools/profile_phases.py 10000 > $O/phases_graph.log 2>&1
tail -2 $O/phases_graph.log | cut -c1-700
