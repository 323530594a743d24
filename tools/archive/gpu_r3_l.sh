#!/bin/bash
# experiment: pack + table + links alone (MANTA_ASM_STOP_AFTER_GRAPH builds) at 4 / 6 / 8 waves per SIMD
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
python tools/ab_probe.py 2>/dev/null | tail -1
MANTA_AMD_ASM_WAVES_PER_CU=16 python tools/ab_probe.py manta_amd/variants/lib_graph4.so 2>/dev/null | tail -1
MANTA_AMD_ASM_WAVES_PER_CU=24 python tools/ab_probe.py manta_amd/variants/lib_graph6.so 2>/dev/null | tail -1
MANTA_AMD_ASM_WAVES_PER_CU=32 python tools/ab_probe.py manta_amd/variants/lib_graph8.so 2>/dev/null | tail -1
MANTA_AMD_ASM_WAVES_PER_CU=24 python tools/ab_probe.py manta_amd/variants/lib_graph8.so 2>/dev/null | tail -1
