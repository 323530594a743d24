#!/bin/bash
# experiment: walkSlots as a function call under the three-waves-per-SIMD register budget (variant library)
cd "$(dirname "$0")/.."
export MANTA_AMD_ASM_PATH=fast
for team in 4 2; do
  echo "== noinl3 team $team"
  DBG_LIB=manta_amd/variants/lib_noinl3.so MANTA_AMD_FAST_TEAM=$team timeout 200 python tools/debug_fast.py 2>&1 | grep -E "^team|raised" | cut -c1-330
done
