#!/bin/bash
cd "$(dirname "$0")/.."
export MANTA_AMD_ASM_PATH=fast
for v in run1 w2; do
for team in 1 4; do
  echo "== $v team $team"
  DBG_LIB=manta_amd/variants/lib_$v.so MANTA_AMD_FAST_TEAM=$team timeout 200 python tools/debug_fast.py 2>&1 | grep -E "^team|raised" | cut -c1-200
done
done
