#!/bin/bash
# experiment: the fully parallel team build under the 168-register budget with backend options (variant libraries)
cd "$(dirname "$0")/.."
export MANTA_AMD_ASM_PATH=fast
for v in w3 w3a w3b; do
  [ -f manta_amd/variants/lib_$v.so ] || continue
  echo "== $v team 4"
  DBG_LIB=manta_amd/variants/lib_$v.so MANTA_AMD_FAST_TEAM=4 timeout 200 python tools/debug_fast.py 2>&1 | grep -E "^team|raised" | cut -c1-300
done
