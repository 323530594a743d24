#!/bin/bash
# developer build -DMANTA_ASM_PROFILE -DMANTA_ASM_PROFILE_EXACT: where a tandem locus' exact repeat search spends its clocks
# (printed names map to: pack = insertion sequence + hashes, table = rank inside groups, links / cycle-check = the two
#  unordered_map orders, seed = DFS set-up, walk = DFS, select+emit = everything else of the kernel)
cd "$(dirname "$0")/../.."
timeout 600 python tools/profile_tandem.py 48 2>&1 | grep -E "^==|phase share" | tail -8 | cut -c1-330
