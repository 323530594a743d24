#!/bin/bash
# round 5: where a tandem-repeat locus spends its time in assemble_kernel (coarse + the exact repeat search split)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05f
rm -rf $O && mkdir -p $O
cd $R
for lib in libmanta_amd_prof.so libmanta_amd_profx.so; do
MANTA_AMD_ASM_PATH=general MANTA_PROF_LIB=$lib timeout 300 python tools/profile_tandem.py 256 > $O/tandem_$lib.log 2>&1
grep -v "^manta_amd: " $O/tandem_$lib.log | cut -c1-330 | tail -12
done
