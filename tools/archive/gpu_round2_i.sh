#!/bin/bash
# soak: concurrent blocks (2 workers x 5000 loci, 4 x 2500) for many steps; any per-item device status is printed
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02i
rm -rf $O && mkdir -p $O
cd $R
export MANTA_AMD_DEBUG_STATUS=1
S="python bench.py --warmup 2 --no-cpu-baseline --no-extras"
for cfg in "plain2x5000:--steps 300 --block-loci 5000 --workers 2" "plain4x2500:--steps 300 --block-loci 2500 --workers 4" "gates2x5000:--steps 200 --block-loci 5000 --workers 2 --pipeline-stages" "plain3x3334:--steps 200 --block-loci 3334 --workers 3"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  timeout 250 $S $fl > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$? statuses=$(grep -c 'status' $O/$tag.err)"; grep "status" $O/$tag.err | sort | uniq -c | sort -rn | head -8
done
