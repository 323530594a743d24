#!/bin/bash
# does a streamed config-5 upload survive rocprofv3 --pmc (serialised dispatches) with 16 free workgroup slots?  2 048 loci, hard limits.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload spanning --loci 2048 --steps 1 --warmup 0 --no-cpu-baseline"
for f in 16 64; do
  s=$(date +%s)
  MANTA_AMD_STREAM_FREE_WGS=$f timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_$f -o p -- $B > /tmp/pmc_$f.log 2>&1
  echo "free=$f rc=$? seconds=$(( $(date +%s) - s ))"; tail -c 300 /tmp/pmc_$f.log | tr '\n' ' ' | cut -c1-300; echo
done
