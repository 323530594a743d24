#!/bin/bash
# round 6: graph_big_kernel's full-length radix sort -- phase shares on plain / tandem piles, big-class parity on hardware, spanning rates
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06i
rm -rf $O && mkdir -p $O
cd $R
python tools/perf_big_rounds.py 2048 2>&1 | grep "phase share\|tandem_frac" | grep -v "over 128 loci" | cut -c1-400
timeout 900 python -m pytest tests/test_digests.py tests/test_assemble_kernels.py tests/test_spanning_pipeline.py -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -2 $O/pytest_gpu.txt
B="python $R/bench.py --workload spanning --no-cpu-baseline --no-extras"
for l in 16384 65536; do timeout 400 $B --loci $l --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spanning $l: value=%.0f ms_per_step=%.1f' % (d['value'], d['ms_per_step']), {k:v for k,v in d['kernels_ms_per_step'].items() if k!='note'}, d['config']['parity'][:40])"; done
MANTA_AMD_EARLY_ALIGN=0 timeout 400 $B --loci 65536 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spanning 65536 no early pass: value=%.0f ms_per_step=%.1f' % (d['value'], d['ms_per_step']), {k:v for k,v in d['kernels_ms_per_step'].items() if k!='note'})"
