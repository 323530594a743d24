#!/bin/bash
# round 4: blocks / workers of the whole-batch call with the LDS assembler (does the aligner of block A hide behind the assembler of block B now?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04g
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2"
for W in "1 0" "2 5000" "2 2500" "3 2500" "4 2500" "2 3334" "3 3334"; do
  set -- $W
  timeout 80 $B --workers $1 --block-loci $2 > $O/bench_w$1_b$2.json 2> /dev/null
done
MANTA_AMD_NO_STREAM_UPLOAD=1 timeout 80 $B --workers 2 --block-loci 5000 > $O/bench_w2_b5000_nostream.json 2> /dev/null
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04g")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j["value"], j["ms_per_step"], j["kernels_ms_per_step"], j["config"]["parity"][:20])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
