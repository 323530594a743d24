#!/bin/bash
# what bounds assemble_kernel: scattered-gather ceiling of the chip + the kernel's own waves-per-CU curve
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02f
rm -rf $O && mkdir -p $O
cd $R
timeout 300 tools/microbench/gather_ceiling > $O/gather_ceiling.txt 2> $O/gather_ceiling.err
cat $O/gather_ceiling.txt
for w in 4 8 12 16; do
  MANTA_AMD_ASM_WAVES_PER_CU=$w timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/wpc$w.json 2> $O/wpc$w.err
  python - $O/wpc$w.json $w <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("waves/CU", sys.argv[2], "value", d["value"], "ms", d["ms_per_step"], "asm", d["kernels_ms_per_step"]["assemble_kernel"])
except Exception as e: print("FAILED", sys.argv[2], e)
PY
done
