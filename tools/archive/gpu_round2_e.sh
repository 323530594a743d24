#!/bin/bash
# spanning (config-5 shape) at larger batches: one block vs blocks on 2 / 4 worker pipelines
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02e
rm -rf $O && mkdir -p $O
cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["pcie"]["host_ms_per_step"], "asm", d["kernels_ms_per_step"]["assemble_kernel"], "align", d["kernels_ms_per_step"]["align_kernels"], "loci", d["config"]["loci_per_gpu"])
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
}
run() { tag=$1; shift; timeout 400 "$@" > $O/$tag.json 2> $O/$tag.err; show $O/$tag.json $tag; }
run span64k python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline --loci 65536
run span64k_b16k_w2 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline --loci 65536 --block-loci 16384 --workers 2
run span64k_b8k_w3 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline --loci 65536 --block-loci 8192 --workers 3
run span16k_b4k_w2 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline --loci 16384 --block-loci 4096 --workers 2
run span128k_b32k_w2 python bench.py --workload spanning --steps 1 --warmup 1 --no-cpu-baseline --loci 131072 --block-loci 32768 --workers 2
