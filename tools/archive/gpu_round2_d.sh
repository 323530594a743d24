#!/bin/bash
# streamed upload: parity + A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02d
rm -rf $O && mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_digests.py tests/test_batch_calls.py tests/test_read_pile.py tests/test_dropin.py tests/test_vcf_candidate.py tests/test_demo_real_data.py -m gpu -q > $O/gputests.log 2>&1
tail -4 $O/gputests.log
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["pcie"]["host_ms_per_step"], "asm", d["kernels_ms_per_step"]["assemble_kernel"], "align", d["kernels_ms_per_step"]["align_kernels"], "packed", d.get("packed_input",{}).get("value"), d.get("packed_input",{}).get("ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/$tag.json 2> $O/$tag.err; show $O/$tag.json $tag; }
run stream_default python bench.py --no-cpu-baseline
MANTA_AMD_NO_STREAM_UPLOAD=1 run nostream python bench.py --no-cpu-baseline --no-extras
run stream_20k python bench.py --no-cpu-baseline --no-extras --loci 20000
run stream_20k_2blocks python bench.py --no-cpu-baseline --no-extras --loci 20000 --block-loci 10000 --workers 2
MANTA_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --loci 4000 --no-cpu-baseline > $O/gloo2.json 2> $O/gloo2.err; tail -c 700 $O/gloo2.json; tail -3 $O/gloo2.err
