#!/bin/bash
# third GPU pass of round 2: full GPU tier, HBM-path block/worker sweep incl. serial kernels, phase profile, spanning
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02c
rm -rf $O && mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1
tail -6 $O/gputests.log
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["pcie"]["host_ms_per_step"], "asm", d["kernels_ms_per_step"]["assemble_kernel"], "align", d["kernels_ms_per_step"]["align_kernels"], "d2hMB", d["pcie"]["d2h_MB_per_step"], "packed", d.get("packed_input",{}).get("value"), d.get("packed_input",{}).get("ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/$tag.json 2> $O/$tag.err; show $O/$tag.json $tag; }
run default python bench.py --no-cpu-baseline
run b5000_w2 python bench.py --no-cpu-baseline --no-extras --block-loci 5000 --workers 2
run b5000_w2_serial python bench.py --no-cpu-baseline --no-extras --block-loci 5000 --workers 2 --serial-kernels
run b3334_w3_serial python bench.py --no-cpu-baseline --no-extras --block-loci 3334 --workers 3 --serial-kernels
run b2500_w2_serial python bench.py --no-cpu-baseline --no-extras --block-loci 2500 --workers 2 --serial-kernels
run n20000_b10000_w2 python bench.py --no-cpu-baseline --no-extras --loci 20000 --block-loci 10000 --workers 2
run n20000_b10000_w2_serial python bench.py --no-cpu-baseline --no-extras --loci 20000 --block-loci 10000 --workers 2 --serial-kernels
run n20000_b20000_w1 python bench.py --no-cpu-baseline --no-extras --loci 20000
MANTA_AMD_PROFILE=1 timeout 200 python tools/profile_phases.py 10000 > $O/phases_hbm.log 2>&1; tail -3 $O/phases_hbm.log
run spanning_default python bench.py --workload spanning --steps 2 --warmup 1
