#!/bin/bash
# second GPU pass of round 2: GPU tier with the LDS-resident assembler, A/B against the HBM path, block/worker sweep
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02b
rm -rf $O && mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1
tail -8 $O/gputests.log
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["pcie"]["host_ms_per_step"], d["kernels_ms_per_step"]["assemble_kernel"], d["kernels_ms_per_step"]["align_kernels"], "d2hMB", d["pcie"]["d2h_MB_per_step"], "kernel_only", d.get("kernel_only",{}).get("kernels_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/$tag.json 2> $O/$tag.err; show $O/$tag.json $tag; }
run lds_b10000_w1 python bench.py --no-cpu-baseline --block-loci 10000 --workers 1
MANTA_AMD_ASM_PATH=hbm run hbm_b10000_w1 python bench.py --no-cpu-baseline --block-loci 10000 --workers 1
MANTA_AMD_LDS_OFF=1 run ldsoff_b10000_w1 python bench.py --no-cpu-baseline --block-loci 10000 --workers 1
run lds_b5000_w2 python bench.py --no-cpu-baseline --block-loci 5000 --workers 2
run lds_b2500_w2 python bench.py --no-cpu-baseline --block-loci 2500 --workers 2
run lds_b2500_w4 python bench.py --no-cpu-baseline --block-loci 2500 --workers 4
run lds_b2500_w3_serial python bench.py --no-cpu-baseline --block-loci 2500 --workers 3 --serial-kernels
run lds_b1250_w4 python bench.py --no-cpu-baseline --block-loci 1250 --workers 4
run lds_b3334_w3 python bench.py --no-cpu-baseline --block-loci 3334 --workers 3
MANTA_AMD_PROFILE=1 timeout 200 python tools/profile_phases.py 10000 > $O/phases.log 2>&1; tail -4 $O/phases.log
run spanning_b16384_w1 python bench.py --workload spanning --steps 2 --warmup 1 --block-loci 16384 --workers 1 --no-cpu-baseline
run spanning_b4096_w2 python bench.py --workload spanning --steps 2 --warmup 1 --block-loci 4096 --workers 2 --no-cpu-baseline
