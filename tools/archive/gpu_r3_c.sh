#!/bin/bash
# SQ counters of assemble_fast_kernel (instruction mix, waits)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03c
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
P="--steps 1 --warmup 0 --no-cpu-baseline --no-extras"
export MANTA_AMD_NO_STREAM_UPLOAD=1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_sq -o p -- $B $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- $B $P > /dev/null 2>&1
find $O -name "*.rocpd" -delete
python - <<PY
import csv,glob,collections
for d in ("pmc_sq","pmc_sq2"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "assemble" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])]+=float(r["Counter_Value"])
        for k,v in sorted(acc.items()): print(k, "%.4g"%v)
PY
