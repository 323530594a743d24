#!/bin/bash
# round 3: HBM-side traffic of assemble_kernel after the dense support array (two counter passes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03x
rm -rf $O && mkdir -p $O
export MANTA_AMD_NO_STREAM_UPLOAD=1
P="--steps 1 --warmup 0 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/bench.py $P > /dev/null 2>&1
python - <<PY
import csv, glob
for d in ("pmc_fetch", "pmc_write"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        tot = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            tot[k] = tot.get(k, 0) + float(row["Counter_Value"])
        for k, v in tot.items():
            if "assemble" in k: print(d, k, "%.3f GB" % (v * 1024 / 1e9))
PY
find $O -name "*.rocpd" -delete; find $O -name "*_kernel_trace.csv" -size +8M -delete
