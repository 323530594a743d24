#!/bin/bash
# kernel trace of the side-by-side launch: do the two assembler kernels overlap?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03f
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
find $O -name "*.rocpd" -delete
python - <<PY
import csv,glob
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f))]
    t0=min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows[-14:]:
        print(r["Kernel_Name"].split("(")[0][:40], (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-t0)/1e6, "ms  grid", r.get("Grid_Size"), "wg", r.get("Workgroup_Size"), "lds", r.get("LDS_Block_Size"))
PY
