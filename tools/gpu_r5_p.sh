#!/bin/bash
# round 5: kernel trace of the spanning workload with the big class' word-length rounds (which kernel of which round takes the time)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05p
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o span -- python $R/bench.py --workload spanning --loci 16384 --steps 1 --warmup 1 --no-cpu-baseline > $O/out.json 2> $O/err.txt
find $O/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python - {} <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# the last step only: find the last graph_big_kernel launch group
names = [r["Kernel_Name"] for r in rows]
out = []
for r in rows:
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("manta_dev::", "").replace("void ", "")
    out.append(((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, short, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Stream_Id", "")))
# print the second half (the timed step)
half = len(out) // 2
for s, d, n, g, st in out[half:]:
    if d >= 0.05 or "big" in n: print("%10.3f %9.3f  %-40s grid %s stream %s" % (s, d, n[:40], g, st))
PY
