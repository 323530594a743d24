#!/bin/bash
# round 4: schedule kernel knobs; SQ counters of graph_kernel / contig_kernel / schedule kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04m
rm -rf $O && mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2"
for K in "12 2" "12 1" "12 4" "6 2" "24 1" "3 4"; do
  set -- $K
  MANTA_AMD_SCHED_WAVES_PER_CU=$1 MANTA_AMD_SCHED_CHUNK=$2 timeout 80 $B > $O/bench_s$1_c$2.json 2> /dev/null
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04m")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j["value"], j["ms_per_step"], j["kernels_ms_per_step"]["schedule_kernel"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
cd /tmp
P="python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0"
export MANTA_AMD_NO_STREAM_UPLOAD=1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc1 -o p -- $P > /dev/null 2>&1
find $O -name "*.rocpd" -delete
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04m")
for f in glob.glob(O + "/pmc1/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("manta_dev::", "").replace("void ", "")
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
    for (k, c), v in sorted(acc.items()):
        if any(x in k for x in ("graph", "contig", "schedule", "pair_kernel<5>")):
            print(k, c, "%.4g" % v)
PY
