#!/bin/bash
# round 3: HBM traffic of the graph phases alone (pack + table + links; -DMANTA_ASM_STOP_AFTER_GRAPH variant) against the whole kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03z
rm -rf $O && mkdir -p $O
cat > /tmp/one.py <<PY
import os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from manta_amd._capi import Lib, SmallSvBatch
from oracle_lib import asm_opts
from synth import config2_batch
lib = Lib(path=sys.argv[1] if len(sys.argv) > 1 else None)
b = SmallSvBatch(lib, asm_opts(minWordLength=31), [2, -8, -24, -1, -1, 0], -100)
b.upload_packed(*config2_batch(10000, seed=12345))
b.run()
print(b.stats()["assemble_ms"])
PY
for v in full graph; do
  L=""; [ $v = graph ] && L=$R/manta_amd/variants/lib_graph.so
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${v}_$c -o p -- python /tmp/one.py $L > $O/${v}_$c.log 2>&1
  done
done
python - <<PY
import csv, glob
for v in ("full", "graph"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$O/%s_%s/**/*counter_collection.csv" % (v, c), recursive=True):
            tot = 0
            for row in csv.DictReader(open(f)):
                if "assemble_kernel" in row["Kernel_Name"]: tot += float(row["Counter_Value"])
            print(v, c, "%.3f GB" % (tot * 1024 / 1e9))
PY
find $O -name "*.rocpd" -delete; find $O -name "*_kernel_trace.csv" -delete
