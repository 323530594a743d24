#!/bin/bash
# round 4: occupancy the runtime grants contig_kernel per LDS size, fine-grained graph_kernel phases, SQ counters of both kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04c
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
MANTA_AMD_DEBUG=1 MANTA_AMD_LG_CLASSES="16384,20480,27136,54272" timeout 100 $B > $O/bench_dbg.json 2> $O/bench_dbg.err
grep "contig_kernel class" $O/bench_dbg.err | sort | uniq -c
timeout 100 python tools/profile_phases.py 10000 > $O/phases.log 2>&1
MANTA_AMD_LIB=$R/manta_amd/variants/libmanta_amd_profg.so timeout 100 python tools/profile_phases.py 10000 > $O/phases_graph.log 2>&1
tail -2 $O/phases.log | cut -c1-600; tail -2 $O/phases_graph.log | cut -c1-600
cd /tmp
P="python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0"
export MANTA_AMD_NO_STREAM_UPLOAD=1 MANTA_AMD_LG_CLASSES="32768,54272"
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/pmc1 -o p -- $P > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc2 -o p -- $P > /dev/null 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $P > /dev/null 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $P > /dev/null 2>&1
find $O -name "*.rocpd" -delete
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04c")
for d in ("pmc1", "pmc2", "pmc_fetch", "pmc_write"):
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("manta_dev::", "").replace("void ", "")
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
        for (k, c), v in sorted(acc.items()):
            if "graph" in k or "contig" in k or "assemble" in k:
                print(d, k, c, "%.4g" % v)
PY
