#!/bin/bash
# round 4: SQ counters of the packed aligner at 16 and 4 waves per CU
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04l
rm -rf $O && mkdir -p $O
cd /tmp
P="python $R/tools/profile_phases.py 10000"
export MANTA_AMD_LIB=$R/manta_amd/libmanta_amd.so
for W in 16 4; do
  MANTA_AMD_ALIGN_WAVES_PER_CU=$W timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_w$W -o p -- $P > /dev/null 2>&1
  MANTA_AMD_ALIGN_WAVES_PER_CU=$W timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc2_w$W -o p -- $P > /dev/null 2>&1
done
find $O -name "*.rocpd" -delete
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04l")
for d in sorted(glob.glob(O + "/pmc*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("manta_dev::", "").replace("void ", "")
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
        for (k, c), v in sorted(acc.items()):
            if "align_pair_kernel<5>" in k or "align_pair_kernel<4>" in k:
                print(os.path.basename(d), k, c, "%.4g" % (v / n[(k, c)]))
PY
