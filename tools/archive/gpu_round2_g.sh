#!/bin/bash
# calibrate the memory-side counters on the scattered-gather microbenchmark (bytes fetched per scattered lane-load)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02g
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G=$R/tools/microbench/gather_ceiling
for cfg in "0 4 64 16" "0 4 4 16" "1 4 64 16"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$tag -o p -- $G 2097152 $cfg > $O/run_$tag.txt 2>&1
  timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/tcc_$tag -o p -- $G 2097152 $cfg > $O/run2_$tag.txt 2>&1
  timeout 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TCP_GATE_EN2_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $O/tcp_$tag -o p -- $G 2097152 $cfg > $O/run3_$tag.txt 2>&1
done
timeout 60 rocprofv3 -L > $O/counters_avail.txt 2>&1
find $O -name "*.rocpd" -delete
python - $O <<'PY'
import csv,glob,sys,collections
for f in sorted(glob.glob(sys.argv[1]+'/*/*counter_collection.csv')+glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv')):
    tot=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        tot[(r['Kernel_Name'][:30],r['Counter_Name'])]+=float(r['Counter_Value'])
    print(f.split('/')[-2] if 'r02g' in f.split('/')[-3] else f.split('/')[-3])
    for k,v in sorted(tot.items()): print('   ',k,v)
PY
