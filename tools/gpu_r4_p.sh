#!/bin/bash
# round 4: read gathering with the record-parallel test kernel and the pile-read-parallel pack: parity, throughput (page-locked
# host memory), kernel stats and one counter pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04p
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_read_class.py tests/test_demo_real_data.py tests/test_host_adapter.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python tools/bench_read_class.py 150 > $O/read_class.log 2>&1
tail -5 $O/read_class.log | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rc -o rc -- python $R/tools/bench_read_class.py 150 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_rc -o p -- python $R/tools/bench_read_class.py 150 > /dev/null 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.rocpd" -delete
find $O/stats_rc -name "*kernel_stats.csv" | head -1 | xargs -r head -9 | cut -c1-150
ls $O/pmc_rc | head
