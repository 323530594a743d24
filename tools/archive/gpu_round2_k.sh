#!/bin/bash
# SmallAssembler on hardware + the whole GPU tier
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02k
rm -rf $O && mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_small_assembler.py tests/test_host_adapter.py tests/test_dropin.py -m gpu -x -q > $O/small.log 2>&1; tail -5 $O/small.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
