#!/bin/bash
# node layer on hardware: two contexts on the one GPU; the multi-rank bench paths on one GPU (RCCL with one rank; two gloo ranks sharing the device)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03k
rm -rf $O && mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_node.py tests/test_multi_rank.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
MANTA_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_forced.json 2> $O/bench_forced.err
tail -1 $O/bench_forced.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('forced dist (nccl, 1 rank):', d['value'], d['ms_per_step'], c['queue'], c['backend'], c['dist_world'], c['loci_per_rank'], c['block_loci'], c['timed_region'][-60:], c['parity'][-20:])"
MANTA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_2rank.json 2> $O/bench_2rank.err
tail -1 $O/bench_2rank.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('2 ranks on one GPU (gloo):', d['value'], d['ms_per_step'], c['queue'], c['backend'], c['dist_world'], c['loci_per_rank'], c['block_loci'], c['parity'][-20:])"
MANTA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --mix --loci 4000 --no-cpu-baseline --no-extras > $O/bench_2rank_mix.json 2> $O/bench_2rank_mix.err
tail -1 $O/bench_2rank_mix.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('2 ranks mix:', d['value'], d['ms_per_step'], c['loci_per_rank'], c['mix'])"
tail -3 $O/bench_2rank.err
