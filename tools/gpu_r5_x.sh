#!/bin/bash
# round 5: random sweep of the big class' word-length rounds on hardware (8 000 repeat-rich piles of 129..236 reads under random assembler
# options -- word lengths from 8, steps 1..7, minCoverage 1..3, maxAssemblyCount 2..10 -- each against the CPU restatement)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05x
rm -rf $O && mkdir -p $O
cd $R
for s in 2000 3000 4000 5000 6000 7000 8000 9000; do
  timeout 800 python tools/sweeps/sweep_rounds.py $s 1000 > $O/sweep_$s.txt 2>&1 &
done
wait
cat $O/sweep_*.txt | tail -9
