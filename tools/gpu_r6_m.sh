#!/bin/bash
# round 6, late: the two host overlaps of the whole-batch small-SV call (DMA started before plan(); the assembler's share of the compaction
# beside the aligners): the step with each switched off / both on, three rounds of 40 steps each (the box's CPU quota makes single runs noisy)
for rep in 1 2 3; do
for v in "A=1" "MANTA_AMD_NO_EARLY_STREAM=1" "MANTA_AMD_NO_EARLY_STAGE=1" "MANTA_AMD_NO_EARLY_STREAM=1 MANTA_AMD_NO_EARLY_STAGE=1"; do
  env $v python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['pcie']['host_ms_per_step'], d['kernels_ms_per_step']['assembler_stage'], d['config']['parity'][-14:])"
done
done
