#!/bin/bash
# round 6, late: the early staging in the spanning batch call -- parity, then the config-4/5 block with and without it; the metric's step again
python -m pytest tests/test_batch_calls.py tests/test_digests.py tests/test_node.py -x -q -m gpu 2>&1 | tail -3
for v in "A=1" "MANTA_AMD_NO_EARLY_STAGE=1" "A=1"; do
  env $v python bench.py --workload spanning --loci 65536 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v spanning', d['value'], d['ms_per_step'], d['pcie']['host_ms_per_step'], d['config']['parity'][-14:])"
done
for i in 1 2 3; do python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pcie']['host_ms_per_step'], d['config']['parity'][-14:])"; done
