#!/bin/bash
# round 5: the assembler's big LDS class on hardware -- parity (big-class cases, config-5 digests, config-2 digests), then the spanning
# workload at 16 384 and 65 536 loci
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05b
rm -rf $O && mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_assemble_kernels.py tests/test_digests.py -m gpu -x -q -k "big_class or config5 or fast_kernel_matches or config2" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
MANTA_AMD_DEBUG=1 timeout 300 python bench.py --workload spanning --loci 16384 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_spanning_16k.json 2> $O/bench_spanning_16k.err
grep "LDS assembler pipeline" $O/bench_spanning_16k.err | tail -2
python - <<PY
import json
d=json.loads(open("$O/bench_spanning_16k.json").read().strip().splitlines()[-1])
print("16k:", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["config"]["parity"])
PY
timeout 600 python bench.py --workload spanning --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_spanning_64k.json 2> $O/bench_spanning_64k.err
python - <<PY
import json
d=json.loads(open("$O/bench_spanning_64k.json").read().strip().splitlines()[-1])
print("64k:", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["config"]["parity"])
PY
