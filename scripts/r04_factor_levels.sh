#!/bin/bash
# numeric ILU(0) of natural-order factors: one launch per level from 48 rows on (was: one workgroup for every level below 1 024 rows)
set -o pipefail
out=gpurun_out/r04_factor_levels; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config5.py tests/test_golden.py -q -x 2>&1 | tail -3 || exit 1
NSK_VERBOSE=1 timeout -k 10 200 python scripts/time_ring.py 600,200 5 2>&1 | grep "set-up\|factorise\|ring  :\|walker" | head -12
timeout -k 10 300 python bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --no-cpu-baseline > $out/bench_config5.json 2> $out/bench_config5.err || exit 1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_factor_levels/bench_config5.json").read().strip().splitlines()[-1])
print("ms_per_step %.3f" % d["ms_per_step"], d.get("phases"))
PY
