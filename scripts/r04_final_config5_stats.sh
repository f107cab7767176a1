#!/bin/bash
# config-5 kernel statistics with the round's final code + the config-5-size ring bit test
set -o pipefail
repo=$(pwd); out=gpurun_out/r04_final5; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_config5.py -q -k "ring_solve" 2>&1 | tail -3 || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $repo/$out/kt --output-format csv -- python3 $repo/bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --no-cpu-baseline > $repo/$out/bench_line_config5_under_rocprof_K40.json 2> $repo/$out/rocprof.err) || echo "kernel trace failed"
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_config5.csv
head -8 $out/kernel_stats_config5.csv | cut -c1-160
timeout -k 10 300 python bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --no-cpu-baseline > $out/bench_line_config5_K40.json 2> $out/bench.err || exit 1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_final5/bench_line_config5_K40.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"])
PY
