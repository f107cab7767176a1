#!/bin/bash
# the ring solve with its records prefetched into the memory-side cache by the whole chip (NSK_RING_PREFETCH, default on):
# time loop back to back / with an SpMV of F between two applications, and where it is used — config 5's bench line
set -o pipefail
out=gpurun_out/r04_ring_prefetch; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config5.py -q -k "lds_ring or ring_solve" 2>&1 | tail -3 || exit 1
for v in 0 1; do echo "== time loop, NSK_RING_PREFETCH=$v"; NSK_RING_PREFETCH=$v timeout -k 10 200 python scripts/time_ring.py 600,200 20 2>&1 | grep "ring  :\|walker" || exit 1; done
for v in 0 1 0 1; do
  NSK_RING_PREFETCH=$v timeout -k 10 300 python bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --no-cpu-baseline > $out/bench_prefetch_$v.json 2> $out/bench_prefetch_$v.err || exit 1
  python - $v <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r04_ring_prefetch/bench_prefetch_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("NSK_RING_PREFETCH=%s ms_per_step %.3f" % (sys.argv[1], d["ms_per_step"]), "pressure-solve class avg_ms", round(d["roofline"]["avg_ms"],4), d["roofline"]["kernel"][:20])
PY
done
