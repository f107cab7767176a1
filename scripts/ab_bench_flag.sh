#!/bin/bash
# same-box A/B inside the solve: "$1" = bench flag to vary (default --sync-free), values $2 $3 (default 1 2)
set +e; set +o pipefail
mkdir -p gpurun_out
flag=${1:---sync-free}; a=${2:-1}; b=${3:-2}
for rep in 1 2; do
  for sf in $a $b; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 $flag $sf --cpu-steps 1 > gpurun_out/ab_sf${sf}_$rep.log 2>&1
    echo "$flag $sf rep $rep: $(grep '^{' gpurun_out/ab_sf${sf}_$rep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), 'ms/step', [round(k['avg_ms'],4) for k in d['kernel_classes']], 'res', d['config']['residual_after_K'])")"
  done
done
exit 0
