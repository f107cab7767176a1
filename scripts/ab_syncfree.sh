#!/bin/bash
# same-box A/B of NSK_OPT_TRI_SYNC_FREE (1: scalar factors single-launch; 2: also the blocked velocity factor)
set +e; set +o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for sf in 1 2; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 --sync-free $sf --cpu-steps 1 > gpurun_out/ab_sf${sf}_$rep.log 2>&1
    echo "sync-free $sf rep $rep: $(grep '^{' gpurun_out/ab_sf${sf}_$rep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), 'ms/step', [round(k['avg_ms'],4) for k in d['kernel_classes']])")"
  done
done
exit 0
