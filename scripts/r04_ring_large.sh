#!/bin/bash
# the ring with 16 384 slots (128 KB of LDS) and epochs down to 12 passes: meshes whose levels did not fit 8 192 slots;
# the records prefetched half by half
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config5.py -q -k "lds_ring or ring_solve" 2>&1 | tail -3 || exit 1
for m in 600,200 1200,400; do for pf in 1 0; do echo "== $m NSK_RING_PREFETCH=$pf"; NSK_RING_PREFETCH=$pf timeout -k 10 400 python scripts/time_ring.py $m 5 2>&1 | grep "set-up\|ring  :\|walker\|Error\|ring solve" ; done; done
timeout -k 10 300 python bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5 ms_per_step %.3f' % d['ms_per_step'], 'pressure solve %.4f ms' % d['roofline']['avg_ms'])"
