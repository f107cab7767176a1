#!/bin/bash
set +e
set +o pipefail
O=gpurun_out/r04_e
mkdir -p $O
echo "== ring with the merged per-row load"
python scripts/time_ring.py 600,200 20 2>&1 | grep "ring  :\|walker"
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "lds_ring or negated_schur or line_groups_are_dropped" 2>&1 | tail -3
python -m pytest tests/test_gpu_config5.py tests/test_partition.py -q -m gpu 2>&1 | tail -3
echo "== config 5 bench and the CLI's first level (M_p factor kept between set-ups)"
timeout -k 10 300 python bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_config5_p0_K40.json 2> /dev/null
python -c "
import json; d = json.load(open('$O/bench_config5_p0_K40.json')); print('ms_per_step', round(d['ms_per_step'], 3), [(k['kernel'][:22], round(k['avg_ms'], 4)) for k in d['kernel_classes']])"
( time timeout -k 10 400 navier_stokes_solver_amd/bin/NSSolver -T 0.01,0.01 -m 600,200 -r 1 -p 0 -t 1e-6 ) > $O/cli_first_level.log 2>&1
grep "Newton iteration\|\[nsk\]\|real" $O/cli_first_level.log
echo "== achieved parity errors of the converged solves"
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "solve_matches_oracle_and_direct" 2>&1 | grep "PARITY\|passed\|failed" | cut -c1-260
