#!/bin/bash
# FGMRES + blockTriangular (velocity AMG) at 1200x400: bench lines at two viscosities and the whole Newton run of the C++ driver.
#   gpurun -- bash scripts/amg_solve_check.sh [tag]
tag=${1:-amg}
R=$GRAFT_REPO_ROOT; L=$R/navier_stokes_solver_amd
export LD_LIBRARY_PATH=$L:/opt/rocm/lib:$LD_LIBRARY_PATH
for re in 11 100; do
  timeout -k 10 300 python bench.py --preconditioner 1 --reynolds $re --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_p1_re$re.json 2> gpurun_out/${tag}_bench_p1_re$re.err || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/${tag}_bench_p1_re$re.json').read().strip().splitlines()[-1])
print('re $re: ms/step %.1f  inner F its/step %.2f  res %.3e' % (d['ms_per_step'], d['config']['inner_F_its_per_step'], d['config']['residual_after_K']))"
done
NSK_AMG_TIMING=1 timeout -k 10 300 $L/bin/StationaryNSSolver -m 1200,400 -r 30 -s 1 -p 1 -t 1e-6 > gpurun_out/${tag}_cli_newton_1200x400.log 2> gpurun_out/${tag}_cli_newton_1200x400.err || exit 1
tail -1 gpurun_out/${tag}_cli_newton_1200x400.log; grep "set-up: total" gpurun_out/${tag}_cli_newton_1200x400.err
