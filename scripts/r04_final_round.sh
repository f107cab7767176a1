#!/bin/bash
# after the ring prefetch: the round's profile set once more (bench lines, kernel statistics, PMC passes for the changed
# kernel sources) and the CLI's first level of config 5
set -u
O=gpurun_out/r04_final; mkdir -p $O
bash scripts/profile_round.sh r04e > $O/profile_round.log 2>&1
tail -5 $O/profile_round.log
( time timeout -k 10 400 navier_stokes_solver_amd/bin/NSSolver -T 0.01,0.01 -m 600,200 -r 1 -p 0 -t 1e-6 ) > $O/cli_first_level.log 2>&1
grep -i "iterations\|real" $O/cli_first_level.log | tail -8
