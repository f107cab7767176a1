#!/bin/bash
# wall time of the small Newton drivers with each preconditioner (choosing test configurations)
set +e
set +o pipefail
for p in 1 0 2; do
  s=$(date +%s.%N)
  timeout -k 10 200 python -m navier_stokes_solver_amd.cli StationaryNSSolver -m 16,10 -r 30 -s 1 -p $p -t 1e-12 > /tmp/o.txt 2>&1
  e=$(date +%s.%N)
  echo "stationary prec $p: rc=$? $(echo "$e - $s" | bc) s; $(grep nsk /tmp/o.txt | tail -1)"
done
for p in 1 0 2; do
  s=$(date +%s.%N)
  timeout -k 10 200 python -m navier_stokes_solver_amd.cli NSSolver -T 0.02,0.01 -m 16,10 -r 11 -s 1 -p $p -t 1e-12 > /tmp/o.txt 2>&1
  e=$(date +%s.%N)
  echo "unsteady prec $p: rc=$? $(echo "$e - $s" | bc) s; $(grep nsk /tmp/o.txt | tail -1)"
done
exit 0
