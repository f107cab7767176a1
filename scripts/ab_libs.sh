#!/bin/bash
# A/B of compile-time variants of the library (NSK_HIP_LIBRARY=<path>): tri applies on several meshes
# usage: scripts/ab_libs.sh out.log lib1.so lib2.so ...
out=$1; shift
: > $out
for mesh in 1200,400 600,200 100,70; do
  for lib in "$@"; do
    echo "== $mesh $(basename $lib)" >> $out
    NSK_HIP_LIBRARY=$lib python3 scripts/time_ops.py --mesh $mesh --reps 60 2>/dev/null | grep -E "^tri|^spmv S" >> $out || exit 1
  done
done
