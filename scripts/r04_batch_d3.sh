#!/bin/bash
set +e
set +o pipefail
O=gpurun_out/r04_d
mkdir -p $O
echo "== A/B: skewed vector allocations (channel phases of the nine Gram-Schmidt streams), K = 8"
for w in 0 1; do
  NSK_VEC_SKEW=$w timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/ab_vec_skew_$w.json 2> /dev/null
  python -c "
import json; d = json.load(open('$O/ab_vec_skew_$w.json')); print('NSK_VEC_SKEW=$w ms_per_step', round(d['ms_per_step'], 1), 'blas1 GB', round(d['phases']['blas1_GB'],1))"
done
for w in 0 1; do
  echo "kernel stats with NSK_VEC_SKEW=$w"
  (cd /tmp && export TMPDIR=/tmp && NSK_VEC_SKEW=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kts$w --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> /dev/null)
  rm -f $O/kts$w/*/*_kernel_trace.csv
  grep "multi_dot\|multi_axpy\|vec_dot\|vec_cg_update\|vec_axpy(" $O/kts$w/*/*kernel_stats.csv | awk -F, '{print $1, $2, $4}' | cut -c1-60,150-260 | head -8
done
