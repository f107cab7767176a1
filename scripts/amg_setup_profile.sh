#!/bin/bash
# Kernel statistics and phase times of the device AMG set-up at a mesh (default 1200,400).
#   gpurun -- bash scripts/amg_setup_profile.sh [mesh] [tag]
mesh=${1:-1200,400}; tag=${2:-amg}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
NSK_AMG_TIMING=1 timeout -k 10 300 python3 $GRAFT_REPO_ROOT/scripts/amg_info.py $mesh > $out/${tag}_phases.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/${tag}_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/amg_info.py $mesh > $out/${tag}_prof.log 2>&1 || exit 1
f=$(ls $out/${tag}_prof/*/*kernel_stats.csv | head -1)
grep amgk $f | awk -F'","' '{printf "%-60s calls %4s total %10.3f ms\n", substr($1,1,60), $2, $3/1e6}' | sed 's/"void nsk::amgk::(anonymous namespace):://; s/"nsk::amgk::(anonymous namespace):://' > $out/${tag}_kernels.txt
grep -v "level [1-9]" $out/${tag}_phases.log | tail -14; cat $out/${tag}_kernels.txt
