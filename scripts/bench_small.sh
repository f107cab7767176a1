#!/bin/bash
# the launch-latency-bound configuration (BASELINE configs[5]: 600x200, unsteady, FGMRES + aSIMPLE) with kernel stats
set -e
mkdir -p gpurun_out/small
python3 bench.py --mesh 600,200 --variant 1 --preconditioner 2 --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/small/bench_600x200_unsteady.json 2> gpurun_out/small/bench_600x200_unsteady.err
tail -1 gpurun_out/small/bench_600x200_unsteady.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/small/prof -o small -- python3 $GRAFT_REPO_ROOT/bench.py --mesh 600,200 --variant 1 --preconditioner 2 --steps 200 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/small/bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find gpurun_out/small/prof -name "*kernel_trace.csv" -delete
find gpurun_out/small/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -25 {}
