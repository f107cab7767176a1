#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh r02
# Writes under gpurun_out/prof_<tag>/: the bench lines, the rocprofv3 kernel statistics of the driver's bench command
# and the FETCH_SIZE / WRITE_SIZE passes (counters in their own runs, no tracing) + their per-class summary.
set -u
tag=${1:-r02}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/prof_$tag
mkdir -p $repo/$out
cd $repo
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $out/bench_line_K20.json 2> $out/bench_line_K20.err || echo "bench K20 failed"
echo "bench K20 done"
timeout -k 10 500 python3 bench.py --no-cpu-baseline > $out/bench_line_default_K58.json 2> $out/bench_line_default_K58.err || echo "bench K58 failed"
echo "bench K58 done"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $repo/$out/kt --output-format csv -- python3 $repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $repo/$out/bench_line_under_rocprof_K20.json 2> $repo/$out/rocprof_kt.err) || echo "kernel trace failed"
rm -f $out/kt/*/*_kernel_trace.csv   # (70 MB; the statistics file is what is kept)
echo "kernel trace done"
mkdir -p $out/pmc
bash scripts/pmc_passes.sh $out/pmc "FETCH_SIZE" "WRITE_SIZE"
f=$(ls $out/pmc/pass1/*/*counter_collection.csv 2>/dev/null | head -1)
w=$(ls $out/pmc/pass2/*/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && [ -n "$w" ] && python3 scripts/pmc_traffic.py $f $w 1200 400 $out/pmc_traffic_1200x400.json > $out/pmc_traffic.log 2>&1
ls -la $out $out/kt/* 2>/dev/null | head -40
