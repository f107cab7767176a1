#!/bin/bash
# round 4: the rebuilt natural-order ring solve on config 5 (600x200 unsteady, -p 0): timings, the CLI's first level, kernel stats
set +e
set +o pipefail
O=gpurun_out/r04_ring
mkdir -p $O
echo "== by-class dealing of rows to wavefronts: on / off"
NSK_RING_BY_CLASS=1 python scripts/time_ring.py 600,200 20 2>&1 | grep "ring  :"
NSK_RING_BY_CLASS=0 python scripts/time_ring.py 600,200 20 2>&1 | grep "ring  :"
echo "== tests"
python -m pytest tests/test_partition.py -q -m gpu -k "failing_rank" 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "vector_ops or dot_norm or one_launch or lds_ring or first_restart" 2>&1 | tail -4
echo "== bench config 5, 8-byte BLAS-1 (round-3 arithmetic) and 16-byte BLAS-1"
NSK_BLAS1_PAIRS=0 python bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_config5_K40_blas1_scalar.json 2> $O/bench_config5_K40_blas1_scalar.err
python bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_config5_K40.json 2> $O/bench_config5_K40.err
python - <<'PY'
import json
for f in ("bench_config5_K40_blas1_scalar", "bench_config5_K40"):
    try:
        d = json.load(open(f"gpurun_out/r04_ring/{f}.json"))
        print(f, "ms_per_step", round(d["ms_per_step"], 3), "value", d["value"], [(k["kernel"][:28], round(k["avg_ms"], 4)) for k in d["kernel_classes"]])
    except Exception as e:
        print(f, "failed", e)
PY
echo "== CLI first level (-T 0.01,0.01 -r 1): round-3 BLAS-1 arithmetic, then the 16-byte loads"
( time NSK_BLAS1_PAIRS=0 timeout -k 10 400 navier_stokes_solver_amd/bin/NSSolver -T 0.01,0.01 -m 600,200 -r 1 -p 0 -t 1e-6 ) > $O/cli_first_level_blas1_scalar.log 2>&1
grep "Newton iteration\|\[nsk\]\|real" $O/cli_first_level_blas1_scalar.log
( time timeout -k 10 400 navier_stokes_solver_amd/bin/NSSolver -T 0.01,0.01 -m 600,200 -r 1 -p 0 -t 1e-6 ) > $O/cli_first_level.log 2>&1
grep "Newton iteration\|\[nsk\]\|real" $O/cli_first_level.log
echo "== rocprofv3 kernel stats of the config-5 bench"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o cfg5 -- python3 $GRAFT_REPO_ROOT/bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof_bench.err
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1)
echo "stats file: $f"; head -12 "$f"
