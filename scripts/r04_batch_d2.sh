#!/bin/bash
set +e
set +o pipefail
O=gpurun_out/r04_d
mkdir -p $O
echo "== converged FGMRES + aSIMPLE with the negated Schur sign at 300x100 (config 2's mesh)"
timeout -k 10 700 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --mesh 300,100 --converge 1e-10 --converge-mesh 300,100 --converge-preconditioner 2 --schur-sign -1 --converge-budget 450 > $O/converge_asimple_negated_300x100.json 2> $O/converge_asimple_negated_300x100.err
python -c "
import json; d = json.load(open('$O/converge_asimple_negated_300x100.json'))['converged_solve']; print({k: d[k] for k in ('workload','iters','final_res','status','seconds','true_residual','cancelled_after_budget_s','inner_F_its_per_step','inner_P_its_per_step') if k in d})"
echo "== the same with the reference's sign, 120 s budget (the plateau)"
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --mesh 300,100 --converge 1e-10 --converge-mesh 300,100 --converge-preconditioner 2 --converge-budget 120 > $O/converge_asimple_reference_sign_300x100.json 2> $O/converge_asimple_reference_sign_300x100.err
python -c "
import json; d = json.load(open('$O/converge_asimple_reference_sign_300x100.json'))['converged_solve']; print({k: d[k] for k in ('iters','final_res','status','seconds','cancelled_after_budget_s') if k in d})"
echo "== N = 1 lines: the N > 1 option set; 8-byte BLAS-1"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --cg-single-reduction 1 --inner-gs 2 --no-cpu-baseline > $O/bench_line_K20_multi_gpu_options.json 2> /dev/null
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --blas1-pairs 0 --no-cpu-baseline > $O/bench_line_K20_blas1_8byte.json 2> /dev/null
python -c "
import json
for f in ('bench_line_K20_multi_gpu_options', 'bench_line_K20_blas1_8byte'):
    d = json.load(open('$O/' + f + '.json')); print(f, 'ms_per_step', d['ms_per_step'], 'value', d['value'], d['config']['inner_cg'], '|', d['config']['inner_gram_schmidt'], '|', d['config']['blas1_reductions'], d['config']['inner_F_its_per_step'], d['config']['inner_S_its_per_step'])"
echo "== config 5 (600x200 unsteady -p 0 and -p 2), kernel stats of -p 0"
timeout -k 10 300 python bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_config5_p0_K40.json 2> /dev/null
timeout -k 10 300 python bench.py --mesh 600,200 --variant 1 --preconditioner 2 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_config5_p2_K200.json 2> /dev/null
python -c "
import json
for f in ('bench_config5_p0_K40', 'bench_config5_p2_K200'):
    d = json.load(open('$O/' + f + '.json')); print(f, 'ms_per_step', round(d['ms_per_step'], 3), [(k['kernel'][:22], round(k['avg_ms'], 4)) for k in d['kernel_classes']])"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt5 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --mesh 600,200 --variant 1 --preconditioner 0 --steps 40 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_config5_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/kt5.err)
rm -f $O/kt5/*/*_kernel_trace.csv
f=$(ls $O/kt5/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-200
