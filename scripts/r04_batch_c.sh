#!/bin/bash
set +e
set +o pipefail
O=gpurun_out/r04_c
mkdir -p $O
echo "== parity with the study switches on"
NSK_TRI_WIDE=2 NSK_SPMV_WIDE=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "ilu or sgs or streamed or line_group or sync_free or spmv" 2>&1 | tail -3
echo "== A/B at the headline (K = 8): pairs of entries per lane in the scalar single-launch solves / the CSR-stream SpMV"
for cfg in "0 0" "2 0" "0 1" "2 1"; do
  set -- $cfg
  NSK_TRI_WIDE=$1 NSK_SPMV_WIDE=$2 timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/ab_wide_$1_$2.json 2> /dev/null
  python -c "
import json; d = json.load(open('$O/ab_wide_$1_$2.json')); print('NSK_TRI_WIDE=$1 NSK_SPMV_WIDE=$2 ms_per_step', round(d['ms_per_step'], 1), [(k['kernel'][:22], round(k['avg_ms'], 4)) for k in d['kernel_classes']])"
done
echo "== headline K20 with CPU samples (1200x400 K=1, 300x100 K=12)"
( time timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_K20.json 2> $O/bench_line_K20.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_c/bench_line_K20.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "setup_first", d["phases"]["setup_first_s"], "numeric", d["phases"]["setup_numeric_s"])
print([(k["kernel"][:30], round(k["avg_ms"], 4), round(k["frac_algorithmic"], 3)) for k in d["kernel_classes"]])
cb = d.get("cpu_baseline", {})
print("cpu:", cb.get("mesh"), cb.get("K"), cb.get("value"), cb.get("cores"), "gpu pair", (cb.get("gpu_same_mesh") or {}).get("value"), cb.get("sample", "")[-90:])
s2 = cb.get("second_sample", {})
print("cpu2:", s2.get("mesh"), s2.get("K"), s2.get("value"), "gpu pair", (s2.get("gpu_same_mesh") or {}).get("value"))
PY
echo "== N = 1 line with the N > 1 option set"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --cg-single-reduction 1 --inner-gs 2 --no-cpu-baseline > $O/bench_line_K20_multi_gpu_options.json 2> /dev/null
python -c "
import json; d = json.load(open('$O/bench_line_K20_multi_gpu_options.json')); print('ms_per_step', d['ms_per_step'], 'value', d['value'], d['config']['inner_cg'], '|', d['config']['inner_gram_schmidt'])"
echo "== headline with 8-byte BLAS-1"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --blas1-pairs 0 --no-cpu-baseline > $O/bench_line_K20_blas1_8byte.json 2> /dev/null
python -c "
import json; d = json.load(open('$O/bench_line_K20_blas1_8byte.json')); print('ms_per_step', d['ms_per_step'], 'value', d['value'], d['config']['inner_F_its_per_step'], d['config']['inner_S_its_per_step'])"
echo "== converged FGMRES + aSIMPLE with the negated Schur sign: 300x100, then 1200x400"
timeout -k 10 700 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --mesh 300,100 --converge 1e-10 --converge-mesh 300,100 --converge-preconditioner 2 --schur-sign -1 --converge-budget 300 > $O/converge_asimple_negated_300x100.json 2> $O/converge_asimple_negated_300x100.err
python -c "
import json; d = json.load(open('$O/converge_asimple_negated_300x100.json'))['converged_solve']; print({k: d[k] for k in ('workload','iters','final_res','status','seconds','true_residual','cancelled_after_budget_s','inner_F_its_per_step','inner_P_its_per_step') if k in d})"
