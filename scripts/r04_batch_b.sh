#!/bin/bash
# round 4, batch B: headline line with the CPU sample on the bench mesh, set-up phases, the N > 1 option set on one GPU,
# the full-size oracle parity study, a converged FGMRES + aSIMPLE solve with the negated Schur sign
set +e
set +o pipefail
O=gpurun_out/r04_b
mkdir -p $O
echo "== wide loads in the single-launch scalar triangular solves: parity tests, then A/B at the headline (K = 8)"
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "ilu or sgs or streamed or line_group or sync_free or lds_ring" 2>&1 | tail -3
python -m pytest tests/test_gpu_full_size.py -q -m gpu 2>&1 | tail -3
for w in 0 1; do
  NSK_TRI_WIDE=$w timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/ab_tri_wide_$w.json 2> /dev/null
  python -c "
import json; d = json.load(open('$O/ab_tri_wide_$w.json')); print('NSK_TRI_WIDE=$w ms_per_step', round(d['ms_per_step'], 1), [(k['kernel'][:22], round(k['avg_ms'], 4)) for k in d['kernel_classes']], d['config']['inner_F_its_per_step'], d['config']['inner_S_its_per_step'])"
done
echo "== headline K20 with CPU samples (1200x400 K=3, 300x100 K=12)"
( time NSK_VERBOSE=1 timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_K20.json 2> $O/bench_line_K20.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_b/bench_line_K20.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "setup_first", d["phases"]["setup_first_s"], "numeric", d["phases"]["setup_numeric_s"])
print([(k["kernel"][:30], round(k["avg_ms"], 4), round(k["frac_algorithmic"], 3)) for k in d["kernel_classes"]])
cb = d.get("cpu_baseline", {})
print("cpu:", cb.get("mesh"), cb.get("K"), cb.get("value"), cb.get("cores"), "gpu pair", (cb.get("gpu_same_mesh") or {}).get("value"), cb.get("sample", "")[-80:])
s2 = cb.get("second_sample", {})
print("cpu2:", s2.get("mesh"), s2.get("K"), s2.get("value"), "gpu pair", (s2.get("gpu_same_mesh") or {}).get("value"))
PY
grep "\[nsk\]" $O/bench_line_K20.err | head -60
echo "== N = 1 line with the N > 1 option set"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --cg-single-reduction 1 --inner-gs 2 --no-cpu-baseline > $O/bench_line_K20_multi_gpu_options.json 2> $O/bench_line_K20_multi_gpu_options.err
python -c "
import json; d = json.load(open('$O/bench_line_K20_multi_gpu_options.json')); print('ms_per_step', d['ms_per_step'], 'value', d['value'], d['config']['inner_cg'], '|', d['config']['inner_gram_schmidt'])"
echo "== headline with 8-byte BLAS-1 (A/B of the 16-byte loads)"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --blas1-pairs 0 --no-cpu-baseline > $O/bench_line_K20_blas1_8byte.json 2> /dev/null
python -c "
import json; d = json.load(open('$O/bench_line_K20_blas1_8byte.json')); print('ms_per_step', d['ms_per_step'], 'value', d['value'], d['config']['inner_F_its_per_step'], d['config']['inner_S_its_per_step'])"
echo "== oracle parity at 1200x400"
( time timeout -k 10 1500 python tests/studies/oracle_parity_full_size.py 1200,400 ) > $O/oracle_parity_1200x400.log 2>&1
cat $O/oracle_parity_1200x400.log | grep -v amdgpu.ids
