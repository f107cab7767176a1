#!/bin/bash
set +e
set +o pipefail
echo "== full-size tests (incl. the oracle at the headline size) and the parity file"
python -m pytest tests/test_gpu_full_size.py -q -m gpu --durations=5 2>&1 | tail -9
python -m pytest tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -3
echo "== profile round r04"
bash scripts/profile_round.sh r04 2>&1 | tail -25
python - <<'PY'
import json
for f in ("bench_line_K20", "bench_line_default_K58", "bench_line_under_rocprof_K20"):
    try:
        d = json.load(open(f"gpurun_out/prof_r04/{f}.json"))
        print(f, "ms_per_step", round(d["ms_per_step"], 1), "value", round(d["value"]), "setup_first", round(d["phases"]["setup_first_s"], 2),
              [(k["kernel"][:22], round(k["avg_ms"], 4), round(k["frac_algorithmic"], 3)) for k in d["kernel_classes"]], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_stale"])
    except Exception as e:
        print(f, "failed", e)
PY
