#!/bin/bash
# Kernel statistics and phase times of the device AMG set-up at a mesh (default 1200,400).
#   gpurun -- bash scripts/amg_setup_profile.sh [mesh] [tag]
mesh=${1:-1200,400}; tag=${2:-amg}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
NSK_AMG_TIMING=1 timeout -k 10 300 python3 $GRAFT_REPO_ROOT/scripts/amg_info.py $mesh > $out/${tag}_phases.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/${tag}_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/amg_info.py $mesh > $out/${tag}_prof.log 2>&1 || exit 1
f=$(ls $out/${tag}_prof/*/*kernel_stats.csv | head -1)
python3 - "$f" > $out/${tag}_kernels.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'amgk' in n:
        short = n.split('amgk::')[1].replace('(anonymous namespace)::', '').split('(')[0]
        print(f"{short:40s} calls {r['Calls']:>4s}  total {int(r['TotalDurationNs'])/1e6:8.3f} ms  largest launch {int(r['MaxNs'])/1e6:8.3f} ms")
PY
grep -v "level [1-9]" $out/${tag}_phases.log | tail -14; cat $out/${tag}_kernels.txt
