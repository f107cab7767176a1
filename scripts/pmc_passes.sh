#!/bin/bash
# usage: scripts/pmc_passes.sh OUTDIR "CTR1 CTR2" "CTR3 ..."   (one rocprofv3 --pmc pass per quoted group)
set -u
out=$1; shift
repo=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $repo/$out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp -d $repo/$out/pass$i --output-format csv -- python3 $repo/scripts/pmc_target.py > $repo/$out/pass$i.log 2>&1 || echo "pass $i ($grp) failed"
  echo "pass $i done: $grp"
done
