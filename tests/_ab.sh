#!/bin/bash
# A/B of environment toggles inside ONE gpurun call (box-to-box variance is ~3%): usage  tests/_ab.sh "VAR=a" "VAR=b" ...
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for cfg in "$@"; do
    v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline ${AB_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d.get('launch_mode'))")
    echo "$cfg -> $v"
  done
done
