#!/bin/bash
# A/B of environment toggles for the CGAN bench inside ONE gpurun call (development aid)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for cfg in "JCK_GRAPH=0" "JCK_GRAPH=1" "JCK_GRAPH=1 JCK_CGAN_SIDE=0" "JCK_GRAPH=1 JCK_CGAN_SIDE=0 JCK_WGRAD_SIDE=0" "JCK_GRAPH=0 JCK_CGAN_SIDE=0 JCK_WGRAD_SIDE=0"; do
    v=$(env $cfg timeout -k 10 200 python bench.py --model cgan --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['launch_mode'])")
    echo "$cfg -> $v"
  done
done
