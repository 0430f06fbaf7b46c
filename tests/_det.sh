#!/bin/bash
# development aid: the determinism test under several switch settings, a few runs each (usage: tests/_det.sh "ENV=.." ...)
cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  for rep in 1 2 3; do
    r=$(env $cfg timeout -k 10 200 python -m pytest tests/test_bf16_envelope.py -q -x -k two_runs 2>&1 | tail -1)
    echo "$cfg [$rep] -> $r"
  done
done
