#!/bin/bash
# development aid: the determinism tests under several switch settings, a few runs each (usage: tools/det.sh "ENV=.." ...; DET_K / DET_N)
cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  for rep in $(seq 1 ${DET_N:-3}); do
    r=$(env $cfg timeout -k 10 200 python -m pytest tests/test_bf16_envelope.py -q -k "${DET_K:-two_runs}" 2>&1 | tail -1)
    echo "$cfg [$rep] -> $r"
  done
done
