#!/bin/bash
# rocprofv3 kernel trace of the default bench command (run on the GPU box from the repo root)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/prof_bench.log 2>&1
ls -la gpurun_out/prof/* | head
