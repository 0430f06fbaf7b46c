#!/bin/bash
# rocprofv3 kernel trace + stats of the CGAN bench (development aid); results under gpurun_out/prof_cgan
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof_cgan
JCK_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cgan -o bench -- python3 bench.py --model cgan --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/prof_cgan/bench_stdout.log 2>&1
ls gpurun_out/prof_cgan | head
