#!/bin/bash
# Round profile: rocprofv3 kernel trace + stats of the default bench command, then separate PMC passes (FETCH_SIZE, WRITE_SIZE)
# as the MI355X guide prescribes.  Run on the GPU box from the repo root; results under gpurun_out/prof.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/prof/bench_stdout.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof -o pmc_fetch -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof -o pmc_write -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/prof/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/prof -o pmc_sq -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/prof/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof -o pmc_mfma -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/prof/pmc_mfma.log 2>&1
ls gpurun_out/prof | head -30
