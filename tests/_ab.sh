cd $GRAFT_REPO_ROOT
python -m pytest tests/test_cgan_gpu.py tests/test_bf16_envelope.py -q -x -k "cgan" > gpurun_out/t_lazy.log 2>&1; echo "tests rc=$?" 
for i in 1 2; do
JCK_LAZY_JOIN=0 python bench.py --model cgan --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy0', d['ms_per_step'], d['value'])"
JCK_LAZY_JOIN=1 python bench.py --model cgan --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy1', d['ms_per_step'], d['value'])"
done
for p in 0 1 -1; do
JCK_SIDE_PRIO=$p python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio$p', d['ms_per_step'], d['value'])"
JCK_SIDE_PRIO=$p python bench.py --model cgan --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cgan prio$p', d['ms_per_step'], d['value'])"
done
tail -5 gpurun_out/t_lazy.log
