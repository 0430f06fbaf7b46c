cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
for v in 0 1; do
JCK_FUSE_TANH=$v JCK_FOLD_ZERO=$v python bench.py --steps 600 --warmup 40 --no-cpu-baseline --no-roofline --no-secondary "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=$v', d['ms_per_step'], d['value'])"
done; done
