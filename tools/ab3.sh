# development aid: bench under several values of one environment switch on the same box: bash tools/ab3.sh VAR v1 v2 v3 ...
cd "$GRAFT_REPO_ROOT"
VAR=$1; shift
for i in 1 2; do
  for v in "$@"; do
    env "$VAR=$v" python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-roofline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['value'])"
  done
done
