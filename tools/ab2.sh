# development aid: A/B of the bench with a set of switches all 0 / all 1 on the same box.  usage: bash tools/ab2.sh "VAR1 VAR2" [bench args]
cd "$GRAFT_REPO_ROOT"
VARS=$1; shift
for i in 1 2 3; do
for v in 0 1; do
  for k in $VARS; do export $k=$v; done
  python bench.py --steps 600 --warmup 40 --no-cpu-baseline --no-roofline --no-secondary "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VARS=$v', d['ms_per_step'], d['value'])"
done; done
