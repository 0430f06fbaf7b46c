# development aid: the fp32 parity path's step under several values of one environment switch: bash tools/abf32.sh VAR v1 v2 ...
cd "$GRAFT_REPO_ROOT"
VAR=$1; shift
for i in 1 2; do
  for v in "$@"; do
    env "$VAR=$v" python bench.py --prec f32 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['value'])"
  done
done
