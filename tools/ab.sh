# development aid: A/B runs of the bench under two settings of one environment switch on the same box
# usage (on the GPU box): bash tools/ab.sh VAR A B [--model cgan]
cd "$GRAFT_REPO_ROOT"
VAR=$1; A=$2; B=$3; shift 3
for i in 1 2; do
  for v in "$A" "$B"; do
    env "$VAR=$v" python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-roofline --no-secondary "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['value'])"
  done
done
