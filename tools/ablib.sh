# development aid: the bench under the library of the tree against another build of it (jck-generation_amd/lib_prev/libjckgan_hip.so,
# e.g. the previous commit's), alternating on one box: bash tools/ablib.sh [bench args]
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for v in prev cur; do
    if [ $v = prev ]; then export JCKGAN_LIB=$PWD/jck-generation_amd/lib_prev/libjckgan_hip.so; else unset JCKGAN_LIB; fi
    python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-roofline --no-secondary "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', d['ms_per_step'], d['value'])"
  done
done
