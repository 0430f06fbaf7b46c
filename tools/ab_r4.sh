# development aid: this tree against another build of the library (jck-generation_amd/lib_prev/libjckgan_hip.so - e.g. round 4's, built
# from its commit; symbols it lacks are skipped) on one box, alternating; the old library runs without the one-GPU prefetch, as it did:
# bash tools/ab_r4.sh [bench args]
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  JCKGAN_LIB=$PWD/jck-generation_amd/lib_prev/libjckgan_hip.so JCKGAN_ALLOW_PARTIAL=1 JCK_PREFETCH_SINGLE=0 python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-roofline --no-secondary "$@" 2>gpurun_out/ab_r4.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=prev', d['ms_per_step'], d['value'])"
  python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-roofline --no-secondary "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=cur', d['ms_per_step'], d['value'])"
done
