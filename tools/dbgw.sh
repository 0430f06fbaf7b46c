# development aid (JCK_DIAG build): ablation of the wave-specialised weight-gradient kernel
cd "$GRAFT_REPO_ROOT"
for v in 1 2 3 4; do python tools/mb2.py wgrad_dbg 0 $v wg2,wg3,wg4 256 512; done 2>&1 | grep -v amdgpu | sed 's/ | max rel diff.*//'
