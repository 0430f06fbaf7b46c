# development aid (JCK_DIAG build): ablation of the persistent gather-GEMM - what each part of a launch costs
cd "$GRAFT_REPO_ROOT"
for v in 4 101 102 103 104 105 106 107 108; do python tools/mb2.py igemm_dbg 0 $v down2,down3,up3,up4 256 768; done 2>&1 | grep -v amdgpu | sed 's/ | max rel diff.*//'
