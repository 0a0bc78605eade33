R=${GRAFT_REPO_ROOT:-/root/repo}
source $R/tools/r4_variants_ab.sh skip
run sb4M_full main SFGPU_EQ_SUBBATCH=4194304
run sb4M_prologue_only insp SFGPU_EQ_SUBBATCH=4194304
run sb4M_no_epilogue insne SFGPU_EQ_SUBBATCH=4194304
