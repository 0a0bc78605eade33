R=${GRAFT_REPO_ROOT:-/root/repo}
source $R/tools/r4_variants_ab.sh skip
run base main
run nophase2 nop2
run base main
run nophase2 nop2
