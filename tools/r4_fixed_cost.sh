# dev (round 4): the per-launch fixed cost of the partition passes -- sub-batches of 4 M reads (insert ~ fixed + 30 us)
R=${GRAFT_REPO_ROOT:-/root/repo}
source $R/tools/r4_variants_ab.sh skip
run sb4M main SFGPU_EQ_SUBBATCH=4194304
run sb16M main SFGPU_EQ_SUBBATCH=16777216
run default main
