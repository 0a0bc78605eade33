R=${GRAFT_REPO_ROOT:-/root/repo}
source $R/tools/r4_variants_ab.sh skip
run quad main SFGPU_EQ_QUAD=1
run direct main SFGPU_EQ_QUAD=0
run quad main SFGPU_EQ_QUAD=1
run direct main SFGPU_EQ_QUAD=0
