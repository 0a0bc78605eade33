# dev (round 4): what the route pass's stores cost after the compact format -- no stores / stores folded into 1 MB / real
R=${GRAFT_REPO_ROOT:-/root/repo}
source $R/tools/r4_variants_ab.sh skip
run base main SFGPU_EQ_PIPE=0 SFGPU_X_ROUTE_ONLY=1
run nostore nostore SFGPU_EQ_PIPE=0 SFGPU_X_ROUTE_ONLY=1
run fold fold SFGPU_EQ_PIPE=0 SFGPU_X_ROUTE_ONLY=1
