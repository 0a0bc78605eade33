# dev: Gibbs phase times (SFGPU_TIMING=1) of the library variants named on the command line
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/sailfish_amd/csrc/variants/libsfgpu_$v.so; [ $v = main ] && lib=$GRAFT_REPO_ROOT/sailfish_amd/csrc/libsfgpu.so
  echo "== $v"; SFGPU_LIB_PATH=$lib GIBBS_CHAINS=${GIBBS_CHAINS:-1024} SFGPU_TIMING=1 timeout -s KILL 90 python tools/cfg5_probe.py 2>&1 | grep -E 'gibbs timing: (init|rounds)' | tail -2
done
