# dev (round 6): sub-batches queued ahead of the host (SFGPU_EQ_PIPE=2, the product's default) against one round trip per sub-batch (0) and the
# two-stream form (1): wall time of add + finish + export (tools/eq_probe.py, no profiler), cfg3 and cfg2, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
export SFGPU_LIB_PATH=$R/sailfish_amd/csrc/variants/libsfgpu_base.so
for rep in 1 2 3; do
  for m in 0 2 1; do
    echo "cfg3 pipe=$m: $(SFGPU_EQ_PIPE=$m EQ_CFG3=1 python $R/tools/eq_probe.py 2>&1 | tail -1 | cut -c1-90)"
  done
done
for rep in 1 2; do
  for m in 0 2; do
    echo "cfg2 pipe=$m: $(SFGPU_EQ_PIPE=$m python $R/tools/eq_probe.py 2>&1 | tail -1 | cut -c1-90)"
  done
done
