# dev: A/B of two builds of libsfgpu.so on one box (variants/libsfgpu_old.so against the tree's): bench step and the sweep probe
for i in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export SFGPU_LIB_PATH=$GRAFT_REPO_ROOT/sailfish_amd/csrc/variants/libsfgpu_old.so; else unset SFGPU_LIB_PATH; fi
  timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-sampling --no-host-pinned 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_em_sweep',{}); print('$lib', round(d['ms_per_step'],3), 'sweep us', round(r.get('avg_launch_ms',0)*1e3,2), 'loop us/iter', d.get('em_loop_us_per_iter'), {k: (round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('t_') or k.endswith('_ms')})"
  timeout 200 python tools/em_probe.py 2>&1 | tail -1
done; done
