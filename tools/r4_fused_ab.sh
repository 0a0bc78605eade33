# dev: timing-only variants of the fused EM iteration (cfg3 bench step; results of the X_ variants are WRONG)
#   bash tools/em_variants.sh cheapx:"-DSFGPU_X_CHEAPX" ...   then on the GPU box:  bash tools/r4_fused_ab.sh name ...
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib=$PWD/sailfish_amd/csrc/variants/libsfgpu_$v.so
  SFGPU_LIB_PATH=${lib:-$PWD/sailfish_amd/csrc/libsfgpu.so} SFGPU_TIMING=1 python bench.py --no-host-pinned --no-cpu-baseline 2>/tmp/err_$v | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), round(d['phase_ms']['em'],3), d['em_iters'], round(d['em_us_per_iter_loop'],2))"
  grep -m1 "em plan" /tmp/err_$v
done
