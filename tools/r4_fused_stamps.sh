# dev: phase time stamps inside the (fused / unfused) sweep of cfg3's EM loop -- needs tools/em_variants.sh stamp:"-DSFGPU_X_STAMP"
export SFGPU_LIB_PATH=$PWD/sailfish_amd/csrc/variants/libsfgpu_stamp.so
for f in 1 0; do SFGPU_EM_FUSED=$f python bench.py --steps 3 --warmup 1 --no-host-pinned --no-cpu-baseline 2>&1 | grep "^stamps" | tail -3; done
