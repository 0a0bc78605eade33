// dev probe: throughput of random-address LDS operations on gfx950, per CU (two 1024-thread blocks per CU, as the EM sweep runs)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/lds_atomic_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int kWin = 1024, kIters = 256;
template <int MODE>
__global__ void __launch_bounds__(1024) k(double* out, int span) {
    __shared__ double a[kWin];
    __shared__ float f[kWin];
    __shared__ unsigned long long u[kWin];
    for (int i = threadIdx.x; i < kWin; i += 1024) { a[i] = 0; f[i] = 0; u[i] = 0; }
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    double acc = 0;
    for (int i = 0; i < kIters; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t j = (s >> 10) % (uint32_t)span;
        if (MODE == 0) atomicAdd(&a[j], 1.0 + i);
        else if (MODE == 1) atomicAdd(&f[j], 1.0f + i);
        else if (MODE == 2) atomicAdd(&u[j], 1ull + i);
        else if (MODE == 3) acc += a[j];
        else if (MODE == 4) { double o = atomicAdd(&a[j], 1.0 + i); acc += o; }
        else if (MODE == 5) a[j] = 1.0 + i;
    }
    __syncthreads();
    if (threadIdx.x < kWin) out[blockIdx.x * kWin + threadIdx.x] = a[threadIdx.x] + f[threadIdx.x] + (double)u[threadIdx.x] + acc;
}
template <int MODE> void run(const char* name, double* d, int span) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<512, 1024>>>(d, span); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k<MODE><<<512, 1024>>>(d, span);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 10;
    const double lane_ops_per_cu = 2.0 * 1024 * kIters;
    printf("%-28s span %5d: %8.2f us  -> %.3f cycles per lane-op per CU (2.4 GHz)\n", name, span, us, us * 2400.0 / lane_ops_per_cu);
}
int main() {
    double* d; hipMalloc(&d, 512 * kWin * sizeof(double));
    for (int span : {1024, 64}) {
        run<0>("ds_add_f64 (no return)", d, span);
        run<4>("ds_add_rtn_f64", d, span);
        run<1>("ds_add_f32 (no return)", d, span);
        run<2>("ds_add_u64 (no return)", d, span);
        run<3>("ds_read_b64", d, span);
        run<5>("ds_write_b64", d, span);
    }
    return 0;
}
