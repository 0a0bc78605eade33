// dev probe: throughput of random-address LDS operations on gfx950, per CU (two 1024-thread blocks per CU, as the EM sweep runs)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/lds_atomic_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int kWin = 1024, kIters = 256;
template <int MODE, int PAT>
__global__ void __launch_bounds__(1024) k(double* out, int span) {
    __shared__ double a[kWin];
    __shared__ float f[kWin];
    __shared__ unsigned long long u[kWin];
    for (int i = threadIdx.x; i < kWin; i += 1024) { a[i] = 0; f[i] = 0; u[i] = 0; }
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    double acc = 0;
    if (PAT >= 10) {
        // addresses fixed per lane outside the loop: nothing but the LDS instruction in it
        uint32_t jj[8];
        for (int q = 0; q < 8; ++q) {
            s = s * 1664525u + 1013904223u;
            jj[q] = (s >> 10) % (uint32_t)span;
            if (PAT == 11) jj[q] = ((uint32_t)__shfl((int)jj[q], 0) + 7u * (threadIdx.x & 63u)) & (kWin - 1);
            if (PAT == 12) jj[q] = ((threadIdx.x & 63u) + 64u * q) & (kWin - 1);                  // consecutive doubles: conflict free
        }
        for (int i = 0; i < kIters / 8; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (MODE == 0) atomicAdd(&a[jj[q]], 1.0);
                else if (MODE == 3) { acc += a[jj[q]]; }
                else if (MODE == 5) a[jj[q]] = acc;
            }
            if (MODE == 3) { __builtin_amdgcn_s_waitcnt(0); }
        }
        __syncthreads();
        if (threadIdx.x < kWin) out[blockIdx.x * kWin + threadIdx.x] = a[threadIdx.x] + acc;
        return;
    }
    for (int i = 0; i < kIters; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t j = (s >> 10) % (uint32_t)span;
        if (PAT == 1) j = ((uint32_t)__shfl((int)j, 0) + 7u * (threadIdx.x & 63u)) & (kWin - 1);      // one class per wavefront: base + 7 lane
        if (PAT == 2) j = ((uint32_t)__shfl((int)j, (threadIdx.x & 63u) & ~7u) + 7u * (threadIdx.x & 7u)) & (kWin - 1);   // 8 classes of 8 members
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
template <int MODE, int PAT = 0> void run(const char* name, double* d, int span) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, PAT><<<512, 1024>>>(d, span); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k<MODE, PAT><<<512, 1024>>>(d, span);
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
    run<0, 1>("ds_add_f64 base+7*lane", d, 1024);
    run<0, 2>("ds_add_f64 8 x (base+7*j)", d, 1024);
    run<3, 1>("ds_read_b64 base+7*lane", d, 1024);
    run<3, 2>("ds_read_b64 8 x (base+7*j)", d, 1024);
    printf("-- addresses fixed outside the loop\n");
    run<0, 10>("ds_add_f64 random", d, 1024);
    run<0, 10>("ds_add_f64 random", d, 300);
    run<0, 11>("ds_add_f64 base+7*lane", d, 1024);
    run<0, 12>("ds_add_f64 consecutive", d, 1024);
    run<3, 10>("ds_read_b64 random", d, 1024);
    run<3, 12>("ds_read_b64 consecutive", d, 1024);
    run<5, 10>("ds_write_b64 random", d, 1024);
    return 0;
}
