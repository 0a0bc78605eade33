// dev probe (round 6): does a full-chip launch of 1024-thread blocks get ALL its blocks resident when another stream's small kernels are
// in flight while it starts?
// Question behind it: the persistent EM loop (csrc/em_persist.h) needs every one of its 512 blocks resident; bootstrap lanes that ran it
// next to each other's kernels gave up after ~50 ms because 10 - 45 blocks, the last-dispatched ones of one to three XCDs, started only
// when the others had left (profiles/r6_em_notes.md 4).  Nothing else was on the chip after the first 0.3 ms -- so they were not waiting
// for space somebody held.  This probe shows the same thing without the library.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/residency_probe.hip -o tools/probes/residency_probe.bin
// Kernel A: 2 x #CU blocks of 1024 threads, 64 VGPRs (8 wavefronts per SIMD: two blocks fill a CU), `lds` bytes of LDS each.  Every block
// stamps its start time, adds itself to a counter and waits until the counter says ALL have arrived -- or `limit_us` have passed.
// Kernel B (another stream, optional): `nb` blocks of 256 threads that spin for `b_us` each, launched `lead_us` BEFORE kernel A, in
// `nk` back-to-back launches -- the neighbours' resampling kernels.
// Reported per configuration: blocks of A that started within 1 ms / later / never before the limit, and the latest start.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_full(unsigned long long* start, unsigned int* arrived, unsigned int n_blocks, unsigned long long limit_ticks, unsigned int* gave_up) {
    extern __shared__ double lds[];
    asm volatile("v_mov_b32 v63, 0" ::: "v63");                          // (the kernel is allocated 64 VGPRs, like k_em_persist)
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        start[blockIdx.x] = t0;
        lds[0] = 0.0;
        __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_blocks) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > limit_ticks) { __hip_atomic_fetch_add(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}
template <int VG>
__global__ void __launch_bounds__(256) k_small(unsigned long long ticks, float* sink) {
    float v[VG];                                                        // (keeps VG registers live: the small kernel's allocation granule differs from A's)
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < VG; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
    int dev = 0, n_cu = 0;
    CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const unsigned n_blocks = 2u * (unsigned)n_cu;
    const size_t lds = argc > 1 ? (size_t)atol(argv[1]) : 54000;
    const double limit_us = 20000.0;
    unsigned long long* d_start; unsigned int *d_arr, *d_gu; float* d_sink;
    CK(hipMalloc(&d_start, n_blocks * 8)); CK(hipMalloc(&d_arr, 4)); CK(hipMalloc(&d_gu, 4)); CK(hipMalloc(&d_sink, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_full), hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    printf("device: %d CUs, kernel A = %u blocks x 1024 threads, %zu bytes of LDS each (two per CU), limit %.0f ms; clock 100 MHz\n", n_cu, n_blocks, lds, limit_us / 1e3);
    struct Cfg { const char* name; int vg; unsigned nb; double b_us; int nk; double lead_us; };
    const Cfg cfgs[] = {
        {"alone", 0, 0, 0, 0, 0},
        {"B: 40-VGPR blocks, 4096 x 20 us, 8 launches, A starts 30 us into them", 40, 4096, 20, 8, 30},
        {"B: 24-VGPR blocks, 4096 x 20 us, 8 launches, A starts 30 us into them", 24, 4096, 20, 8, 30},
        {"B: 40-VGPR blocks, 16384 x 5 us, 16 launches, A starts 10 us into them", 40, 16384, 5, 16, 10},
        {"B: 40-VGPR blocks, 1024 x 100 us, 2 launches, A starts 50 us into them", 40, 1024, 100, 2, 50},
    };
    for (const Cfg& c : cfgs) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(d_start, 0, n_blocks * 8, sa)); CK(hipMemsetAsync(d_arr, 0, 4, sa)); CK(hipMemsetAsync(d_gu, 0, 4, sa));
            CK(hipStreamSynchronize(sa));
            for (int k = 0; k < c.nk; ++k) {
                const unsigned long long ticks = (unsigned long long)(c.b_us * 100.0);
                if (c.vg == 40) hipLaunchKernelGGL(k_small<40>, dim3(c.nb), dim3(256), 0, sb, ticks, d_sink);
                else hipLaunchKernelGGL(k_small<24>, dim3(c.nb), dim3(256), 0, sb, ticks, d_sink);
            }
            if (c.nk) {                                                 // (busy-wait on the host: A goes out `lead_us` after B's first launch)
                const auto t0 = std::clock();
                while ((double)(std::clock() - t0) / CLOCKS_PER_SEC * 1e6 < c.lead_us) { }
            }
            hipLaunchKernelGGL(k_full, dim3(n_blocks), dim3(1024), lds, sa, d_start, d_arr, n_blocks, (unsigned long long)(limit_us * 100.0), d_gu);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
            std::vector<unsigned long long> h(n_blocks); unsigned gu = 0;
            CK(hipMemcpy(h.data(), d_start, n_blocks * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&gu, d_gu, 4, hipMemcpyDeviceToHost));
            const unsigned long long tmin = *std::min_element(h.begin(), h.end()), tmax = *std::max_element(h.begin(), h.end());
            unsigned early = 0, late = 0;
            for (unsigned long long t : h) { if ((t - tmin) * 0.01 < 1000.0) ++early; else ++late; }
            printf("%-75s run %d: %u blocks started within 1 ms, %u later (latest %.1f us after the first), %u blocks gave up waiting\n", c.name, rep, early, late, (tmax - tmin) * 0.01, gu);
        }
    }
    return 0;
}
