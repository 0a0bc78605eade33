// dev probe (round 4): what does a SCATTERED write of U contiguous bytes cost the memory system of an MI355X, for U = 16 .. 256?
// Question behind it: pass 1 of the class build (k_part_route) writes 16-byte granules to ~0.5 M different bins; its lines leave
// the L2 as partial 32-byte sectors.  Is the pass bound by the NUMBER of such writes (then whole 64-byte units through LDS
// rings would write 4x the bytes per request) or by bytes?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/scatter_write_probe.hip -o tools/probes/scatter_write_probe.bin
// Each "unit" of U bytes is written by U/16 adjacent lanes (16 bytes each: the TA merges them into one request); unit i goes to
// slot perm(i) of a buffer of n_units x U bytes with perm = multiply by a large odd number modulo a power of two: every unit is
// written exactly once, neighbours in time are far apart in memory.  Also: the same pattern READ (scattered loads of U bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LANES_PER_UNIT, bool READ>
__global__ void __launch_bounds__(1024) k_scatter(uint4* buf, uint64_t n_units, uint64_t mask, uint64_t mult, uint4* sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t t = tid; t < n_units * LANES_PER_UNIT; t += stride) {
        const uint64_t unit = t / LANES_PER_UNIT, part = t % LANES_PER_UNIT;
        const uint64_t slot = (unit * mult) & mask;
        uint4* p = buf + slot * LANES_PER_UNIT + part;
        if (READ) { const uint4 v = *p; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
        else *p = make_uint4((uint32_t)t, (uint32_t)unit, (uint32_t)part, 7u);
    }
    if (READ && acc.x == 0x12345678u) sink[0] = acc;
}
__global__ void k_stream_write(uint4* buf, uint64_t n) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) buf[t] = make_uint4((uint32_t)t, 1, 2, 3);
}

template <int L, bool READ>
static void run(uint4* buf, uint64_t bytes, uint4* sink, const char* what) {
    const uint64_t n_units = bytes / (16ull * L);            // a power of two
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_scatter<L, READ>), dim3(512), dim3(1024), 0, 0, buf, n_units, n_units - 1, 0x9E3779B97F4A7C15ull | 1ull, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%s unit %4d B: %8.3f ms  %7.1f GB/s  %7.2f G units/s\n", what, 16 * L, best, bytes / best / 1e6, n_units / best / 1e6);
}

int main() {
    const uint64_t bytes = 4ull << 30;                       // 4 GiB: far beyond L2 (32 MB) and the Infinity Cache (256 MB)
    uint4* buf; uint4* sink; CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_stream_write, dim3(2048), dim3(1024), 0, 0, buf, bytes / 16);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it == 2) printf("streaming write of 4 GiB: %.3f ms = %.1f GB/s\n", ms, bytes / ms / 1e6);
    }
    run<1, false>(buf, bytes, sink, "scattered WRITE"); run<2, false>(buf, bytes, sink, "scattered WRITE"); run<4, false>(buf, bytes, sink, "scattered WRITE");
    run<8, false>(buf, bytes, sink, "scattered WRITE"); run<16, false>(buf, bytes, sink, "scattered WRITE");
    run<1, true>(buf, bytes, sink, "scattered READ "); run<2, true>(buf, bytes, sink, "scattered READ "); run<4, true>(buf, bytes, sink, "scattered READ ");
    run<8, true>(buf, bytes, sink, "scattered READ "); run<16, true>(buf, bytes, sink, "scattered READ ");
    // a working set the Infinity Cache holds (128 MiB): do partial writes merge there?
    const uint64_t small = 128ull << 20;
    printf("-- 128 MiB working set (inside the 256 MiB Infinity Cache), written 32 times over\n");
    for (int L : {1, 4}) {
        const uint64_t n_units = small / (16ull * L);
        float best = 1e30f;
        for (int it = 0; it < 3; ++it) {
            CK(hipEventRecord(a));
            for (int rep = 0; rep < 32; ++rep) {
                if (L == 1) hipLaunchKernelGGL((k_scatter<1, false>), dim3(512), dim3(1024), 0, 0, buf, n_units, n_units - 1, 0x9E3779B97F4A7C15ull | 1ull, sink);
                else hipLaunchKernelGGL((k_scatter<4, false>), dim3(512), dim3(1024), 0, 0, buf, n_units, n_units - 1, 0x9E3779B97F4A7C15ull | 1ull, sink);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("scattered WRITE unit %4d B into 128 MiB: %8.3f ms per 4 GiB  %7.1f GB/s\n", 16 * L, best, 32.0 * small / best / 1e6);
    }
    return 0;
}
