// dev probe (round 5): how long does a tagged 16-byte granule take from the store of one block to the successful poll of another --
// between two XCDs and inside one -- for the cache policies the persistent EM loop could use?
// Question behind it: k_em_persist's step is ~2.5 us longer than its phases add up to; the tiles of neighbouring windows sit on
// different XCDs (block b runs on XCD b mod 8).  If a granule travels much faster between two blocks of ONE XCD (through its L2),
// tiles should be dealt to blocks XCD by XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/xcd_pingpong_probe.hip -o tools/probes/xcd_pingpong_probe.bin
// Two blocks of a 16-block launch play ping-pong over two granules (block A stores tag k into g0, B polls g0 until it sees k, stores
// k into g1, A polls g1 ...); the others leave at once.  One lane per block does it (the latency of one chain, not a throughput).
// Reported: ns per ONE-WAY trip (round trip / 2), and the XCC_ID each of the two blocks read from the hardware register.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int gr4 __attribute__((ext_vector_type(4)));

// AUX: the cache-policy bits of the buffer instructions (gfx940+): 1 = sc0, 2 = slc/nt, 16 = sc1
template <int LD_AUX, int ST_AUX>
__global__ void __launch_bounds__(64) k_pingpong(void* buf, uint32_t bytes, int blk_a, int blk_b, uint32_t trips, unsigned long long* out) {
    const int b = blockIdx.x;
    if (b != blk_a && b != blk_b) return;
    if (threadIdx.x != 0) return;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(buf, 0, bytes, 0x00020000);
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;          // HW_REG_XCC_ID
    const bool is_a = b == blk_a;
    const uint32_t mine = is_a ? 0u : 4096u, theirs = is_a ? 4096u : 0u;            // two granules, far apart (different lines)
    const unsigned long long t0 = wall_clock64();
    for (uint32_t k = 1; k <= trips; ++k) {
        if (is_a) { gr4 w = {k, k, k, k}; __builtin_amdgcn_raw_buffer_store_b128(w, rx, mine, 0, ST_AUX); }
        gr4 g;
        uint32_t spins = 0;
        do { asm volatile("" ::: "memory"); g = __builtin_amdgcn_raw_buffer_load_b128(rx, theirs, 0, LD_AUX); } while ((g.y != k || g.w != k) && ++spins < (1u << 22));
        if (spins >= (1u << 22)) { out[8 + (is_a ? 0 : 1)] = k; break; }          // gave up: the policy does not make the store visible
        if (!is_a) { gr4 w = {k, k, k, k}; __builtin_amdgcn_raw_buffer_store_b128(w, rx, mine, 0, ST_AUX); }
    }
    const unsigned long long t1 = wall_clock64();
    out[is_a ? 0 : 1] = t1 - t0;
    out[is_a ? 2 : 3] = xcc;
}

template <int LD_AUX, int ST_AUX>
static void run(void* buf, unsigned long long* d_out, int a, int b, const char* what) {
    const uint32_t trips = 2000;
    unsigned long long h[10];
    CK(hipMemset(buf, 0, 8192)); CK(hipMemset(d_out, 0, sizeof(h)));
    hipLaunchKernelGGL((k_pingpong<LD_AUX, ST_AUX>), dim3(16), dim3(64), 0, 0, buf, 8192u, a, b, trips, d_out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    // wall_clock64: 100 MHz
    if (h[8] || h[9]) printf("  %-34s blocks %2d (XCD %llu) <-> %2d (XCD %llu): GAVE UP at trip %llu / %llu\n", what, a, h[2], b, h[3], h[8], h[9]);
    else printf("  %-34s blocks %2d (XCD %llu) <-> %2d (XCD %llu): %7.0f ns one way\n", what, a, h[2], b, h[3], (double)h[0] * 10.0 / trips / 2.0);
}

int main() {
    void* buf; unsigned long long* d_out;
    CK(hipMalloc(&buf, 8192)); CK(hipMalloc(&d_out, 128));
    for (int pass = 0; pass < 2; ++pass) {
        printf("pass %d\n", pass);
        run<16, 16>(buf, d_out, 0, 1, "load sc1 / store sc1");
        run<16, 16>(buf, d_out, 0, 8, "load sc1 / store sc1");
        run<1, 16>(buf, d_out, 0, 1, "load sc0 / store sc1");
        run<1, 16>(buf, d_out, 0, 8, "load sc0 / store sc1");
        run<1, 1>(buf, d_out, 0, 8, "load sc0 / store sc0");
        run<1, 0>(buf, d_out, 0, 8, "load sc0 / store plain");
        run<17, 17>(buf, d_out, 0, 1, "load sc0 sc1 / store sc0 sc1");
        run<17, 17>(buf, d_out, 0, 8, "load sc0 sc1 / store sc0 sc1");
    }
    return 0;
}
