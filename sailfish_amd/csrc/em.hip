// em.hip -- CollapsedEMOptimizer on the device (rows a6-a12 of SURVEY.md section 8).
//
// Replaces  src/CollapsedEMOptimizer.cpp:711-893 (optimize), :224-281 (EMUpdate_),
//           :288-369 (VBEMUpdate_), :36-44 (truncateCountVector), :849-861 (convergence).
//
// Design (the reference is TBB parallel_for over an AoS vector of heap vectors with a CAS
// loop per contribution):
//   * classes are a flat CSR (rowptr u32, ids u32, counts u32) that stays resident in HBM;
//   * the per-class aux weights of the reference are NOT materialised: w_i is proportional to
//     1/effLen_i and its per-class normaliser cancels in v_i/denom (:762-768, :256-270), so the
//     sweep gathers x_t = alpha_t/effLen_t (EM) or expTheta_t/effLen_t (VBEM) -- one M-vector
//     rebuilt each iteration -- and streams only labels and counts: 4 B per nonzero instead of
//     12 (algorithmic bytes B_iter' = 4L + 8C + 48M);
//   * an iteration is [sweep over classes] -> [per-transcript update: convergence test, swap,
//     next x]; the loop condition of :820 is evaluated ON THE DEVICE from a small state block,
//     every kernel turns into a no-op once it holds, so the host can enqueue iterations in
//     hipGraph chunks and still stop at exactly the reference's iteration;
//   * all reductions that feed decisions (n_active, sum alpha, alphaSum) are two-stage and
//     order-deterministic; the only nondeterministic order is the f64 atomic accumulation into
//     alphaOut, as in the reference's CAS loop (:70-79).
#include "common.h"
#include "primitives.h"

#include <cfloat>
#include <vector>

namespace sfgpu {

constexpr int kEmBlock = 256;
constexpr int kMaxPartials = 1024;          // blocks of the per-transcript kernels
constexpr uint32_t kDoneMark = 0xFFFFFFFFu;
constexpr double kTiny = 4.9406564584124654e-324;   // numeric_limits<double>::denorm_min() (:33-34)
constexpr double kPriorAlpha = 0.01;        // :786
constexpr double kMinAlpha = 1e-8;          // :810
constexpr double kCheckCutoff = 1e-2;       // :811

struct EmState {
    uint32_t it_a;                 // iterations completed; written by update, read by sweep
    uint32_t it_b;                 // iteration in flight (or kDoneMark); written by sweep, read by update
    uint32_t notconv[2];           // per iteration parity: some gated transcript moved by > tol
    uint32_t gated[2];             // per iteration parity: some transcript passed the gate
    unsigned long long n_active;
    double alpha_sum;
};

// loop condition of :820, negated:  stop  <=>  it >= minIter && (it >= maxIter || converged)
__device__ __forceinline__ bool em_stop(uint32_t it, const EmState* s, uint32_t min_iter, uint32_t max_iter) {
    if (it < min_iter) return false;
    if (it >= max_iter) return true;
    return it > 0 && s->notconv[(it - 1) & 1] == 0;
}

// psi(x), x > 0: recurrence up to x >= 10 then the asymptotic series through B_14
// (boost::math::digamma at :303, :314 in the reference; |err| ~ 1e-15).
__device__ __forceinline__ double digamma_pos(double x) {
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    double inv = 1.0 / x, inv2 = inv * inv;
    double s = inv2 * (1.0 / 12.0 - inv2 * (1.0 / 120.0 - inv2 * (1.0 / 252.0 - inv2 * (1.0 / 240.0
             - inv2 * (1.0 / 132.0 - inv2 * (691.0 / 32760.0 - inv2 * (1.0 / 12.0)))))));
    return r + log(x) - 0.5 * inv - s;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}

// deterministic block sum; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* lds /* kEmBlock/kWave */) {
    v = wave_sum(v);
    int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) lds[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < kEmBlock / kWave; ++i) t += lds[i];
    __syncthreads();
    return t;
}

// sum of the nb per-block partials, same order in every block -> identical everywhere
__device__ __forceinline__ double sum_partials(const double* partials, int nb, double* lds) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += kEmBlock) v += partials[i];
    double t = block_sum(v, lds);
    __shared__ double bc;
    if (threadIdx.x == 0) bc = t;
    __syncthreads();
    return bc;
}

__global__ void k_clamp_len(uint64_t M, const double* __restrict__ len, double* __restrict__ lenc) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) { double l = len[t]; lenc[t] = (l <= 1.0) ? 1.0 : l; }   // :738
}

__global__ void k_narrow_counts(uint64_t C, const uint64_t* __restrict__ c64, uint32_t* __restrict__ c32,
                                unsigned int* overflow) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    uint64_t v = c64[c];
    if (v >= 0x7FFFFFFFull) atomicOr(overflow, 1u);      // the sweep's head slot holds the count in 31 bits
    c32[c] = (uint32_t)v;
}

// :774-782  every transcript that appears in a class is active
__global__ void k_mark_active(uint64_t L, const uint32_t* __restrict__ ids, double* alpha_out) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < L) alpha_out[ids[j]] = 1.0;
}

__global__ void k_count_active(uint64_t M, const double* __restrict__ flags, double* partials) {
    __shared__ double lds[kEmBlock / kWave];
    double v = 0.0;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock)
        v += (flags[t] > 0.0) ? 1.0 : 0.0;
    double s = block_sum(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// :800-803 alpha_t = active ? (1/n_active) * numMapped : 0 ; also x for iteration 0 (EM) or
// the partial sums of alpha (VBEM)
template <bool VB>
__global__ void k_init_alpha(uint64_t M, double* alpha, double* alpha_out, double* x, const double* __restrict__ lenc,
                             double total_frags, const double* act_partials, double* sum_partials_out, int nb,
                             EmState* st) {
    __shared__ double lds[kEmBlock / kWave];
    double n_act = sum_partials(act_partials, nb, lds);
    double scale = 1.0 / n_act;
    double local = 0.0;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock) {
        double a = (alpha_out[t] > 0.0) ? scale * total_frags : 0.0;
        alpha[t] = a; alpha_out[t] = 0.0;
        if (VB) local += a; else x[t] = a / lenc[t];
    }
    if (VB) { double s = block_sum(local, lds); if (threadIdx.x == 0) sum_partials_out[blockIdx.x] = s; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->it_a = 0; st->it_b = 0; st->notconv[0] = st->notconv[1] = 0; st->gated[0] = st->gated[1] = 0;
        st->n_active = (unsigned long long)n_act; st->alpha_sum = 0.0;
    }
}

// VBEMUpdate_ prologue (:300-320): expTheta_t = alpha_t > denorm_min ? exp(psi(alpha_t) - psi(sum alpha)) : 0,
// folded with the 1/effLen factor of the aux weight.
__global__ void k_vb_prepare(uint64_t M, const double* __restrict__ alpha, double* __restrict__ x,
                             const double* __restrict__ lenc, const double* sum_partials_in, int nb,
                             const EmState* st, int force) {
    if (!force && st->it_b == kDoneMark) return;
    __shared__ double lds[kEmBlock / kWave];
    double asum = sum_partials(sum_partials_in, nb, lds);
    double log_norm = digamma_pos(asum);
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock) {
        double a = alpha[t];
        x[t] = (a > kTiny) ? exp(digamma_pos(a) - log_norm) / lenc[t] : 0.0;
    }
}

// E-step sweep, one lane per class (EMUpdate_ :236-277 / VBEMUpdate_ :325-366).
template <bool VB>
__global__ void __launch_bounds__(kEmBlock)
k_sweep_lane(uint64_t C, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
             const uint32_t* __restrict__ counts, const double* __restrict__ x, double* alpha_out,
             EmState* st, uint32_t min_iter, uint32_t max_iter) {
    uint32_t it = st->it_a;
    bool stop = em_stop(it, st, min_iter, max_iter);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->it_b = stop ? kDoneMark : it;
        if (!stop) { st->notconv[it & 1] = 0; st->gated[it & 1] = 0; }
    }
    if (stop) return;
    uint64_t c = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x;
    if (c >= C) return;
    uint32_t b = rowptr[c], e = rowptr[c + 1];
    double cnt = (double)counts[c];
    if (e - b == 1) { atomicAdd(&alpha_out[ids[b]], cnt); return; }     // :275 / :364
    if (e == b) return;
    double denom = 0.0;
    for (uint32_t j = b; j < e; ++j) {
        double v = x[ids[j]];
        if (VB) { if (v > 0.0) denom += v; } else denom += v;
    }
    if (!(denom > kTiny)) return;                                       // :260 / :349
    double inv = cnt / denom;                                           // :264
    for (uint32_t j = b; j < e; ++j) {
        uint32_t t = ids[j];
        double v = x[t];
        if (VB ? (v > 0.0) : (v == v)) atomicAdd(&alpha_out[t], v * inv);
    }
}

// ---- tiled, wave-segmented E-step sweep --------------------------------------------------------
// Classes are stored in the canonical order (first id ascending), so a run of consecutive classes
// touches a narrow band of transcripts.  Once per problem the CSR is re-packed into the layout the
// sweep streams:
//   * a tile is the run of classes whose first nonzero falls into one kTileNnz-sized bucket of the
//     CSR (nnz-balanced); its window is the band [lo, lo+span) of transcripts it touches (<= kWin);
//   * inside a tile the labels are laid out in 64-slot chunks -- one chunk per wavefront step --
//     padded so that no label straddles a chunk; a label of k ids takes k+1 slots: a head slot
//     (bit 31 set) carrying the class count, then the k transcript ids (0xFFFFFFFF = padding), so one
//     coalesced 256-byte load per wavefront step brings everything the step needs;
//   * labels that do not fit a chunk (> 63 ids; rare) are listed separately and handled by extra blocks.
// Per tile, one 256-thread block stages the x-window in LDS, then each wavefront walks chunks:
// coalesced slot load -> x gather from LDS -> ballot of the head flags -> segmented inclusive scan
// across the 64 lanes (fixed tree order) -> count/denom broadcast -> ds_add_f64 into the LDS window
// accumulators (EMUpdate_ :251-271 for 64 nonzeros at once, no divergent loops).  The window is
// published with plain stores and folded per transcript by the update, so the common path has no
// global atomics and is bit-reproducible.  Members outside the window take global gathers/atomics:
// any input is handled, locality only buys speed.
#ifndef SFGPU_TILE_NNZ
#define SFGPU_TILE_NNZ 2048
#endif
#ifndef SFGPU_SWEEP_BLOCK
#define SFGPU_SWEEP_BLOCK 256
#endif
#ifndef SFGPU_SWEEP_UNROLL
#define SFGPU_SWEEP_UNROLL 4
#endif
constexpr int kTileNnz = SFGPU_TILE_NNZ;     // CSR bucket that defines a tile
constexpr int kWin = 1024;                   // LDS window (transcripts): 2 x 8 KB
constexpr int kChunk = 64;                   // slots per chunk = wavefront width
constexpr int kSweepBlock = SFGPU_SWEEP_BLOCK;  // threads per tile block
constexpr int kUnroll = SFGPU_SWEEP_UNROLL;     // chunks a wavefront fetches per step
constexpr int kPackCap = 4096;               // classes a tile's packer stages in LDS
constexpr uint32_t kPad = 0xFFFFFFFFu;
constexpr uint32_t kHead = 0x80000000u;
constexpr uint32_t kLongPos = 0xFFFFFFFFu;

// tile i = classes [tile_c0[i], tile_c0[i+1]) : those with rowptr[c] in [i*kTileNnz, (i+1)*kTileNnz)
__global__ void k_tile_plan(uint64_t C, uint32_t n_tiles, const uint32_t* __restrict__ rowptr, uint32_t* tile_c0) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_tiles) return;
    uint64_t target = (uint64_t)i * kTileNnz;
    uint64_t lo = 0, hi = C;                         // first class with rowptr[c] >= target
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (rowptr[mid] >= target) hi = mid; else lo = mid + 1; }
    tile_c0[i] = (uint32_t)lo;
}

// window of tile i: [lo, lo + span) with lo = smallest member and span <= kWin
__global__ void __launch_bounds__(kEmBlock)
k_tile_window(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ tile_c0,
              uint32_t* tile_lo, uint32_t* tile_span) {
    __shared__ uint32_t rmin[kEmBlock / kWave], rmax[kEmBlock / kWave];
    uint32_t b = rowptr[tile_c0[blockIdx.x]], e = rowptr[tile_c0[blockIdx.x + 1]];
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    for (uint32_t j = b + threadIdx.x; j < e; j += kEmBlock) { uint32_t v = ids[j]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    for (int o = kWave / 2; o > 0; o >>= 1) {
        uint32_t v = __shfl_down(mn, o, kWave); mn = v < mn ? v : mn;
        uint32_t w = __shfl_down(mx, o, kWave); mx = w > mx ? w : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { rmin[threadIdx.x / kWave] = mn; rmax[threadIdx.x / kWave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kEmBlock / kWave; ++i) { mn = rmin[i] < mn ? rmin[i] : mn; mx = rmax[i] > mx ? rmax[i] : mx; }
        if (e == b) { tile_lo[blockIdx.x] = 0; tile_span[blockIdx.x] = 0; }
        else {
            uint32_t span = mx - mn + 1;
            tile_lo[blockIdx.x] = mn;
            tile_span[blockIdx.x] = span < (uint32_t)kWin ? span : (uint32_t)kWin;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) tile_span[gridDim.x] = 0;   // scan sentinel
}

// greedy chunk packing of one tile: slot position of every class (tile relative); labels that do
// not fit a chunk go to the long list.  The greedy walk is sequential, so one lane does it out of
// LDS; tiles with more classes than the LDS stage holds walk global memory (slow, rare).
__global__ void __launch_bounds__(kEmBlock)
k_tile_pack(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ tile_c0, uint32_t* slotpos,
            uint32_t* tile_chunks, uint32_t* long_cls, unsigned int* n_long) {
    __shared__ uint32_t lens[kPackCap];
    const uint32_t c0 = tile_c0[blockIdx.x], c1 = tile_c0[blockIdx.x + 1];
    const uint32_t nc = c1 - c0;
    const bool staged = nc <= (uint32_t)kPackCap;
    if (staged) for (uint32_t i = threadIdx.x; i < nc; i += kEmBlock) lens[i] = rowptr[c0 + i + 1] - rowptr[c0 + i];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t pos = 0;
        for (uint32_t i = 0; i < nc; ++i) {
            uint32_t k = staged ? lens[i] : rowptr[c0 + i + 1] - rowptr[c0 + i];
            uint32_t p;
            if (k + 1 > (uint32_t)kChunk) { p = kLongPos; long_cls[atomicAdd(n_long, 1u)] = c0 + i; }
            else if (k == 0) { p = kLongPos; }              // empty class: occupies nothing
            else {
                uint32_t rem = kChunk - (pos & (kChunk - 1));
                if (k + 1 > rem) pos += rem;                 // pad to the next chunk
                p = pos; pos += k + 1;
            }
            if (staged) lens[i] = p; else slotpos[c0 + i] = p;
        }
        tile_chunks[blockIdx.x] = (pos + kChunk - 1) / kChunk;
        if (blockIdx.x == gridDim.x - 1) tile_chunks[gridDim.x] = 0;
    }
    __syncthreads();
    if (staged) for (uint32_t i = threadIdx.x; i < nc; i += kEmBlock) slotpos[c0 + i] = lens[i];
}

__global__ void k_fill_slots(uint64_t C, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
                             const uint32_t* __restrict__ counts, const uint32_t* __restrict__ slotpos,
                             const uint64_t* __restrict__ tile_chunk0, uint32_t* slots) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    uint32_t p = slotpos[c];
    if (p == kLongPos) return;
    uint32_t b = rowptr[c], k = rowptr[c + 1] - b;
    uint64_t s = tile_chunk0[b / kTileNnz] * kChunk + p;
    slots[s] = kHead | counts[c];
    for (uint32_t m = 0; m < k; ++m) slots[s + 1 + m] = ids[b + m];
}

// one (transcript, slot) pair per window entry; sorted by transcript this is the cover list that the
// per-transcript update walks to fold the tiles' partial sums in a fixed order
__global__ void __launch_bounds__(kEmBlock)
k_cover_pairs(const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_span,
              const uint64_t* __restrict__ tile_off, uint64_t* keys, uint32_t* vals) {
    uint32_t lo = tile_lo[blockIdx.x], span = tile_span[blockIdx.x];
    uint64_t off = tile_off[blockIdx.x];
    for (uint32_t d = threadIdx.x; d < span; d += kEmBlock) { keys[off + d] = (uint64_t)lo + d; vals[off + d] = (uint32_t)(off + d); }
}

__global__ void k_cover_ptr(uint64_t M, uint64_t P, const uint64_t* __restrict__ sorted_keys, uint32_t* cov_ptr) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > M) return;
    uint64_t lo = 0, hi = P;
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (sorted_keys[mid] >= t) hi = mid; else lo = mid + 1; }
    cov_ptr[t] = (uint32_t)lo;
}

// value of `v` from the lane `N` places lower inside the same 16-lane row (0.0 where there is none):
// v_mov_b32 with the row_shr:N data-parallel-primitive modifier, one per half of the double
template <int N>
__device__ __forceinline__ double row_shr_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// inclusive scan of v over [start, lane] for every lane (start = first lane of the lane's segment),
// in a fixed order: DPP shifts inside each 16-lane row, then the three row carries in sequence
__device__ __forceinline__ double segmented_scan64(double v, int lane, int start) {
    double s = v, up;
    up = row_shr_f64<1>(s); if (lane >= start + 1) s += up;
    up = row_shr_f64<2>(s); if (lane >= start + 2) s += up;
    up = row_shr_f64<4>(s); if (lane >= start + 4) s += up;
    up = row_shr_f64<8>(s); if (lane >= start + 8) s += up;
    double c = readlane_f64(s, 15); if (lane >= 16 && lane < 32 && start < 16) s += c;
    c = readlane_f64(s, 31);        if (lane >= 32 && lane < 48 && start < 32) s += c;
    c = readlane_f64(s, 47);        if (lane >= 48 && start < 48) s += c;
    return s;
}

struct SweepArgs {
    uint32_t n_tiles;
    const uint32_t* rowptr; const uint32_t* ids; const uint32_t* counts;        // caller CSR (long labels)
    const uint32_t* slots;
    const uint64_t* tile_chunk0; const uint32_t* tile_lo; const uint32_t* tile_span; const uint64_t* tile_off;
    const uint32_t* long_cls;
    const double* x; double* alpha_out; double* partial;
    EmState* st; uint32_t min_iter, max_iter; int ablate;
};

template <bool VB>
__global__ void __launch_bounds__(kSweepBlock)
k_sweep_chunk(SweepArgs a) {
    EmState* st = a.st;
    uint32_t it = st->it_a;
    bool stop = em_stop(it, st, a.min_iter, a.max_iter);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->it_b = stop ? kDoneMark : it;
        if (!stop) { st->notconv[it & 1] = 0; st->gated[it & 1] = 0; }
    }
    if (stop) return;
    __shared__ double xs[kWin];
    __shared__ double acc[kWin];
    __shared__ double red[kSweepBlock / kWave];
    const double* __restrict__ x = a.x;
    if (blockIdx.x >= a.n_tiles) {
        // ---- one label longer than a chunk: whole block, direct global accesses
        uint32_t c = a.long_cls[blockIdx.x - a.n_tiles];
        uint32_t b = a.rowptr[c], e = a.rowptr[c + 1];
        double part = 0.0;
        for (uint32_t j = b + threadIdx.x; j < e; j += kSweepBlock) {
            double v = x[a.ids[j]];
            if (VB) { if (v > 0.0) part += v; } else part += v;
        }
        part = wave_sum(part);
        if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = part;
        __syncthreads();
        double denom = 0.0;
        for (int i = 0; i < kSweepBlock / kWave; ++i) denom += red[i];   // same order in every thread
        if (!(denom > kTiny)) return;
        double inv = (double)a.counts[c] / denom;
        for (uint32_t j = b + threadIdx.x; j < e; j += kSweepBlock) {
            uint32_t t = a.ids[j]; double v = x[t];
            if (VB ? (v > 0.0) : (v == v)) atomicAdd(&a.alpha_out[t], v * inv);
        }
        return;
    }
    const uint32_t lo = a.tile_lo[blockIdx.x];
    const uint32_t span = a.tile_span[blockIdx.x];         // members at lo + [0, span) live in the LDS window
    const uint64_t q0 = a.tile_chunk0[blockIdx.x], q1 = a.tile_chunk0[blockIdx.x + 1];
    if (span == 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned long long le_mask = (2ull << lane) - 1ull;          // lanes <= lane
    // Each wavefront walks groups of kUnroll consecutive chunks.  A group's slots are fetched with
    // kUnroll independent 256-byte loads (addresses clamped to the tile so the loads are
    // unconditional) one group ahead of their use, which keeps >= 1 KB per wavefront in flight:
    // the stream is latency-bound otherwise.
    const uint64_t qlast = q1 - 1;
    const uint64_t gstride = (uint64_t)(kSweepBlock / kWave) * kUnroll;
    uint64_t base = q0 + (uint64_t)(threadIdx.x >> 6) * kUnroll;
    uint32_t nxt[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) { uint64_t q = base + u; nxt[u] = a.slots[(q < q1 ? q : qlast) * kChunk + lane]; }
    for (uint32_t i = threadIdx.x; i < span; i += kSweepBlock) { xs[i] = x[(uint64_t)lo + i]; acc[i] = 0.0; }
    __syncthreads();
    if (!(a.ablate & 1))
    for (; base < q1; base += gstride) {
        uint32_t cur[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) cur[u] = nxt[u];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) { uint64_t q = base + gstride + u; nxt[u] = a.slots[(q < q1 ? q : qlast) * kChunk + lane]; }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (base + u >= q1) break;                                  // wave-uniform
            const uint32_t slot = cur[u];
            const bool valid = slot != kPad;
            const bool head = valid && (slot & kHead);                  // head slot: carries the class count
            const bool member = valid && !(slot & kHead);               // member slot: a transcript id
            const unsigned long long headmask = __ballot(head);
            const unsigned long long validmask = __ballot(valid);
            const uint32_t d = slot - lo;
            const bool inwin = member && d < span;
            double v = xs[inwin ? d : 0u];                              // LDS gather (ds_read_b64)
            if (!inwin) v = 0.0;
            if (__ballot(member && !inwin)) { if (member && !inwin) v = x[slot]; }   // rare: outside the window
            if (VB) { if (!(v > 0.0)) v = 0.0; }                        // expTheta == 0 terms are skipped (:344, :356)
            // segment (= head + label) geometry inside the chunk; every chunk starts with a head
            const unsigned long long below = headmask & le_mask;
            const int start = 63 - __clzll((long long)below);
            const unsigned long long above = headmask & ~le_mask;
            const int end = above ? (__ffsll((long long)above) - 2) : (__popcll(validmask) - 1);
            // segmented inclusive scan across the wavefront, fixed order
            const double s = segmented_scan64(v, lane, start);
            if (a.ablate & 2) { if (s == 123.456) acc[0] = s; continue; }
            const double total = __shfl(s, end, kWave);
            const double cnt = __shfl((double)(slot & ~kHead), start, kWave);
            double contrib;
            if (end - start == 1) contrib = cnt;                        // singleton label: the full count (:275 / :364)
            else contrib = (total > kTiny && v == v) ? v * (cnt / total) : 0.0;   // :260-270 (NaN terms skipped :269)
            if (member && contrib != 0.0 && !(a.ablate & 4)) {
                if (inwin) atomicAdd(&acc[d], contrib); else atomicAdd(&a.alpha_out[slot], contrib);
            }
        }
    }
    __syncthreads();
    // ---- publish the window (plain coalesced stores; folded per transcript by the update)
    const uint64_t off = a.tile_off[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < span; i += kSweepBlock) a.partial[off + i] = acc[i];
}

// alphaOut[t] += sum of the tiles' window entries for t, in cover-list order (deterministic)
__device__ __forceinline__ double fold_partials(uint64_t t, const uint32_t* __restrict__ cov_ptr,
                                                const uint32_t* __restrict__ cov_pos, const double* __restrict__ partial) {
    double s = 0.0;
    for (uint32_t k = cov_ptr[t], e = cov_ptr[t + 1]; k < e; ++k) s += partial[cov_pos[k]];
    return s;
}

// piecewise API: make alphaOut complete before the caller's all-reduce
__global__ void k_fold(uint64_t M, double* alpha_out, const uint32_t* __restrict__ cov_ptr,
                       const uint32_t* __restrict__ cov_pos, const double* __restrict__ partial, const EmState* st) {
    if (st->it_b == kDoneMark) return;
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) alpha_out[t] += fold_partials(t, cov_ptr, cov_pos, partial);
}

// per-transcript update (:849-861): gate, relative change, alpha <- alphaOut, alphaOut <- 0, ++it.
// No atomics: the convergence flags are idempotent plain stores (every writer stores 1) and the
// per-block maximum goes to blkmax[parity][block], reduced by the host when it polls.
template <bool VB, bool FOLD>
__global__ void __launch_bounds__(kEmBlock)
k_update(uint64_t M, double* alpha, double* alpha_out, double* x, const double* __restrict__ lenc,
         double tol, int check_mode, double* sum_partials_out, double* blkmax, EmState* st,
         const uint32_t* __restrict__ cov_ptr, const uint32_t* __restrict__ cov_pos,
         const double* __restrict__ partial) {
    uint32_t it = st->it_b;
    if (it == kDoneMark) return;
    __shared__ double lds[kEmBlock / kWave];
    __shared__ double lmax[kEmBlock / kWave];
    double local_sum = 0.0, local_max = -1.0;
    unsigned notconv = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock) {
        double a = alpha[t];
        double ap = alpha_out[t];
        if (FOLD) ap += fold_partials(t, cov_ptr, cov_pos, partial);
        if (VB) ap += kPriorAlpha;                     // alphaOut starts at the prior (:318)
        double gate = check_mode ? a : ap;             // :852 vs :499
        if (gate > kCheckCutoff) {
            double rel = fabs(a - ap) / ap;
            if (rel > local_max) local_max = rel;      // NaN never wins, as in the reference (:854)
            if (rel > tol) notconv = 1;
            if (local_max < 0.0) local_max = 0.0;      // gated at least once
        }
        alpha[t] = ap; alpha_out[t] = 0.0;
        if (VB) local_sum += ap; else x[t] = ap / lenc[t];
    }
    for (int o = kWave / 2; o > 0; o >>= 1) {
        double m = __shfl_down(local_max, o, kWave); if (m > local_max) local_max = m;
        notconv |= __shfl_down(notconv, o, kWave);
    }
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (notconv) st->notconv[it & 1] = 1;
        lmax[w] = local_max;
    }
    if (VB) { double s = block_sum(local_sum, lds); if (threadIdx.x == 0) sum_partials_out[blockIdx.x] = s; }
    else __syncthreads();
    if (threadIdx.x == 0) {
        double m = lmax[0];
        for (int i = 1; i < kEmBlock / kWave; ++i) if (lmax[i] > m) m = lmax[i];
        blkmax[(it & 1) * kMaxPartials + blockIdx.x] = m;          // -1 = nothing gated in this block
        if (blockIdx.x == 0) st->it_a = it + 1;
    }
}

// truncateCountVector (:36-44) + alphaSum
__global__ void k_truncate(uint64_t M, const double* __restrict__ alpha, double cutoff, double* out, double* partials) {
    __shared__ double lds[kEmBlock / kWave];
    double v = 0.0;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock) {
        double a = alpha[t];
        if (a <= cutoff) a = 0.0;
        out[t] = a; v += a;
    }
    double s = block_sum(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// :885-891 mass = alpha / alphaSum
__global__ void k_mass(uint64_t M, const double* __restrict__ est, double* mass, const double* partials, int nb,
                       EmState* st) {
    __shared__ double lds[kEmBlock / kWave];
    double asum = sum_partials(partials, nb, lds);
    if (blockIdx.x == 0 && threadIdx.x == 0) st->alpha_sum = asum;
    if (!mass) return;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock)
        mass[t] = est[t] / asum;
}

static inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + kEmBlock - 1) / kEmBlock); }

}  // namespace sfgpu

using namespace sfgpu;

struct sfgpu_em {
    hipStream_t user_stream = nullptr;
    hipStream_t stream = nullptr;          // own stream: graph capture is illegal on the null stream
    hipStream_t cur = nullptr;             // where work goes: `stream` inside optimize(), the caller's stream for the piecewise API
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_join = nullptr;
    sfgpu_problem prob{};
    uint64_t L = 0;
    int nb = 1;                            // blocks of the per-transcript kernels
    double *alpha = nullptr, *alpha_out = nullptr, *x = nullptr, *lenc = nullptr;
    double *partials = nullptr, *sum_partials = nullptr, *scratch = nullptr;
    uint32_t* counts32 = nullptr;
    uint32_t* tile_lo = nullptr; uint32_t* tile_c0 = nullptr; uint32_t* tile_span = nullptr; uint32_t n_tiles = 0;
    uint64_t* tile_off = nullptr; uint64_t P = 0;          // window slots over all tiles
    double* partial = nullptr;                              // [P] per-tile window sums of one sweep
    uint32_t* cov_ptr = nullptr; uint32_t* cov_pos = nullptr;   // transcript -> its window slots
    uint32_t* slots = nullptr;                                  // chunked labels (see k_sweep_chunk)
    uint64_t* tile_chunk0 = nullptr; uint32_t* long_cls = nullptr; uint32_t n_long = 0; uint64_t n_chunks = 0;
    double* blkmax = nullptr; double* h_blkmax = nullptr;   // [2][kMaxPartials]
    int ablate = 0;                                         // timing experiments only (SFGPU_EM_ABLATE)
    int sweep_variant = 1;                                  // 0 = lane-per-class/global atomics, 1 = LDS tiles
    EmState* d_state = nullptr;
    EmState* h_state = nullptr;            // pinned
    sfgpu_em_opts opts{};
    bool begun = false;
    hipGraphExec_t graph = nullptr;
    sfgpu_em_opts graph_opts{};
    uint32_t graph_iters = 0;
};

static void em_free(sfgpu_em* em) {
    if (!em) return;
    if (em->stream) (void)hipStreamSynchronize(em->stream);
    if (em->graph) (void)hipGraphExecDestroy(em->graph);
    void* bufs[] = {em->alpha, em->alpha_out, em->x, em->lenc, em->partials, em->sum_partials, em->scratch,
                    em->counts32, em->d_state, em->tile_lo, em->tile_c0, em->tile_span, em->tile_off, em->partial,
                    em->cov_ptr, em->cov_pos, em->slots, em->tile_chunk0, em->long_cls,
                    em->blkmax};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (em->h_state) (void)hipHostFree(em->h_state);
    if (em->h_blkmax) (void)hipHostFree(em->h_blkmax);
    if (em->ev_a) (void)hipEventDestroy(em->ev_a);
    if (em->ev_b) (void)hipEventDestroy(em->ev_b);
    if (em->ev_join) (void)hipEventDestroy(em->ev_join);
    if (em->stream) (void)hipStreamDestroy(em->stream);
    delete em;
}

static int em_fill_opts(sfgpu_em* em, const sfgpu_em_opts* o) {
    SF_REQUIRE(o, SFGPU_ERR_INVALID, "null sfgpu_em_opts");
    SF_REQUIRE(o->tol >= 0.0, SFGPU_ERR_INVALID, "tol must be >= 0");
    em->opts = *o;
    if (em->opts.iters_per_launch == 0) em->opts.iters_per_launch = 32;
    return SFGPU_OK;
}

// enqueue one full iteration on the handle's stream
static int em_enqueue_sweep(sfgpu_em* em) {
    const sfgpu_problem& p = em->prob;
    if (p.C == 0) return SFGPU_OK;
    dim3 b(kEmBlock);
    const bool vb = em->opts.use_vbem != 0;
    if (em->sweep_variant == 0) {
        dim3 g(blocks_for(p.C));
        if (vb) hipLaunchKernelGGL(k_sweep_lane<true>, g, b, 0, em->cur, p.C, p.d_rowptr, p.d_ids, em->counts32, em->x,
                                   em->alpha_out, em->d_state, em->opts.min_iter, em->opts.max_iter);
        else hipLaunchKernelGGL(k_sweep_lane<false>, g, b, 0, em->cur, p.C, p.d_rowptr, p.d_ids, em->counts32, em->x,
                                em->alpha_out, em->d_state, em->opts.min_iter, em->opts.max_iter);
    } else {
        dim3 g(em->n_tiles + em->n_long);
        b = dim3(kSweepBlock);
        SweepArgs a{em->n_tiles, p.d_rowptr, p.d_ids, em->counts32, em->slots,
                    em->tile_chunk0, em->tile_lo, em->tile_span, em->tile_off, em->long_cls, em->x, em->alpha_out,
                    em->partial, em->d_state, em->opts.min_iter, em->opts.max_iter, em->ablate};
        if (vb) hipLaunchKernelGGL(k_sweep_chunk<true>, g, b, 0, em->cur, a);
        else hipLaunchKernelGGL(k_sweep_chunk<false>, g, b, 0, em->cur, a);
    }
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

// `fold`: the sweep's per-tile window sums still have to be folded into alphaOut (true inside
// optimize(); false in the piecewise API, where sfgpu_em_sweep folds before the caller's all-reduce)
static int em_enqueue_update(sfgpu_em* em, bool fold) {
    const sfgpu_problem& p = em->prob;
    dim3 g(em->nb), b(kEmBlock);
    fold = fold && em->sweep_variant != 0 && p.C != 0;
#define UPD_ARGS p.M, em->alpha, em->alpha_out, em->x, em->lenc, em->opts.tol, em->opts.check_mode, em->sum_partials, \
                 em->blkmax, em->d_state, em->cov_ptr, em->cov_pos, em->partial
    if (em->opts.use_vbem) {
        if (fold) hipLaunchKernelGGL((k_update<true, true>), g, b, 0, em->cur, UPD_ARGS);
        else hipLaunchKernelGGL((k_update<true, false>), g, b, 0, em->cur, UPD_ARGS);
        SF_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_vb_prepare, g, b, 0, em->cur, p.M, em->alpha, em->x, em->lenc, em->sum_partials, em->nb,
                           em->d_state, 0);
    } else {
        if (fold) hipLaunchKernelGGL((k_update<false, true>), g, b, 0, em->cur, UPD_ARGS);
        else hipLaunchKernelGGL((k_update<false, false>), g, b, 0, em->cur, UPD_ARGS);
    }
#undef UPD_ARGS
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

static int em_enqueue_fold(sfgpu_em* em) {
    const sfgpu_problem& p = em->prob;
    if (em->sweep_variant == 0 || p.C == 0) return SFGPU_OK;
    hipLaunchKernelGGL(k_fold, dim3(blocks_for(p.M)), dim3(kEmBlock), 0, em->cur, p.M, em->alpha_out, em->cov_ptr,
                       em->cov_pos, em->partial, em->d_state);
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

// order the handle's stream after everything already enqueued on the caller's stream
static int em_join_user(sfgpu_em* em) {
    SF_HIP(hipEventRecord(em->ev_join, em->user_stream));
    SF_HIP(hipStreamWaitEvent(em->stream, em->ev_join, 0));
    return SFGPU_OK;
}

static void em_stats_from_state(sfgpu_em* em, sfgpu_em_stats* s) {
    if (!s) return;
    const EmState* h = em->h_state;
    uint32_t it = h->it_a;
    s->iters = it;
    s->n_active = h->n_active;
    s->alpha_sum = h->alpha_sum;
    if (it == 0) { s->converged = 0; s->max_rel_diff = -DBL_MAX; return; }
    uint32_t par = (it - 1) & 1;
    s->converged = h->notconv[par] == 0;
    double m = -1.0;
    for (int i = 0; i < em->nb; ++i) { double v = em->h_blkmax[par * kMaxPartials + i]; if (v > m) m = v; }
    s->max_rel_diff = (m >= 0.0) ? m : -DBL_MAX;      // the reference starts from -DBL_MAX (:850)
}

extern "C" {

int sfgpu_em_create(sfgpu_em** out, const sfgpu_problem* prob, sfgpu_stream stream) {
    SF_REQUIRE(out && prob, SFGPU_ERR_INVALID, "sfgpu_em_create: null pointer");
    SF_REQUIRE(prob->M > 0 && prob->d_len, SFGPU_ERR_INVALID, "sfgpu_em_create: need M > 0 and d_len");
    SF_REQUIRE(prob->M <= (1ull << 31), SFGPU_ERR_RANGE, "sfgpu_em_create: transcript ids must fit 31 bits");
    SF_REQUIRE(prob->C == 0 || (prob->d_rowptr && prob->d_ids && prob->d_counts), SFGPU_ERR_INVALID,
               "sfgpu_em_create: null CSR pointer");
    sfgpu_em* em = new sfgpu_em();
    em->prob = *prob;
    em->user_stream = as_stream(stream);
    const uint64_t M = prob->M, C = prob->C;
    int nb = (int)((M + kEmBlock - 1) / kEmBlock);
    em->nb = nb < 1 ? 1 : (nb > kMaxPartials ? kMaxPartials : nb);
#define EM_TRY(expr)                                                                                 \
    do { hipError_t _e = (expr); if (_e != hipSuccess) {                                              \
        set_error("%s failed: %s", #expr, hipGetErrorString(_e)); em_free(em); return SFGPU_ERR_HIP; } } while (0)
    EM_TRY(hipStreamCreateWithFlags(&em->stream, hipStreamNonBlocking));
    em->cur = em->stream;
    EM_TRY(hipEventCreate(&em->ev_a)); EM_TRY(hipEventCreate(&em->ev_b));
    EM_TRY(hipEventCreateWithFlags(&em->ev_join, hipEventDisableTiming));
    EM_TRY(hipMalloc(&em->alpha, M * 8)); EM_TRY(hipMalloc(&em->alpha_out, M * 8));
    EM_TRY(hipMalloc(&em->x, M * 8)); EM_TRY(hipMalloc(&em->lenc, M * 8)); EM_TRY(hipMalloc(&em->scratch, M * 8));
    EM_TRY(hipMalloc(&em->partials, kMaxPartials * 8)); EM_TRY(hipMalloc(&em->sum_partials, kMaxPartials * 8));
    EM_TRY(hipMalloc(&em->counts32, (C ? C : 1) * 4));
    EM_TRY(hipMalloc(&em->d_state, sizeof(EmState)));
    EM_TRY(hipMalloc(&em->blkmax, 2 * kMaxPartials * 8));
    EM_TRY(hipHostMalloc(&em->h_blkmax, 2 * kMaxPartials * 8, hipHostMallocDefault));
    EM_TRY(hipMalloc(&em->tile_lo, 4));   // sized once nnz is known (below)
    if (const char* v = getenv("SFGPU_EM_SWEEP")) em->sweep_variant = atoi(v);
    if (const char* v = getenv("SFGPU_EM_ABLATE")) em->ablate = atoi(v);
    EM_TRY(hipHostMalloc(&em->h_state, sizeof(EmState), hipHostMallocDefault));
    EM_TRY(hipMemsetAsync(em->d_state, 0, sizeof(EmState), em->cur));
    int rc = em_join_user(em);
    if (rc) { em_free(em); return rc; }
    hipLaunchKernelGGL(k_clamp_len, dim3(blocks_for(M)), dim3(kEmBlock), 0, em->cur, M, prob->d_len, em->lenc);
    uint32_t rp_end = 0;
    if (C) {
        unsigned int* ovf = reinterpret_cast<unsigned int*>(em->partials);
        EM_TRY(hipMemsetAsync(ovf, 0, 4, em->cur));
        hipLaunchKernelGGL(k_narrow_counts, dim3(blocks_for(C)), dim3(kEmBlock), 0, em->cur, C, prob->d_counts,
                           em->counts32, ovf);
        unsigned int h_ovf = 0;
        EM_TRY(hipMemcpyAsync(&h_ovf, ovf, 4, hipMemcpyDeviceToHost, em->cur));
        EM_TRY(hipMemcpyAsync(&rp_end, prob->d_rowptr + C, 4, hipMemcpyDeviceToHost, em->cur));
        EM_TRY(hipStreamSynchronize(em->cur));
        if (h_ovf) { set_error("sfgpu_em_create: a class count >= 2^31 - 1"); em_free(em); return SFGPU_ERR_RANGE; }
    } else {
        EM_TRY(hipStreamSynchronize(em->cur));
    }
    em->L = rp_end;
    if (C) {   // nnz-balanced tile plan of the sweep + the cover lists of the fold
        em->n_tiles = (uint32_t)(((uint64_t)rp_end + kTileNnz - 1) / kTileNnz);
        if (em->n_tiles == 0) em->n_tiles = 1;
        const uint32_t nt = em->n_tiles;
        (void)hipFree(em->tile_lo); em->tile_lo = nullptr;
        EM_TRY(hipMalloc(&em->tile_lo, (size_t)nt * 4));
        EM_TRY(hipMalloc(&em->tile_span, ((size_t)nt + 1) * 4));
        EM_TRY(hipMalloc(&em->tile_c0, ((size_t)nt + 1) * 4));
        EM_TRY(hipMalloc(&em->tile_off, ((size_t)nt + 1) * 8));
        EM_TRY(hipMalloc(&em->cov_ptr, ((size_t)M + 1) * 4));
        hipLaunchKernelGGL(k_tile_plan, dim3((nt + 1 + kEmBlock - 1) / kEmBlock), dim3(kEmBlock), 0, em->cur, C, nt,
                           prob->d_rowptr, em->tile_c0);
        hipLaunchKernelGGL(k_tile_window, dim3(nt), dim3(kEmBlock), 0, em->cur, prob->d_rowptr, prob->d_ids, em->tile_c0,
                           em->tile_lo, em->tile_span);
        EM_TRY(hipGetLastError());
        if (exclusive_scan_u32(em->tile_span, em->tile_off, nt, em->cur)) { em_free(em); return SFGPU_ERR_HIP; }
        {   // chunk packing of the labels (see k_sweep_chunk)
            uint32_t *slotpos = nullptr, *t_chunks = nullptr; unsigned int* d_nlong = nullptr;
            EM_TRY(hipMalloc(&slotpos, C * 4)); EM_TRY(hipMalloc(&t_chunks, ((size_t)nt + 1) * 4));
            EM_TRY(hipMalloc(&em->tile_chunk0, ((size_t)nt + 1) * 8));
            EM_TRY(hipMalloc(&em->long_cls, C * 4)); EM_TRY(hipMalloc(&d_nlong, 4));
            EM_TRY(hipMemsetAsync(d_nlong, 0, 4, em->cur));
            hipLaunchKernelGGL(k_tile_pack, dim3(nt), dim3(kEmBlock), 0, em->cur, prob->d_rowptr, em->tile_c0, slotpos, t_chunks,
                               em->long_cls, d_nlong);
            EM_TRY(hipGetLastError());
            int sr = exclusive_scan_u32(t_chunks, em->tile_chunk0, nt, em->cur);
            uint64_t nq = 0; unsigned int nl = 0;
            if (!sr) {
                EM_TRY(hipMemcpyAsync(&nq, em->tile_chunk0 + nt, 8, hipMemcpyDeviceToHost, em->cur));
                EM_TRY(hipMemcpyAsync(&nl, d_nlong, 4, hipMemcpyDeviceToHost, em->cur));
                EM_TRY(hipStreamSynchronize(em->cur));
                em->n_chunks = nq; em->n_long = nl;
                EM_TRY(hipMalloc(&em->slots, (nq ? nq : 1) * kChunk * 4));
                EM_TRY(hipMemsetAsync(em->slots, 0xFF, (nq ? nq : 1) * kChunk * 4, em->cur));
                hipLaunchKernelGGL(k_fill_slots, dim3(blocks_for(C)), dim3(kEmBlock), 0, em->cur, C, prob->d_rowptr, prob->d_ids,
                                   em->counts32, slotpos, em->tile_chunk0, em->slots);
                EM_TRY(hipGetLastError());
                EM_TRY(hipStreamSynchronize(em->cur));
            }
            (void)hipFree(slotpos); (void)hipFree(t_chunks); (void)hipFree(d_nlong);
            if (sr) { em_free(em); return sr; }
        }
        uint64_t P = 0;
        EM_TRY(hipMemcpyAsync(&P, em->tile_off + nt, 8, hipMemcpyDeviceToHost, em->cur));
        EM_TRY(hipStreamSynchronize(em->cur));
        if (P >= (1ull << 32)) { set_error("sfgpu_em_create: window slots exceed 2^32"); em_free(em); return SFGPU_ERR_RANGE; }
        em->P = P;
        EM_TRY(hipMalloc(&em->partial, (P ? P : 1) * 8));
        EM_TRY(hipMalloc(&em->cov_pos, (P ? P : 1) * 4));
        EM_TRY(hipMemsetAsync(em->partial, 0, (P ? P : 1) * 8, em->cur));
        if (P) {
            uint64_t *k_in = nullptr, *k_out = nullptr; uint32_t* v_in = nullptr;
            EM_TRY(hipMalloc(&k_in, P * 8)); EM_TRY(hipMalloc(&k_out, P * 8)); EM_TRY(hipMalloc(&v_in, P * 4));
            hipLaunchKernelGGL(k_cover_pairs, dim3(nt), dim3(kEmBlock), 0, em->cur, em->tile_lo, em->tile_span, em->tile_off,
                               k_in, v_in);
            int bits = 1; while (bits < 32 && (1ull << bits) <= M) ++bits;
            int src = sort_pairs_u64_u32(k_in, k_out, v_in, em->cov_pos, P, em->cur, bits);
            if (!src) {
                hipLaunchKernelGGL(k_cover_ptr, dim3(blocks_for(M + 1)), dim3(kEmBlock), 0, em->cur, M, P, k_out, em->cov_ptr);
                (void)hipStreamSynchronize(em->cur);
            }
            (void)hipFree(k_in); (void)hipFree(k_out); (void)hipFree(v_in);
            if (src) { em_free(em); return src; }
        } else {
            EM_TRY(hipMemsetAsync(em->cov_ptr, 0, ((size_t)M + 1) * 4, em->cur));
        }
        EM_TRY(hipGetLastError());
    }
#undef EM_TRY
    *out = em;
    return SFGPU_OK;
}

int sfgpu_em_destroy(sfgpu_em* em) { em_free(em); return SFGPU_OK; }

double* sfgpu_em_alpha_out(sfgpu_em* em) { return em ? em->alpha_out : nullptr; }

static int em_begin_on(sfgpu_em* em, const sfgpu_em_opts* opts, hipStream_t work) {
    SF_REQUIRE(em, SFGPU_ERR_INVALID, "sfgpu_em_begin: null handle");
    int rc = em_fill_opts(em, opts);
    if (rc) return rc;
    em->cur = work;
    SF_HIP(hipMemsetAsync(em->alpha_out, 0, em->prob.M * 8, em->cur));
    if (em->L) {
        hipLaunchKernelGGL(k_mark_active, dim3(blocks_for(em->L)), dim3(kEmBlock), 0, em->cur, em->L, em->prob.d_ids,
                           em->alpha_out);
        SF_CHECK_LAUNCH();
    }
    em->begun = true;
    return SFGPU_OK;
}

int sfgpu_em_begin(sfgpu_em* em, const sfgpu_em_opts* opts) {
    SF_REQUIRE(em, SFGPU_ERR_INVALID, "sfgpu_em_begin: null handle");
    (void)hipStreamSynchronize(em->stream);    // nothing of a previous optimize() may be in flight
    return em_begin_on(em, opts, em->user_stream);
}

static int sfgpu_em_init_impl(sfgpu_em* em) {
    SF_REQUIRE(em && em->begun, SFGPU_ERR_STATE, "sfgpu_em_init: call begin first");
    const sfgpu_problem& p = em->prob;
    dim3 g(em->nb), b(kEmBlock);
    hipLaunchKernelGGL(k_count_active, g, b, 0, em->cur, p.M, em->alpha_out, em->partials);
    SF_CHECK_LAUNCH();
    double total = (double)p.num_mapped;   // :792
    if (em->opts.use_vbem) {
        hipLaunchKernelGGL(k_init_alpha<true>, g, b, 0, em->cur, p.M, em->alpha, em->alpha_out, em->x, em->lenc, total,
                           em->partials, em->sum_partials, em->nb, em->d_state);
        SF_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_vb_prepare, g, b, 0, em->cur, p.M, em->alpha, em->x, em->lenc, em->sum_partials, em->nb,
                           em->d_state, 1);
    } else {
        hipLaunchKernelGGL(k_init_alpha<false>, g, b, 0, em->cur, p.M, em->alpha, em->alpha_out, em->x, em->lenc, total,
                           em->partials, em->sum_partials, em->nb, em->d_state);
    }
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

int sfgpu_em_init(sfgpu_em* em) { return sfgpu_em_init_impl(em); }

int sfgpu_em_sweep(sfgpu_em* em) {
    SF_REQUIRE(em && em->begun, SFGPU_ERR_STATE, "sfgpu_em_sweep: call begin/init first");
    int rc = em_enqueue_sweep(em);
    return rc ? rc : em_enqueue_fold(em);
}

int sfgpu_em_update(sfgpu_em* em) {
    SF_REQUIRE(em && em->begun, SFGPU_ERR_STATE, "sfgpu_em_update: call begin/init first");
    return em_enqueue_update(em, false);
}

int sfgpu_em_poll(sfgpu_em* em, int* done, sfgpu_em_stats* stats) {
    SF_REQUIRE(em, SFGPU_ERR_INVALID, "sfgpu_em_poll: null handle");
    SF_HIP(hipMemcpyAsync(em->h_state, em->d_state, sizeof(EmState), hipMemcpyDeviceToHost, em->cur));
    SF_HIP(hipMemcpyAsync(em->h_blkmax, em->blkmax, 2 * kMaxPartials * 8, hipMemcpyDeviceToHost, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    const EmState* h = em->h_state;
    uint32_t it = h->it_a;
    bool stop = false;
    if (it >= em->opts.min_iter) stop = (it >= em->opts.max_iter) || (it > 0 && h->notconv[(it - 1) & 1] == 0);
    if (done) *done = stop ? 1 : 0;
    em_stats_from_state(em, stats);
    return SFGPU_OK;
}

int sfgpu_em_finish(sfgpu_em* em, double* d_alpha_out, double* d_mass_out, sfgpu_em_stats* stats) {
    SF_REQUIRE(em && d_alpha_out, SFGPU_ERR_INVALID, "sfgpu_em_finish: null pointer");
    const sfgpu_problem& p = em->prob;
    double cutoff = em->opts.use_vbem ? (kPriorAlpha + kMinAlpha) : kMinAlpha;   // :812
    dim3 g(em->nb), b(kEmBlock);
    hipLaunchKernelGGL(k_truncate, g, b, 0, em->cur, p.M, em->alpha, cutoff, d_alpha_out, em->partials);
    SF_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_mass, g, b, 0, em->cur, p.M, d_alpha_out, d_mass_out, em->partials, em->nb, em->d_state);
    SF_CHECK_LAUNCH();
    SF_HIP(hipMemcpyAsync(em->h_state, em->d_state, sizeof(EmState), hipMemcpyDeviceToHost, em->cur));
    SF_HIP(hipMemcpyAsync(em->h_blkmax, em->blkmax, 2 * kMaxPartials * 8, hipMemcpyDeviceToHost, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    em_stats_from_state(em, stats);
    if (em->h_state->alpha_sum < kTiny) {                                       // :877-881
        set_error("Total alpha weight was too small! Make sure you ran sailfish correctly.");
        return SFGPU_ERR_ALPHA_SUM;
    }
    return SFGPU_OK;
}

static bool same_opts(const sfgpu_em_opts& a, const sfgpu_em_opts& b) {
    return a.use_vbem == b.use_vbem && a.tol == b.tol && a.min_iter == b.min_iter && a.max_iter == b.max_iter &&
           a.check_mode == b.check_mode && a.iters_per_launch == b.iters_per_launch;
}

// capture `n` iterations into an executable graph (kernel arguments are baked, the iteration
// index and the stop latch live in device memory)
static int em_build_graph(sfgpu_em* em, uint32_t n) {
    if (em->graph && same_opts(em->graph_opts, em->opts) && em->graph_iters == n) return SFGPU_OK;
    if (em->graph) { (void)hipGraphExecDestroy(em->graph); em->graph = nullptr; }
    hipGraph_t g = nullptr;
    SF_HIP(hipStreamBeginCapture(em->cur, hipStreamCaptureModeThreadLocal));
    int rc = SFGPU_OK;
    for (uint32_t i = 0; i < n && rc == SFGPU_OK; ++i) {
        rc = em_enqueue_sweep(em);
        if (rc == SFGPU_OK) rc = em_enqueue_update(em, true);
    }
    hipError_t e = hipStreamEndCapture(em->cur, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    SF_HIP(e);
    hipError_t ei = hipGraphInstantiate(&em->graph, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    SF_HIP(ei);
    em->graph_opts = em->opts; em->graph_iters = n;
    return SFGPU_OK;
}

int sfgpu_em_optimize(sfgpu_em* em, const sfgpu_em_opts* opts, double* d_alpha_out, double* d_mass_out,
                      sfgpu_em_stats* stats) {
    SF_REQUIRE(em && d_alpha_out, SFGPU_ERR_INVALID, "sfgpu_em_optimize: null pointer");
    int rc;
    if ((rc = em_join_user(em))) return rc;
    if ((rc = em_begin_on(em, opts, em->stream))) return rc;
    if ((rc = sfgpu_em_init_impl(em))) return rc;
    int done = 0;
    sfgpu_em_stats st{};
    if ((rc = sfgpu_em_poll(em, &done, &st))) return rc;
    if (st.n_active == 0) {                                                      // :794-798
        set_error("It seems that no transcripts are expressed; something is likely wrong!");
        if (stats) *stats = st;
        return SFGPU_ERR_NO_ACTIVE;
    }
    log_msg(0, "Optimizing over %llu equivalence classes", (unsigned long long)em->prob.C);   // :790
    const bool use_graph = getenv("SFGPU_EM_NOGRAPH") == nullptr;
    const uint32_t chunk = em->opts.iters_per_launch;
    if (use_graph && (rc = em_build_graph(em, chunk))) return rc;
    SF_HIP(hipEventRecord(em->ev_a, em->cur));
    while (!done) {
        if (use_graph) {
            SF_HIP(hipGraphLaunch(em->graph, em->cur));
        } else {
            for (uint32_t i = 0; i < chunk; ++i) {
                if ((rc = em_enqueue_sweep(em))) return rc;
                if ((rc = em_enqueue_update(em, true))) return rc;
            }
        }
        if ((rc = sfgpu_em_poll(em, &done, &st))) return rc;
    }
    SF_HIP(hipEventRecord(em->ev_b, em->cur));
    rc = sfgpu_em_finish(em, d_alpha_out, d_mass_out, &st);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, em->ev_a, em->ev_b);
    st.loop_ms = ms;
    log_msg(0, "iteration = %u | max rel diff. = %g", st.iters, st.max_rel_diff);   // :871-872
    if (stats) *stats = st;
    return rc;
}

int sfgpu_em_time_sweep(sfgpu_em* em, const sfgpu_em_opts* opts, uint32_t n, double* avg_ms) {
    SF_REQUIRE(em && avg_ms && n > 0, SFGPU_ERR_INVALID, "sfgpu_em_time_sweep: bad argument");
    int rc;
    if (!em->begun) {
        if ((rc = sfgpu_em_begin(em, opts))) return rc;
        if ((rc = sfgpu_em_init(em))) return rc;
    } else if ((rc = em_fill_opts(em, opts))) return rc;
    // keep alphaOut and the state block intact: time on copies
    EmState saved;
    SF_HIP(hipMemcpyAsync(em->scratch, em->alpha_out, em->prob.M * 8, hipMemcpyDeviceToDevice, em->cur));
    SF_HIP(hipMemcpyAsync(&saved, em->d_state, sizeof(EmState), hipMemcpyDeviceToHost, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    EmState run = saved; run.it_a = 0;
    uint32_t keep_min = em->opts.min_iter, keep_max = em->opts.max_iter;
    em->opts.min_iter = 1; em->opts.max_iter = 2;   // it_a = 0 never satisfies the stop test
    SF_HIP(hipMemcpyAsync(em->d_state, &run, sizeof(EmState), hipMemcpyHostToDevice, em->cur));
    for (int w = 0; w < 3; ++w) if ((rc = em_enqueue_sweep(em))) return rc;
    SF_HIP(hipEventRecord(em->ev_a, em->cur));
    for (uint32_t i = 0; i < n; ++i) if ((rc = em_enqueue_sweep(em))) return rc;
    SF_HIP(hipEventRecord(em->ev_b, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    float ms = 0.f;
    SF_HIP(hipEventElapsedTime(&ms, em->ev_a, em->ev_b));
    *avg_ms = (double)ms / n;
    em->opts.min_iter = keep_min; em->opts.max_iter = keep_max;
    SF_HIP(hipMemcpyAsync(em->d_state, &saved, sizeof(EmState), hipMemcpyHostToDevice, em->cur));
    SF_HIP(hipMemcpyAsync(em->alpha_out, em->scratch, em->prob.M * 8, hipMemcpyDeviceToDevice, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    return SFGPU_OK;
}

}  // extern "C"
