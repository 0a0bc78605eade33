// merge.hip -- the device side of the multi-GPU class-table exchange, behind the C ABI.
//
// The reference has no multi-GPU path; what is partitioned here is what its EquivalenceClassBuilder holds after
// finish() (include/EquivalenceClassBuilder.hpp:64-80): label -> count.  One process (or thread) per GPU builds the
// table of ITS reads; the tables are then merged so that every rank ends with the table a single builder would have
// produced from all reads (integer work: bit-exact).  The library does the device work, the HOST owns the transport
// (RCCL / MPI / anything that moves device buffers) -- sailfish_amd/distributed.py with torch.distributed, a C++ host
// with librccl, tests/cpp_host_test.cpp with plain device copies:
//
//   owner(class) = f(XXH64 of its label) mod N                                  every rank agrees on it
//   sfgpu_eqvec_pack_by_owner : my table -> N blocks [counts u64 | lens u32 | ids u32 | pad], one per owner
//        (all-to-all of the blocks: block d goes to rank d)
//   sfgpu_eq_add_block_device : the owner upserts what it received (equal labels add their counts)     -> its partition
//   sfgpu_eqvec_export_block  : the partition as one block [counts u64 | hashes u64 | lens u32 | ids u32]
//        (all-gather of the partitions: they are DISJOINT)
//   sfgpu_eqvec_merge_disjoint: the union in the canonical order (first id, XXH64, length, label) -- a sort of
//        (first id, hash) keys and a gather; nothing is hashed again.
//
// Round 1 assembled these steps from generic torch kernels (argsort / cumsum / repeat_interleave / fancy indexing,
// sailfish_amd/distributed.py); they are hand-written here: one stable 8-bit radix pass + one gather kernel per step.
#include <vector>

#include "common.h"
#include "primitives.h"

namespace sfgpu {

constexpr int kMergeBlock = 256;
constexpr uint32_t kMaxOwners = 256;

__host__ __device__ __forceinline__ uint32_t owner_of(uint64_t hash, uint32_t n) {
    return (uint32_t)(((hash >> 33) & 0x3FFFFFFFull) % n);          // any function of the label every rank agrees on
}

static inline unsigned mgrid(uint64_t n) { return (unsigned)((n + kMergeBlock - 1) / kMergeBlock); }

// per-owner class and id counts (block-private LDS histograms, one global atomic per owner and block)
__global__ void __launch_bounds__(kMergeBlock)
k_owner_sizes(uint64_t C, const uint32_t* __restrict__ rowptr, const uint64_t* __restrict__ hashes, uint32_t n_owners,
              unsigned long long* sizes /* [2 * n_owners]: classes, ids */) {
    __shared__ unsigned long long h[2 * kMaxOwners];
    for (uint32_t i = threadIdx.x; i < 2 * n_owners; i += kMergeBlock) h[i] = 0;
    __syncthreads();
    for (uint64_t c = (uint64_t)blockIdx.x * kMergeBlock + threadIdx.x; c < C; c += (uint64_t)gridDim.x * kMergeBlock) {
        const uint32_t d = owner_of(hashes[c], n_owners);
        atomicAdd(&h[2 * d], 1ull);
        atomicAdd(&h[2 * d + 1], (unsigned long long)(rowptr[c + 1] - rowptr[c]));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2 * n_owners; i += kMergeBlock) if (h[i]) atomicAdd(&sizes[i], h[i]);
}

__global__ void k_owner_keys(uint64_t C, const uint64_t* __restrict__ hashes, uint32_t n_owners, uint64_t* keys, uint32_t* vals) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { keys[c] = owner_of(hashes[c], n_owners); vals[c] = (uint32_t)c; }
}

// lens[j] = length of the j-th class in owner order (+ a zero sentinel for the scan)
__global__ void k_perm_lens(uint64_t C, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ rowptr, uint32_t* lens) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < C) { const uint32_t c = perm[j]; lens[j] = rowptr[c + 1] - rowptr[c]; }
    else if (j == C) lens[j] = 0;
}

struct PackTables { uint64_t cls0[kMaxOwners + 1]; uint64_t ids0[kMaxOwners + 1]; uint64_t blk0[kMaxOwners + 1]; };

// class j (owner order) -> its owner's block: counts[j - cls0[d]], lens[...], ids at ids_off[j] - ids0[d]
__global__ void __launch_bounds__(kMergeBlock)
k_pack(uint64_t C, uint32_t n_owners, PackTables t, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ ids_off,
       const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids, const uint64_t* __restrict__ counts,
       unsigned char* __restrict__ blocks) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= C) return;
    uint32_t d = 0;                                               // owner of position j: last d with cls0[d] <= j
    { uint32_t lo = 0, hi = n_owners; while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (t.cls0[mid] <= j) lo = mid; else hi = mid; } d = lo; }
    const uint64_t cd = t.cls0[d + 1] - t.cls0[d];
    unsigned char* b = blocks + t.blk0[d];
    const uint64_t k = j - t.cls0[d];
    const uint32_t c = perm[j];
    const uint32_t len = rowptr[c + 1] - rowptr[c];
    reinterpret_cast<uint64_t*>(b)[k] = counts[c];
    reinterpret_cast<uint32_t*>(b + 8 * cd)[k] = len;
    uint32_t* dst = reinterpret_cast<uint32_t*>(b + 12 * cd) + (ids_off[j] - t.ids0[d]);
    const uint32_t* src = ids + rowptr[c];
    for (uint32_t q = 0; q < len; ++q) dst[q] = src[q];
}

// offsets of a block's labels from its lens (uint32 offsets for the weighted upsert)
__global__ void k_block_lens(uint64_t n, const uint32_t* __restrict__ lens_in, uint32_t* lens) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) lens[j] = lens_in[j]; else if (j == n) lens[j] = 0;
}
__global__ void k_narrow_u64(uint64_t n, const uint64_t* __restrict__ in, uint32_t* out) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = (uint32_t)in[j];
}

// the finished table as ONE block [counts u64[C] | hashes u64[C] | lens u32[C] | ids u32[L]]
__global__ void k_export_block(uint64_t C, uint64_t L, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
                               const uint64_t* __restrict__ counts, const uint64_t* __restrict__ hashes, unsigned char* __restrict__ b) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) {
        reinterpret_cast<uint64_t*>(b)[i] = counts[i];
        reinterpret_cast<uint64_t*>(b + 8 * C)[i] = hashes[i];
        reinterpret_cast<uint32_t*>(b + 16 * C)[i] = rowptr[i + 1] - rowptr[i];
    }
    if (i < L) reinterpret_cast<uint32_t*>(b + 20 * C)[i] = ids[i];
}

struct PartTables { const unsigned char* blk[64]; uint64_t cls0[65]; uint64_t n_cls[64]; };

// global class g of the union -> (part, index in part); keys for the canonical order
__device__ __forceinline__ uint32_t part_of(const PartTables& t, uint32_t n_parts, uint64_t g) {
    uint32_t lo = 0, hi = n_parts;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (t.cls0[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}
// src_off[g] = id offset of class g inside its part (prefix of the part's lens), computed per part with a scan before
__global__ void __launch_bounds__(kMergeBlock)
k_union_keys(uint64_t n, uint32_t n_parts, PartTables t, const uint64_t* __restrict__ src_off, uint64_t* keys, uint32_t* vals) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t p = part_of(t, n_parts, g);
    const uint64_t k = g - t.cls0[p], C = t.n_cls[p];
    const unsigned char* b = t.blk[p];
    const uint64_t hash = reinterpret_cast<const uint64_t*>(b + 8 * C)[k];
    const uint32_t first = reinterpret_cast<const uint32_t*>(b + 20 * C)[src_off[g]];       // every class has at least one id
    keys[g] = ((uint64_t)first << 32) | (hash >> 32);
    vals[g] = (uint32_t)g;
}
__global__ void __launch_bounds__(kMergeBlock)
k_union_lens(uint64_t n, uint32_t n_parts, PartTables t, uint32_t* lens) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n) return;
    if (g == n) { lens[g] = 0; return; }
    const uint32_t p = part_of(t, n_parts, g);
    lens[g] = reinterpret_cast<const uint32_t*>(t.blk[p] + 16 * t.n_cls[p])[g - t.cls0[p]];
}
__global__ void k_sorted_union_lens(uint64_t n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ lens, uint32_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = lens[order[i]]; else if (i == n) out[i] = 0;
}
// two classes with the same (first id, XXH64): the (length, label) keys of the canonical order would be needed
__global__ void __launch_bounds__(kMergeBlock)
k_union_ties(uint64_t n, uint32_t n_parts, PartTables t, const uint64_t* __restrict__ sorted_keys, const uint32_t* __restrict__ order,
             unsigned int* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n || sorted_keys[i] != sorted_keys[i + 1]) return;
    auto hash_of = [&](uint64_t g) { const uint32_t p = part_of(t, n_parts, g); return reinterpret_cast<const uint64_t*>(t.blk[p] + 8 * t.n_cls[p])[g - t.cls0[p]]; };
    if (hash_of(order[i]) == hash_of(order[i + 1])) atomicOr(flag, 1u);
    else atomicOr(flag, 2u);                                       // same high hash half only: order them by the full hash below
}
// rare: runs of equal (first id, hash >> 32) keys are put into full-hash order by the thread at the head of the run
__global__ void __launch_bounds__(kMergeBlock)
k_union_tie_fix(uint64_t n, uint32_t n_parts, PartTables t, const uint64_t* __restrict__ sorted_keys, uint32_t* order, unsigned int* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (i > 0 && sorted_keys[i - 1] == sorted_keys[i])) return;
    uint64_t e = i + 1;
    while (e < n && sorted_keys[e] == sorted_keys[i]) ++e;
    if (e == i + 1) return;
    auto hash_of = [&](uint64_t g) { const uint32_t p = part_of(t, n_parts, g); return reinterpret_cast<const uint64_t*>(t.blk[p] + 8 * t.n_cls[p])[g - t.cls0[p]]; };
    for (uint64_t a = i + 1; a < e; ++a) {
        const uint32_t v = order[a]; const uint64_t hv = hash_of(v); uint64_t b = a;
        while (b > i && hv < hash_of(order[b - 1])) { order[b] = order[b - 1]; --b; }
        order[b] = v;
    }
    // two members of a run of three or more may share their FULL hash without having been neighbours before the run was put
    // into order (k_union_ties only saw sorted-adjacent pairs): they are neighbours now
    for (uint64_t a = i + 1; a < e; ++a) if (hash_of(order[a]) == hash_of(order[a - 1])) atomicOr(flag, 1u);
}
__global__ void __launch_bounds__(kMergeBlock)
k_union_gather(uint64_t n, uint32_t n_parts, PartTables t, const uint32_t* __restrict__ order, const uint64_t* __restrict__ src_off,
               const uint64_t* __restrict__ dst_off, uint32_t* rowptr, uint32_t* ids, uint64_t* counts, uint64_t* hashes) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    rowptr[i] = (uint32_t)dst_off[i];
    if (i == n) return;
    const uint64_t g = order[i];
    const uint32_t p = part_of(t, n_parts, g);
    const uint64_t k = g - t.cls0[p], C = t.n_cls[p];
    const unsigned char* b = t.blk[p];
    counts[i] = reinterpret_cast<const uint64_t*>(b)[k];
    if (hashes) hashes[i] = reinterpret_cast<const uint64_t*>(b + 8 * C)[k];
    const uint32_t len = (uint32_t)(dst_off[i + 1] - dst_off[i]);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(b + 20 * C) + src_off[g];
    uint32_t* dst = ids + dst_off[i];
    for (uint32_t q = 0; q < len; ++q) dst[q] = src[q];
}
// id offsets of the classes inside their own part: an exclusive scan of the union's lens restarted at every part
__global__ void k_rebase_parts(uint64_t n, uint32_t n_parts, PartTables t, const uint64_t* __restrict__ scan, uint64_t* src_off) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t p = part_of(t, n_parts, g);
    src_off[g] = scan[g] - scan[t.cls0[p]];
}

}  // namespace sfgpu

using namespace sfgpu;

extern "C" {

int sfgpu_eqvec_owner_sizes(const uint32_t* d_rowptr, const uint64_t* d_hashes, uint64_t C, uint32_t n_owners,
                            uint64_t* h_classes, uint64_t* h_ids, sfgpu_stream stream) {
    SF_REQUIRE(n_owners >= 1 && n_owners <= kMaxOwners && h_classes && h_ids, SFGPU_ERR_INVALID, "sfgpu_eqvec_owner_sizes: 1 <= n_owners <= 256");
    SF_REQUIRE(C == 0 || (d_rowptr && d_hashes), SFGPU_ERR_INVALID, "sfgpu_eqvec_owner_sizes: null pointer");
    hipStream_t st = as_stream(stream);
    DevBuf<unsigned long long> sizes;
    int rc;
    if ((rc = sizes.reserve(2 * n_owners, st, false))) return rc;
    SF_HIP(hipMemsetAsync(sizes.p, 0, 2 * n_owners * sizeof(unsigned long long), st));
    if (C) {
        const unsigned g = mgrid(C) < 1024u ? mgrid(C) : 1024u;
        hipLaunchKernelGGL(k_owner_sizes, dim3(g), dim3(kMergeBlock), 0, st, C, d_rowptr, d_hashes, n_owners, sizes.p);
        SF_CHECK_LAUNCH();
    }
    std::vector<unsigned long long> h(2 * n_owners);
    SF_HIP(hipMemcpyAsync(h.data(), sizes.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    for (uint32_t d = 0; d < n_owners; ++d) { h_classes[d] = h[2 * d]; h_ids[d] = h[2 * d + 1]; }
    return SFGPU_OK;
}

int sfgpu_eqvec_pack_by_owner(const uint32_t* d_rowptr, const uint32_t* d_ids, const uint64_t* d_counts, const uint64_t* d_hashes,
                              uint64_t C, uint32_t n_owners, const uint64_t* h_classes, const uint64_t* h_ids, void* d_blocks,
                              uint64_t* h_block_off, sfgpu_stream stream) {
    SF_REQUIRE(n_owners >= 1 && n_owners <= kMaxOwners && h_classes && h_ids && h_block_off, SFGPU_ERR_INVALID,
               "sfgpu_eqvec_pack_by_owner: 1 <= n_owners <= 256 and the sizes of sfgpu_eqvec_owner_sizes");
    hipStream_t st = as_stream(stream);
    PackTables t;
    t.cls0[0] = t.ids0[0] = t.blk0[0] = 0;
    for (uint32_t d = 0; d < n_owners; ++d) {
        t.cls0[d + 1] = t.cls0[d] + h_classes[d]; t.ids0[d + 1] = t.ids0[d] + h_ids[d];
        t.blk0[d + 1] = t.blk0[d] + SFGPU_BLOCK_BYTES(h_classes[d], h_ids[d]);
    }
    for (uint32_t d = 0; d <= n_owners; ++d) h_block_off[d] = t.blk0[d];
    SF_REQUIRE(t.cls0[n_owners] == C, SFGPU_ERR_INVALID, "sfgpu_eqvec_pack_by_owner: sizes do not add up to the table");
    if (C == 0) return SFGPU_OK;
    SF_REQUIRE(d_rowptr && d_ids && d_counts && d_hashes && d_blocks, SFGPU_ERR_INVALID, "sfgpu_eqvec_pack_by_owner: null pointer");
    SF_REQUIRE(C < (1ull << 32), SFGPU_ERR_RANGE, "sfgpu_eqvec_pack_by_owner: more than 2^32 classes");
    DevBuf<uint64_t> keys_in, keys_out, ids_off; DevBuf<uint32_t> vals_in, perm, lens;
    int rc;
    if ((rc = keys_in.reserve(C, st, false)) || (rc = keys_out.reserve(C, st, false)) || (rc = vals_in.reserve(C, st, false)) ||
        (rc = perm.reserve(C, st, false)) || (rc = lens.reserve(C + 1, st, false)) || (rc = ids_off.reserve(C + 2, st, false))) return rc;
    hipLaunchKernelGGL(k_owner_keys, dim3(mgrid(C)), dim3(kMergeBlock), 0, st, C, d_hashes, n_owners, keys_in.p, vals_in.p);
    SF_CHECK_LAUNCH();
    // stable: classes keep their canonical order inside an owner's block (one 8-bit radix pass)
    if ((rc = sort_pairs_u64_u32(keys_in.p, keys_out.p, vals_in.p, perm.p, C, st, 8))) return rc;
    hipLaunchKernelGGL(k_perm_lens, dim3(mgrid(C + 1)), dim3(kMergeBlock), 0, st, C, perm.p, d_rowptr, lens.p);
    SF_CHECK_LAUNCH();
    if ((rc = exclusive_scan_u32(lens.p, ids_off.p, C, st))) return rc;
    // the blocks' padding bytes are part of what travels: keep them defined
    SF_HIP(hipMemsetAsync(d_blocks, 0, t.blk0[n_owners], st));
    hipLaunchKernelGGL(k_pack, dim3(mgrid(C)), dim3(kMergeBlock), 0, st, C, n_owners, t, perm.p, ids_off.p, d_rowptr, d_ids, d_counts,
                       reinterpret_cast<unsigned char*>(d_blocks));
    SF_CHECK_LAUNCH();
    SF_HIP(hipStreamSynchronize(st));          // the scratch buffers go out of scope
    return SFGPU_OK;
}

int sfgpu_eq_add_block_device(sfgpu_eq* eq, const void* d_block, uint64_t n_classes, uint64_t n_ids, sfgpu_stream stream) {
    SF_REQUIRE(eq, SFGPU_ERR_INVALID, "sfgpu_eq_add_block_device: null handle");
    if (n_classes == 0) return SFGPU_OK;
    SF_REQUIRE(d_block, SFGPU_ERR_INVALID, "sfgpu_eq_add_block_device: null block");
    SF_REQUIRE(n_classes < (1ull << 31) && n_ids < (1ull << 31), SFGPU_ERR_RANGE, "sfgpu_eq_add_block_device: a block holds < 2^31 classes and ids");
    hipStream_t st = as_stream(stream);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(d_block);
    const uint64_t* counts = reinterpret_cast<const uint64_t*>(b);
    const uint32_t* lens_in = reinterpret_cast<const uint32_t*>(b + 8 * n_classes);
    const uint32_t* ids = reinterpret_cast<const uint32_t*>(b + 12 * n_classes);
    DevBuf<uint32_t> lens, off32; DevBuf<uint64_t> off64;
    int rc;
    if ((rc = lens.reserve(n_classes + 1, st, false)) || (rc = off64.reserve(n_classes + 2, st, false)) || (rc = off32.reserve(n_classes + 1, st, false))) return rc;
    hipLaunchKernelGGL(k_block_lens, dim3(mgrid(n_classes + 1)), dim3(kMergeBlock), 0, st, n_classes, lens_in, lens.p);
    SF_CHECK_LAUNCH();
    if ((rc = exclusive_scan_u32(lens.p, off64.p, n_classes, st))) return rc;
    hipLaunchKernelGGL(k_narrow_u64, dim3(mgrid(n_classes + 1)), dim3(kMergeBlock), 0, st, n_classes + 1, off64.p, off32.p);
    SF_CHECK_LAUNCH();
    SF_HIP(hipStreamSynchronize(st));          // the builder works on its own stream
    rc = sfgpu_eq_add_weighted_device(eq, ids, off32.p, counts, (uint32_t)n_classes);
    return rc;
}

int sfgpu_eqvec_export_block(const uint32_t* d_rowptr, const uint32_t* d_ids, const uint64_t* d_counts, const uint64_t* d_hashes,
                             uint64_t C, uint64_t L, void* d_block, sfgpu_stream stream) {
    if (C == 0) return SFGPU_OK;
    SF_REQUIRE(d_rowptr && d_ids && d_counts && d_hashes && d_block, SFGPU_ERR_INVALID, "sfgpu_eqvec_export_block: null pointer");
    const uint64_t n = C > L ? C : L;
    hipLaunchKernelGGL(k_export_block, dim3(mgrid(n)), dim3(kMergeBlock), 0, as_stream(stream), C, L, d_rowptr, d_ids, d_counts, d_hashes,
                       reinterpret_cast<unsigned char*>(d_block));
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

int sfgpu_eqvec_merge_disjoint(const void* const* d_blocks, const uint64_t* n_classes, const uint64_t* n_ids, uint32_t n_parts,
                               uint32_t* d_rowptr, uint32_t* d_ids, uint64_t* d_counts, uint64_t* d_hashes, int* same_key_twice,
                               sfgpu_stream stream) {
    SF_REQUIRE(n_parts >= 1 && n_parts <= 64 && d_blocks && n_classes && n_ids && d_rowptr, SFGPU_ERR_INVALID,
               "sfgpu_eqvec_merge_disjoint: 1 <= n_parts <= 64");
    hipStream_t st = as_stream(stream);
    PartTables t;
    uint64_t n = 0, L = 0;
    t.cls0[0] = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        t.blk[p] = reinterpret_cast<const unsigned char*>(d_blocks[p]); t.n_cls[p] = n_classes[p];
        n += n_classes[p]; L += n_ids[p]; t.cls0[p + 1] = n;
        SF_REQUIRE(n_classes[p] == 0 || d_blocks[p], SFGPU_ERR_INVALID, "sfgpu_eqvec_merge_disjoint: null block");
    }
    if (same_key_twice) *same_key_twice = 0;
    SF_REQUIRE(n < (1ull << 32) && L < (1ull << 32), SFGPU_ERR_RANGE, "sfgpu_eqvec_merge_disjoint: the union must hold < 2^32 classes and ids");
    if (n == 0) { SF_HIP(hipMemsetAsync(d_rowptr, 0, 4, st)); SF_HIP(hipStreamSynchronize(st)); return SFGPU_OK; }
    SF_REQUIRE(d_ids && d_counts, SFGPU_ERR_INVALID, "sfgpu_eqvec_merge_disjoint: null output");
    DevBuf<uint32_t> lens, vals_in, order, lens_sorted; DevBuf<uint64_t> scan, src_off, keys_in, keys_out, dst_off; DevBuf<unsigned int> flag;
    int rc;
    if ((rc = lens.reserve(n + 1, st, false)) || (rc = vals_in.reserve(n, st, false)) || (rc = order.reserve(n, st, false)) ||
        (rc = lens_sorted.reserve(n + 1, st, false)) || (rc = scan.reserve(n + 2, st, false)) || (rc = src_off.reserve(n, st, false)) ||
        (rc = keys_in.reserve(n, st, false)) || (rc = keys_out.reserve(n, st, false)) || (rc = dst_off.reserve(n + 2, st, false)) ||
        (rc = flag.reserve(1, st, false))) return rc;
    hipLaunchKernelGGL(k_union_lens, dim3(mgrid(n + 1)), dim3(kMergeBlock), 0, st, n, n_parts, t, lens.p);
    SF_CHECK_LAUNCH();
    if ((rc = exclusive_scan_u32(lens.p, scan.p, n, st))) return rc;
    hipLaunchKernelGGL(k_rebase_parts, dim3(mgrid(n)), dim3(kMergeBlock), 0, st, n, n_parts, t, scan.p, src_off.p);
    hipLaunchKernelGGL(k_union_keys, dim3(mgrid(n)), dim3(kMergeBlock), 0, st, n, n_parts, t, src_off.p, keys_in.p, vals_in.p);
    SF_CHECK_LAUNCH();
    if ((rc = sort_pairs_u64_u32(keys_in.p, keys_out.p, vals_in.p, order.p, n, st))) return rc;
    SF_HIP(hipMemsetAsync(flag.p, 0, 4, st));
    hipLaunchKernelGGL(k_union_ties, dim3(mgrid(n)), dim3(kMergeBlock), 0, st, n, n_parts, t, keys_out.p, order.p, flag.p);
    SF_CHECK_LAUNCH();
    unsigned int h_flag = 0;
    SF_HIP(hipMemcpyAsync(&h_flag, flag.p, 4, hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    if (h_flag & 1u) {          // two different labels share (first id, XXH64): the caller folds the partitions through a builder instead
        if (same_key_twice) *same_key_twice = 1;
        return SFGPU_OK;
    }
    if (h_flag & 2u) {
        hipLaunchKernelGGL(k_union_tie_fix, dim3(mgrid(n)), dim3(kMergeBlock), 0, st, n, n_parts, t, keys_out.p, order.p, flag.p);
        SF_CHECK_LAUNCH();
        SF_HIP(hipMemcpyAsync(&h_flag, flag.p, 4, hipMemcpyDeviceToHost, st));   // (rare path: the extra round trip does not matter)
        SF_HIP(hipStreamSynchronize(st));
        if (h_flag & 1u) { if (same_key_twice) *same_key_twice = 1; return SFGPU_OK; }
    }
    hipLaunchKernelGGL(k_sorted_union_lens, dim3(mgrid(n + 1)), dim3(kMergeBlock), 0, st, n, order.p, lens.p, lens_sorted.p);
    SF_CHECK_LAUNCH();
    if ((rc = exclusive_scan_u32(lens_sorted.p, dst_off.p, n, st))) return rc;
    hipLaunchKernelGGL(k_union_gather, dim3(mgrid(n + 1)), dim3(kMergeBlock), 0, st, n, n_parts, t, order.p, src_off.p, dst_off.p, d_rowptr, d_ids,
                       d_counts, d_hashes);
    SF_CHECK_LAUNCH();
    SF_HIP(hipStreamSynchronize(st));
    return SFGPU_OK;
}

}  // extern "C"
