// eqclass_part.h -- radix-partitioned class construction (device kernels; included by eqclass.hip).
//
// The global table is an array of REGIONS of kRegionSlots consecutive slots; a label's home region is
// the high bits of its slot index and linear probing wraps inside the region.  One pass of a sub-batch:
//   1a k_part_hist    : hash every label, histogram label WORDS per region (LDS histogram per block)
//      (scan)
//   1b k_part_scatter : copy every label into its region's segment of a partition buffer -- first word
//                       flagged with bit 31, so a segment is a self-delimiting stream of labels
//   2  k_part_insert  : ONE block per region: the region's slots live in LDS (word u64 + count delta
//                       u32), the segment is streamed through LDS tiles with coalesced loads, labels
//                       are hashed, probed and counted with LDS atomics only; new classes get their
//                       class ids / arena space with one global atomic per block and are committed
//                       by the same block; finally the region is written back.
// HBM sees each label ~3 times, always streaming; there are no global atomics per read and no host
// round trip per sub-batch.  (The one-lane-per-read kernel k_insert probes the table in HBM: three
// random 64-byte sectors per read -- profiles/r1_pmc_summary.md.)
#pragma once

namespace sfgpu {

constexpr int kRegionBits = 12;
constexpr uint32_t kRegionSlots = 1u << kRegionBits;          // 4096 slots: 32 KB words + 16 KB counts in LDS
constexpr uint32_t kRegionLimit = kRegionSlots / 4 * 3;       // inserts beyond this occupancy are deferred
constexpr int kPartBlock = 1024;
constexpr int kTileWords = 4096;                              // 16 KB LDS tile of the label stream
constexpr int kMaxRegions = 4096;                             // LDS histogram size of passes 1a/1b
constexpr uint32_t kHeadBit = 0x80000000u;
constexpr uint32_t kMaxPartLabel = kTileWords / 2;            // longer labels take the generic path

__device__ __forceinline__ uint64_t region_next(uint64_t s) {
    return (s & ~(uint64_t)(kRegionSlots - 1)) | ((s + 1) & (kRegionSlots - 1));
}

// ---- pass 1a: hash every label once; remember its region; count label words per (block, region).
// Block b owns the reads [b*tile, (b+1)*tile) in both 1a and 1b, and writes its histogram row into a
// region-major matrix mat[region * n_blocks + b]; an exclusive scan of that matrix is then every
// (region, block) pair's output offset -- no global atomics, and the partition is stable across blocks.
__global__ void __launch_bounds__(kPartBlock)
k_part_hist(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ off, uint32_t first, uint32_t n,
            uint32_t tile, uint64_t mask, uint32_t n_regions, uint16_t* __restrict__ reg_of, uint32_t* __restrict__ mat,
            unsigned long long* n_long, uint32_t* long_list) {
    __shared__ unsigned int lh[kMaxRegions];
    for (uint32_t i = threadIdx.x; i < n_regions; i += kPartBlock) lh[i] = 0;
    __syncthreads();
    const uint64_t t0 = (uint64_t)blockIdx.x * tile;
    const uint64_t t1 = (t0 + tile < n) ? t0 + tile : n;
    // four reads per lane per step: their offset and label loads are issued together, so a step costs
    // two memory round trips instead of eight
    constexpr int kB = 4;
    for (uint64_t base = t0; base < t1; base += (uint64_t)kB * kPartBlock) {
        uint32_t bb[kB], ll[kB];
#pragma unroll
        for (int k = 0; k < kB; ++k) {
            uint64_t i = base + (uint64_t)k * kPartBlock + threadIdx.x;
            bb[k] = 0; ll[k] = 0;
            if (i < t1) { uint32_t r = first + (uint32_t)i; bb[k] = off[r]; ll[k] = off[r + 1] - bb[k]; }
        }
        uint32_t w[kB][kHead];
#pragma unroll
        for (int k = 0; k < kB; ++k) { const uint32_t* lab = ids + bb[k]; label_head([&](uint32_t q) { return lab[q]; }, ll[k] <= kMaxPartLabel ? ll[k] : 0u, w[k]); }
#pragma unroll
        for (int k = 0; k < kB; ++k) {
            uint64_t i = base + (uint64_t)k * kPartBlock + threadIdx.x;
            if (i >= t1) continue;
            const uint32_t len = ll[k];
            uint16_t rg = 0xFFFFu;                           // 0xFFFF: not in the partition buffer
            if (len > kMaxPartLabel) long_list[atomicAdd(n_long, 1ull)] = first + (uint32_t)i;
            else if (len != 0) {
                const uint32_t* lab = ids + bb[k];
                uint64_t h = (len <= (uint32_t)kHead) ? label_mix64_head(w[k], len)
                                                      : label_mix64_words([&](uint32_t q) { return lab[q]; }, len);
                rg = (uint16_t)((h & mask) >> kRegionBits);
                atomicAdd(&lh[rg], len);
            }
            reg_of[i] = rg;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_regions; i += kPartBlock) mat[(uint64_t)i * gridDim.x + blockIdx.x] = lh[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) mat[(uint64_t)n_regions * gridDim.x] = 0;      // scan sentinel
}

// ---- pass 1b: copy the labels into their region segments (first word head-flagged).
// Writing each label straight to its region would scatter 4-byte stores over n_blocks x n_regions
// open cache lines (measured: 0.72 ms per 16.7 M reads, no better than the random probes it replaces).
// Instead a block counting-sorts a sub-tile of its reads by region inside LDS and then writes every
// region's run with consecutive stores, so HBM sees whole 64..128-byte runs.
constexpr int kSortWords = 24576;                            // 96 KB LDS sort buffer
constexpr int kSubReads = 4096;                              // reads per sub-tile: 4 per thread
constexpr int kSubPer = kSubReads / kPartBlock;

__global__ void __launch_bounds__(kPartBlock)
k_part_scatter(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ off, uint32_t first, uint32_t n,
               uint32_t tile, uint32_t n_regions, const uint16_t* __restrict__ reg_of,
               const uint64_t* __restrict__ offs /* scanned matrix */, uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* buf = reinterpret_cast<uint32_t*>(smem);                               // kSortWords
    unsigned int* hist = reinterpret_cast<unsigned int*>(smem + (size_t)kSortWords * 4);   // n_regions
    unsigned int* sbase = hist + n_regions;                                          // n_regions + 1
    unsigned int* gpos = sbase + n_regions + 1;                                      // n_regions: next free word of this block in each region
    __shared__ unsigned int s_scan[kPartBlock / kWave];
    __shared__ uint32_t s_end;
    for (uint32_t i = threadIdx.x; i < n_regions; i += kPartBlock) gpos[i] = (unsigned int)offs[(uint64_t)i * gridDim.x + blockIdx.x];
    const uint64_t t0 = (uint64_t)blockIdx.x * tile;
    const uint64_t t1 = (t0 + tile < n) ? t0 + tile : n;
    const uint32_t per = (n_regions + kPartBlock - 1) / kPartBlock;                  // regions per thread in the scans
    uint64_t s0 = t0;
    while (s0 < t1) {
        if (threadIdx.x == 0) {                               // sub-tile [s0, s1): at most kSubReads reads and kSortWords words
            uint64_t s1 = (s0 + kSubReads < t1) ? s0 + kSubReads : t1;
            while (s1 - s0 > 1 && off[first + s1] - off[first + s0] > (uint32_t)kSortWords) s1 = s0 + (s1 - s0) / 2;
            s_end = (uint32_t)(s1 - s0);
        }
        for (uint32_t i = threadIdx.x; i < n_regions; i += kPartBlock) hist[i] = 0;
        __syncthreads();
        const uint32_t cnt = s_end;
        uint32_t rg[kSubPer], rk[kSubPer], ln[kSubPer], bs[kSubPer];
#pragma unroll
        for (int k = 0; k < kSubPer; ++k) {
            const uint32_t j = threadIdx.x + k * kPartBlock;
            ln[k] = 0; rg[k] = 0; rk[k] = 0; bs[k] = 0;
            if (j < cnt) {
                const uint16_t r16 = reg_of[s0 + j];
                if (r16 != 0xFFFFu) {
                    const uint32_t r = first + (uint32_t)(s0 + j);
                    bs[k] = off[r]; ln[k] = off[r + 1] - bs[k];
                    if (ln[k] > (uint32_t)kSortWords) ln[k] = 0;      // cannot happen (<= kMaxPartLabel), keeps the buffer safe
                    rg[k] = r16; rk[k] = atomicAdd(&hist[r16], ln[k]);
                }
            }
        }
        __syncthreads();
        // exclusive scan of hist -> sbase (thread t owns regions [t*per, (t+1)*per))
        unsigned int mine = 0;
        for (uint32_t q = 0; q < per; ++q) { uint32_t r = threadIdx.x * per + q; if (r < n_regions) mine += hist[r]; }
        unsigned int incl = mine;
        for (int o = 1; o < kWave; o <<= 1) { unsigned int v = __shfl_up(incl, o, kWave); if ((int)(threadIdx.x & (kWave - 1)) >= o) incl += v; }
        if ((threadIdx.x & (kWave - 1)) == kWave - 1) s_scan[threadIdx.x / kWave] = incl;
        __syncthreads();
        unsigned int run = 0;
        for (int w = 0; w < (int)(threadIdx.x / kWave); ++w) run += s_scan[w];
        run += incl - mine;
        for (uint32_t q = 0; q < per; ++q) { uint32_t r = threadIdx.x * per + q; if (r < n_regions) { sbase[r] = run; run += hist[r]; } }
        __syncthreads();
        // labels -> LDS in region order
#pragma unroll
        for (int k = 0; k < kSubPer; ++k) {
            if (ln[k] == 0) continue;
            const uint32_t* lab = ids + bs[k];
            uint32_t w[kHead];
            label_head([&](uint32_t q) { return lab[q]; }, ln[k], w);
            uint32_t* dst = buf + sbase[rg[k]] + rk[k];
            dst[0] = w[0] | kHeadBit;
#pragma unroll
            for (int q = 1; q < kHead; ++q) if ((uint32_t)q < ln[k]) dst[q] = w[q];
            for (uint32_t q = kHead; q < ln[k]; ++q) dst[q] = lab[q];
        }
        __syncthreads();
        // one lane per region run: consecutive words -> consecutive addresses
        for (uint32_t r = threadIdx.x; r < n_regions; r += kPartBlock) {
            const unsigned int len = hist[r];
            if (len) {
                const uint32_t* src = buf + sbase[r];
                uint32_t* dst = out + gpos[r];
                for (unsigned int q = 0; q < len; ++q) dst[q] = src[q];
                gpos[r] += len;
            }
        }
        __syncthreads();
        s0 += cnt;
    }
}

struct PartArgs {
    uint64_t* table;                       // {word, count} pairs
    const uint64_t* offs; uint32_t n_blocks;   // scanned (region, block) matrix: region r starts at offs[r * n_blocks]
    const uint32_t* words;                 // partition buffer (labels, head-flagged)
    uint64_t* cls_hash; uint64_t* cls_off; uint32_t* cls_len; uint32_t* cls_slot; uint32_t* arena;
    unsigned long long* ctr;               // CTR_* counters (classes / arena cursor / deferred)
    uint32_t* deferred;                    // global word offsets of labels that found their region full
    uint64_t base_classes;                 // classes committed before this launch
};

// label at `p` (first word head-flagged) of `len` words against the stored representative
__device__ __forceinline__ bool stream_label_equals(const uint32_t* a /*label, head masked by caller*/, uint32_t a0,
                                                    const uint32_t* rep, uint32_t len) {
    if ((rep[0] & ~kHeadBit) != a0) return false;
    for (uint32_t i = 1; i < len; ++i) if (rep[i] != a[i]) return false;
    return true;
}

// ---- pass 2: one block per region
__global__ void __launch_bounds__(kPartBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_part_insert(PartArgs a) {
    __shared__ unsigned long long lw[kRegionSlots];     // slot words
    __shared__ unsigned int lc[kRegionSlots];           // count deltas of this launch
    __shared__ uint32_t tile[kTileWords + 4];
    __shared__ uint16_t heads[kTileWords];
    __shared__ uint16_t new_slots[kRegionLimit];
    __shared__ unsigned int s_scan[kPartBlock / kWave];
    __shared__ unsigned int s_nheads, s_occ, s_nnew, s_newwords, s_cid0, s_next;
    __shared__ unsigned long long s_arena0;
    const uint32_t region = blockIdx.x;
    const uint64_t rb = (uint64_t)region * kRegionSlots;
    const uint64_t seg0 = a.offs[(uint64_t)region * a.n_blocks];
    const uint32_t n_words = (uint32_t)(a.offs[(uint64_t)(region + 1) * a.n_blocks] - seg0);
    if (n_words == 0) return;
    const uint32_t* __restrict__ seg = a.words + seg0;

    unsigned int occ_local = 0;
    for (uint32_t s = threadIdx.x; s < kRegionSlots; s += kPartBlock) {
        unsigned long long w = a.table[2 * (rb + s)];
        lw[s] = w; lc[s] = 0; occ_local += (w != kEmpty);
    }
    if (threadIdx.x == 0) { s_occ = 0; s_nnew = 0; s_newwords = 0; }
    __syncthreads();
    if (occ_local) atomicAdd(&s_occ, occ_local);
    __syncthreads();

    uint32_t pos = 0;
    while (pos < n_words) {
        const uint32_t tlen = (n_words - pos < (uint32_t)kTileWords) ? (n_words - pos) : (uint32_t)kTileWords;
        const bool final_tile = (pos + tlen == n_words);
        for (uint32_t i = threadIdx.x; i < tlen; i += kPartBlock) tile[i] = seg[pos + i];
        __syncthreads();
        // ---- head positions, in order (block exclusive scan of per-thread head counts)
        constexpr int kPer = kTileWords / kPartBlock;   // 8 consecutive words per thread
        unsigned int mine = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q) { uint32_t i = threadIdx.x * kPer + q; if (i < tlen && (tile[i] & kHeadBit)) ++mine; }
        unsigned int incl = mine;
        for (int o = 1; o < kWave; o <<= 1) { unsigned int v = __shfl_up(incl, o, kWave); if ((int)(threadIdx.x & (kWave - 1)) >= o) incl += v; }
        if ((threadIdx.x & (kWave - 1)) == kWave - 1) s_scan[threadIdx.x / kWave] = incl;
        __syncthreads();
        unsigned int wave_base = 0;
        for (int w = 0; w < (int)(threadIdx.x / kWave); ++w) wave_base += s_scan[w];
        unsigned int at = wave_base + incl - mine;
#pragma unroll
        for (int q = 0; q < kPer; ++q) { uint32_t i = threadIdx.x * kPer + q; if (i < tlen && (tile[i] & kHeadBit)) heads[at++] = (uint16_t)i; }
        if (threadIdx.x == kPartBlock - 1) s_nheads = wave_base + incl;
        __syncthreads();
        const uint32_t nh = s_nheads;
        // the tile's last label may continue in the next tile: leave it for the next round
        const uint32_t n_proc = final_tile ? nh : (nh > 0 ? nh - 1 : 0);
        for (uint32_t l = threadIdx.x; l < n_proc; l += kPartBlock) {
            const uint32_t st = heads[l];
            const uint32_t en = (l + 1 < nh) ? heads[l + 1] : tlen;
            const uint32_t len = en - st;
            const uint32_t* lab = tile + st;
            const uint32_t w0 = lab[0] & ~kHeadBit;
            uint32_t hw8[kHead];
            const uint64_t h = label_mix64([&](uint32_t k) { return k ? lab[k] : w0; }, len, hw8);
            const uint64_t tag = h >> 32;
            uint32_t s = (uint32_t)h & (kRegionSlots - 1);
            // Probe in two stages so that a wavefront pays the global round trip of the label compare ONCE:
            // (1) walk the LDS slots until an empty slot or a tag match (LDS only: lanes that need a few
            // more steps cost nothing), (2) claim or compare.  With the compare inside the walk, every
            // extra step of any lane repeated the global load for the whole wavefront.
            uint32_t probes = 0;
            for (;;) {
                unsigned long long w = lw[s];
                while (w != kEmpty && (w >> 32) != tag && probes < kRegionSlots) {
                    s = (s + 1) & (kRegionSlots - 1); ++probes; w = lw[s];
                }
                if (probes >= kRegionSlots) {                                             // cannot place: defer
                    a.deferred[atomicAdd(&a.ctr[CTR_DEFER], 1ull)] = (uint32_t)(seg0 + pos + st);
                    break;
                }
                if (w == kEmpty) {
                    if (atomicAdd(&s_occ, 1u) >= kRegionLimit) {                              // region full: defer
                        atomicSub(&s_occ, 1u);
                        a.deferred[atomicAdd(&a.ctr[CTR_DEFER], 1ull)] = (uint32_t)(seg0 + pos + st);
                        break;
                    }
                    unsigned long long me = (tag << 32) | (unsigned long long)(uint32_t)(seg0 + pos + st);
                    unsigned long long old = atomicCAS(&lw[s], (unsigned long long)kEmpty, me);
                    if (old == kEmpty) { new_slots[atomicAdd(&s_nnew, 1u)] = (uint16_t)s; atomicAdd(&lc[s], 1u); break; }
                    atomicSub(&s_occ, 1u);
                    w = old;
                    if ((w >> 32) != tag) { s = (s + 1) & (kRegionSlots - 1); ++probes; continue; }
                }
                // tag match: full label compare
                const uint32_t rep = (uint32_t)w;
                bool same;
                if (rep & kArenaBit) {
                    same = entry_equals(a.arena, rep & ~kArenaBit, [&](uint32_t k) { return lab[k]; }, hw8, len);
                } else {
                    // a label of this launch: `rep` words into the partition buffer; equal iff the first
                    // `len` words match and the representative ends there (next word is a head or the end)
                    const uint32_t* p = a.words + rep;
                    const uint64_t rep_end = (uint64_t)rep + len;
                    same = stream_label_equals(lab, w0, p, len) &&
                           (rep_end >= seg0 + n_words || (a.words[rep_end] & kHeadBit));
                }
                if (same) { atomicAdd(&lc[s], 1u); break; }
                s = (s + 1) & (kRegionSlots - 1); ++probes;
            }
        }
        __syncthreads();
        // labels are <= kMaxPartLabel = half a tile, so a non-final tile always holds >= 2 heads and advances
        if (threadIdx.x == 0) s_next = final_tile ? tlen : ((nh > 1) ? heads[nh - 1] : tlen);
        __syncthreads();
        pos += s_next;
    }

    // ---- commit the classes this block created: ids, arena space, labels, slot re-pointing
    const uint32_t n_new = s_nnew;
    if (n_new) {
        // label lengths: scan to the next head inside the segment
        for (uint32_t i = threadIdx.x; i < n_new; i += kPartBlock) {
            const uint32_t rep = (uint32_t)lw[new_slots[i]];
            uint32_t len = 1;
            while ((uint64_t)rep + len < seg0 + n_words && !(a.words[rep + len] & kHeadBit)) ++len;
            heads[i] = (uint16_t)len;                         // heads[] is free now; len <= kMaxPartLabel < 65536
            atomicAdd(&s_newwords, entry_words(len));
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_cid0 = (unsigned int)atomicAdd(&a.ctr[CTR_NEW], (unsigned long long)n_new);
            s_arena0 = atomicAdd(&a.ctr[CTR_ARENA], (unsigned long long)s_newwords);
            s_newwords = 0;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_new; i += kPartBlock) {
            const uint32_t s = new_slots[i];
            const unsigned long long w = lw[s];
            const uint32_t rep = (uint32_t)w, len = heads[i];
            const uint64_t cid = a.base_classes + s_cid0 + i;
            const uint64_t dst = s_arena0 + atomicAdd(&s_newwords, entry_words(len));
            const uint32_t* p = a.words + rep;
            const uint32_t w0 = p[0] & ~kHeadBit;
            entry_write(a.arena, dst, [&](uint32_t k) { return k ? p[k] : w0; }, len);
            a.cls_hash[cid] = xxh64_words([&](uint32_t k) { return k ? p[k] : w0; }, len);
            a.cls_off[cid] = dst + 1; a.cls_len[cid] = len; a.cls_slot[cid] = (uint32_t)(rb + s);
            lw[s] = (w & 0xFFFFFFFF00000000ull) | (unsigned long long)(kArenaBit | (uint32_t)(dst >> 2));
        }
    }
    __syncthreads();
    // ---- write the region back
    for (uint32_t s = threadIdx.x; s < kRegionSlots; s += kPartBlock) {
        a.table[2 * (rb + s)] = lw[s];
        if (lc[s]) a.table[2 * (rb + s) + 1] += lc[s];
    }
}

// deferred labels (region full): copy them out of the partition buffer into a small CSR batch that the
// generic path can insert after the table has grown
__global__ void k_deferred_lens(uint64_t n, const uint32_t* __restrict__ deferred, const uint32_t* __restrict__ words,
                                uint64_t total_words, uint32_t* lens) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { lens[i] = 0; return; }
    uint64_t at = deferred[i]; uint32_t len = 1;
    while (at + len < total_words && !(words[at + len] & kHeadBit)) ++len;
    lens[i] = len;
}
__global__ void k_deferred_copy(uint64_t n, const uint32_t* __restrict__ deferred, const uint32_t* __restrict__ words,
                                const uint64_t* __restrict__ off64, uint32_t* ids_out, uint32_t* off_out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    off_out[i] = (uint32_t)off64[i];
    if (i == n) return;
    uint32_t len = (uint32_t)(off64[i + 1] - off64[i]);
    const uint32_t* p = words + deferred[i];
    uint32_t* q = ids_out + off64[i];
    q[0] = p[0] & ~kHeadBit;
    for (uint32_t k = 1; k < len; ++k) q[k] = p[k];
}

}  // namespace sfgpu
