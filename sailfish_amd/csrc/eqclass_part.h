// eqclass_part.h -- radix-partitioned class construction (device kernels; included by eqclass.hip).
//
// The global table is an array of REGIONS of kRegionSlots consecutive slots; a label's home region is
// the high bits of its slot index and linear probing wraps inside the region.  One pass of a sub-batch:
//   1a k_part_hist    : hash every label, histogram label WORDS per region (LDS histogram per block)
//      (scan)
//   1b k_part_scatter : copy every label into its region's segment of a partition buffer -- first word
//                       flagged with bit 31, so a segment is a self-delimiting stream of labels
//   2  k_part_insert  : ONE block per region: the region's slots live in LDS (word u64 + count delta
//                       u32); every wavefront streams its own share of the segment through a private
//                       LDS tile with coalesced loads (no block barriers while streaming); labels
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
constexpr int kWaveTile = 256;                                // words of the label stream a wavefront handles at a time
constexpr int kPartWaves = kPartBlock / 64;
constexpr int kMaxRegions = 4096;                             // LDS histogram size of passes 1a/1b
constexpr uint32_t kHeadBit = 0x80000000u;
constexpr uint32_t kMaxPartLabel = kWaveTile / 2;             // longer labels take the generic path
constexpr int kWaveHeads = 68;                                // label starts a wavefront records per tile (it takes <= 64 labels per round)

__device__ __forceinline__ uint64_t region_next(uint64_t s) {
    return (s & ~(uint64_t)(kRegionSlots - 1)) | ((s + 1) & (kRegionSlots - 1));
}

// ---- pass 1a: hash every label once; remember its region; count label words per (block, region).
// Block b owns the reads [b*tile, (b+1)*tile) in both 1a and 1b, and writes its histogram row into a
// region-major matrix mat[region * n_blocks + b]; an exclusive scan of that matrix is then every
// (region, block) pair's output offset -- no global atomics, and the partition is stable across blocks.
// The ids of a block's reads are one contiguous range, so a chunk of reads is first STAGED in LDS with
// 16-byte coalesced loads and the lanes then pick their labels out of LDS.  Reading labels lane-per-read
// straight from global memory costs one load instruction per id with 64 different cache lines behind
// it; that address-processing rate, not HBM, bounded the pass (0.30 ms per 16.7 M reads; 0.16 ms staged).
constexpr int kHistPer = 2;                                   // reads per lane per chunk
constexpr int kHistChunk = kHistPer * kPartBlock;             // 2048 reads
constexpr int kStageWords = 15872;                            // 62 KB: two blocks per CU with the 16 KB histogram

__global__ void __launch_bounds__(kPartBlock)
k_part_hist(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ off, uint32_t first, uint32_t n,
            uint32_t tile, uint64_t mask, uint32_t n_regions, uint16_t* __restrict__ reg_of, uint32_t* __restrict__ mat,
            unsigned long long* n_long, uint32_t* long_list) {
    __shared__ unsigned int lh[kMaxRegions];
    __shared__ __attribute__((aligned(16))) uint32_t stage[kStageWords + 4];
    for (uint32_t i = threadIdx.x; i < n_regions; i += kPartBlock) lh[i] = 0;
    const uint64_t t0 = (uint64_t)blockIdx.x * tile;
    const uint64_t t1 = (t0 + tile < n) ? t0 + tile : n;
    for (uint64_t base = t0; base < t1; base += kHistChunk) {
        const uint64_t cend = (base + kHistChunk < t1) ? base + kHistChunk : t1;
        // id range of the chunk (uniform): [w_lo, w_hi), staged from the 16-byte boundary at or below
        // ids + w_lo.  Only 16-byte granules that hold at least one word of the range are touched.
        const uint32_t w_lo = off[first + (uint32_t)base], w_hi = off[first + (uint32_t)cend];
        const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(ids + w_lo) >> 2) & 3u);
        const uint32_t span = w_hi - w_lo + mis;
        const bool staged = span <= (uint32_t)kStageWords;
        uint32_t bb[kHistPer], ll[kHistPer];
#pragma unroll
        for (int k = 0; k < kHistPer; ++k) {
            uint64_t i = base + (uint64_t)k * kPartBlock + threadIdx.x;
            bb[k] = 0; ll[k] = 0;
            if (i < cend) { uint32_t r = first + (uint32_t)i; bb[k] = off[r]; ll[k] = off[r + 1] - bb[k]; }
        }
        __syncthreads();                                      // previous chunk's labels are consumed (and lh is zeroed)
        if (staged) {
            const uint4* src = reinterpret_cast<const uint4*>(ids + w_lo - mis);
            const uint32_t n4 = (span + 3) >> 2;
            for (uint32_t i = threadIdx.x; i < n4; i += kPartBlock) {
                const uint4 v = src[i];
                *reinterpret_cast<uint4*>(stage + 4 * i) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kHistPer; ++k) {
            uint64_t i = base + (uint64_t)k * kPartBlock + threadIdx.x;
            if (i >= cend) continue;
            const uint32_t len = ll[k];
            uint16_t rg = 0xFFFFu;                           // 0xFFFF: not in the partition buffer
            if (len > kMaxPartLabel) long_list[atomicAdd(n_long, 1ull)] = first + (uint32_t)i;
            else if (len != 0) {
                uint32_t w[kHead];
                uint64_t h;
                uint32_t any = 0;                            // OR of the label's ids
                if (staged) {
                    const uint32_t* lab = stage + (bb[k] - w_lo + mis);
                    h = label_mix64([&](uint32_t q) { return lab[q]; }, len, w);
                    for (uint32_t q = kHead; q < len; ++q) any |= lab[q];
                } else {
                    const uint32_t* lab = ids + bb[k];
                    h = label_mix64([&](uint32_t q) { return lab[q]; }, len, w);
                    for (uint32_t q = kHead; q < len; ++q) any |= lab[q];
                }
#pragma unroll
                for (int q = 0; q < kHead; ++q) any |= w[q];
                if (any & kHeadBit) {
                    // an id >= 2^31 would collide with the label marker of the partition stream: such
                    // labels (no real transcriptome has them) take the generic kernel, like over-long ones
                    long_list[atomicAdd(n_long, 1ull)] = first + (uint32_t)i;
                } else {
                    rg = (uint16_t)((h & mask) >> kRegionBits);
                    atomicAdd(&lh[rg], len);
                }
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
// Instead a block counting-sorts a sub-tile of its reads by region inside LDS and then writes the sorted
// buffer out word-parallel, so HBM sees whole 64..128-byte runs and each store instruction few lines.
// The sub-tile's ids are staged in the sort buffer with coalesced 16-byte loads (see k_part_hist), the
// lanes lift their labels' first 8 ids into registers, and the same buffer is then refilled in region
// order -- one LDS buffer serves as both the staging area and the sort destination.
// Two blocks share a CU (48 KB buffer, <= 64 VGPRs): rocprofv3 shows these passes' wavefronts parked on
// s_waitcnt / barriers for 60-80 % of their cycles, so a second block that works while the first waits is
// worth more than longer runs (96 KB buffer, 4096-read sub-tiles, one block per CU: 0.35 -> 0.29 ms).
// (Tried and dropped: prefetching the next sub-tile into registers during the write-out -- vmcnt counts
// loads and stores together on gfx9, so the block still waits for its stores to drain.)
constexpr int kSortWords = 12288;                            // 48 KB LDS sort buffer: two blocks per CU
constexpr int kSubReads = 2048;                              // reads per sub-tile: 2 per thread (their label heads stay in registers)
constexpr int kSubPer = kSubReads / kPartBlock;

__global__ void __launch_bounds__(kPartBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_part_scatter(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ off, uint32_t first, uint32_t n,
               uint32_t tile, uint32_t n_regions, const uint16_t* __restrict__ reg_of,
               const uint64_t* __restrict__ offs /* scanned matrix */, uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* buf = reinterpret_cast<uint32_t*>(smem);                               // kSortWords (+4 slack)
    unsigned int* hist = reinterpret_cast<unsigned int*>(smem + (size_t)(kSortWords + 4) * 4);   // n_regions
    unsigned int* sbase = hist + n_regions;                                          // n_regions + 1
    unsigned int* gpos = sbase + n_regions + 1;                                      // n_regions: next free word of this block in each region
    uint16_t* first_reg = reinterpret_cast<uint16_t*>(gpos + n_regions);             // kSortWords / 16 + 1: region of every 16th sorted word
    __shared__ unsigned int s_scan[kPartBlock / kWave];
    __shared__ uint32_t s_end;
    for (uint32_t i = threadIdx.x; i < n_regions; i += kPartBlock) gpos[i] = (unsigned int)offs[(uint64_t)i * gridDim.x + blockIdx.x];
    const uint64_t t0 = (uint64_t)blockIdx.x * tile;
    const uint64_t t1 = (t0 + tile < n) ? t0 + tile : n;
    const uint32_t per = (n_regions + kPartBlock - 1) / kPartBlock;                  // regions per thread in the scans
    uint64_t s0 = t0;
    while (s0 < t1) {
        // sub-tile [s0, s0 + cnt): at most kSubReads reads whose ids fit the buffer (uniform decision)
        uint32_t cnt = (uint32_t)((s0 + kSubReads < t1) ? kSubReads : (t1 - s0));
        const uint32_t w_lo = off[first + (uint32_t)s0];
        const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(ids + w_lo) >> 2) & 3u);
        uint32_t w_hi = off[first + (uint32_t)(s0 + cnt)];
        if (w_hi - w_lo + mis > (uint32_t)kSortWords) {       // rare: halve until it fits (or one read is left)
            if (threadIdx.x == 0) {
                uint32_t c = cnt;
                while (c > 1 && off[first + (uint32_t)(s0 + c)] - w_lo + mis > (uint32_t)kSortWords) c /= 2;
                s_end = c;
            }
            __syncthreads();
            cnt = s_end;
            w_hi = off[first + (uint32_t)(s0 + cnt)];
        }
        const uint32_t span = w_hi - w_lo + mis;
        const bool staged = span <= (uint32_t)kSortWords;    // false only for a single over-long label (which is skipped)
        for (uint32_t i = threadIdx.x; i < n_regions; i += kPartBlock) hist[i] = 0;
        uint32_t rg[kSubPer], rk[kSubPer], ln[kSubPer], bs[kSubPer];
#pragma unroll
        for (int k = 0; k < kSubPer; ++k) {
            const uint32_t j = threadIdx.x + k * kPartBlock;
            ln[k] = 0; rg[k] = 0xFFFFu; bs[k] = 0;
            if (j < cnt) {
                rg[k] = reg_of[s0 + j];
                const uint32_t r = first + (uint32_t)(s0 + j);
                bs[k] = off[r]; ln[k] = off[r + 1] - bs[k];
                if (rg[k] == 0xFFFFu || ln[k] > (uint32_t)kSortWords) ln[k] = 0;     // not partitioned (empty / over-long label)
            }
        }
        if (staged) {
            const uint4* src = reinterpret_cast<const uint4*>(ids + w_lo - mis);
            const uint32_t n4 = (span + 3) >> 2;
            for (uint32_t i = threadIdx.x; i < n4; i += kPartBlock) *reinterpret_cast<uint4*>(buf + 4 * i) = src[i];
        }
        __syncthreads();                                      // ids staged, hist zeroed
        uint32_t w[kSubPer][kHead];
#pragma unroll
        for (int k = 0; k < kSubPer; ++k) {
            rk[k] = 0;
            if (ln[k]) {
                rk[k] = atomicAdd(&hist[rg[k]], ln[k]);
                if (staged) { const uint32_t* lab = buf + (bs[k] - w_lo + mis); label_head([&](uint32_t q) { return lab[q]; }, ln[k], w[k]); }
                else { const uint32_t* lab = ids + bs[k]; label_head([&](uint32_t q) { return lab[q]; }, ln[k], w[k]); }
            }
        }
        __syncthreads();                                      // label heads are in registers: the buffer may be overwritten
        // exclusive scan of hist -> sbase (thread t owns regions [t*per, (t+1)*per))
        unsigned int mine = 0;
        for (uint32_t q = 0; q < per; ++q) { uint32_t r = threadIdx.x * per + q; if (r < n_regions) mine += hist[r]; }
        unsigned int incl = mine;
        for (int o = 1; o < kWave; o <<= 1) { unsigned int v = __shfl_up(incl, o, kWave); if ((int)(threadIdx.x & (kWave - 1)) >= o) incl += v; }
        if ((threadIdx.x & (kWave - 1)) == kWave - 1) s_scan[threadIdx.x / kWave] = incl;
        __syncthreads();
        unsigned int run = 0;
        for (int q = 0; q < (int)(threadIdx.x / kWave); ++q) run += s_scan[q];
        run += incl - mine;
        for (uint32_t q = 0; q < per; ++q) {
            uint32_t r = threadIdx.x * per + q;
            if (r < n_regions) {
                const unsigned int h = hist[r];
                sbase[r] = run;
                // 16-word blocks of the sorted buffer whose first word falls into region r
                for (uint32_t b = (run + 15u) >> 4; b < ((run + h + 15u) >> 4); ++b) first_reg[b] = (uint16_t)r;
                run += h;
            }
        }
        if (threadIdx.x == kPartBlock - 1) sbase[n_regions] = run;                   // = words in the sub-tile (sentinel)
        __syncthreads();
        // labels -> LDS in region order (ids past the 8th come from global memory: rare)
#pragma unroll
        for (int k = 0; k < kSubPer; ++k) {
            if (ln[k] == 0) continue;
            uint32_t* dst = buf + sbase[rg[k]] + rk[k];
            dst[0] = w[k][0] | kHeadBit;
#pragma unroll
            for (int q = 1; q < kHead; ++q) if ((uint32_t)q < ln[k]) dst[q] = w[k][q];
            if (ln[k] > (uint32_t)kHead) { const uint32_t* lab = ids + bs[k]; for (uint32_t q = kHead; q < ln[k]; ++q) dst[q] = lab[q]; }
        }
        __syncthreads();
        // write-out, lane i <-> sorted word i: adjacent lanes write adjacent addresses inside a run, so a
        // store instruction touches a handful of cache lines (one per run it spans) instead of 64.  The
        // word's region comes from the 16-word block table plus a short walk over region boundaries.
        const uint32_t total = sbase[n_regions];
        for (uint32_t i = threadIdx.x; i < total; i += kPartBlock) {
            uint32_t r = first_reg[i >> 4];
            while (sbase[r + 1] <= i) ++r;
            out[gpos[r] + (i - sbase[r])] = buf[i];
        }
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < n_regions; r += kPartBlock) gpos[r] += hist[r];
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
// The 16 wavefronts of the block split the region's segment into equal word ranges; a label belongs to
// the wavefront whose range holds its first word.  Each wavefront works alone: 256 words -> registers
// (four coalesced loads) -> its private LDS tile; label starts are found with wave ballots (no scan,
// no barrier); lane j takes the tile's j-th label.  The only block-wide synchronisation is before and
// after the streaming loop, so the 32 wavefronts resident on a CU hide each other's memory latency.
__global__ void __launch_bounds__(kPartBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_part_insert(PartArgs a) {
    __shared__ unsigned long long lw[kRegionSlots];     // slot words
    __shared__ unsigned int lc[kRegionSlots];           // count deltas of this launch
    __shared__ uint32_t wtile[kPartWaves][kWaveTile + 4];
    __shared__ uint16_t heads[kPartWaves][kWaveHeads];  // per-wave lists of the first label starts of the tile
    __shared__ uint32_t new_info[kRegionLimit];         // classes created by this block: slot | len << 16
    __shared__ unsigned int s_occ, s_nnew, s_newwords, s_cid0;
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

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* tile = wtile[wave];
    uint16_t* wh = heads[wave];
    const uint32_t w_beg = (uint32_t)((uint64_t)n_words * wave / kPartWaves);
    const uint32_t w_end = (uint32_t)((uint64_t)n_words * (wave + 1) / kPartWaves);
    uint32_t pos = w_beg;
    // word i of a tile = q * 64 + lane: each of the four loads is one coalesced 256-byte access.  The
    // NEXT tile's words are requested as soon as this tile's label starts are known, i.e. before the
    // hash / probe / compare of this tile's labels: its round trip hides behind theirs.
    uint32_t tw[4];
    auto request = [&](uint32_t at, uint32_t (&dst)[4]) {
        const uint32_t len = (n_words - at < (uint32_t)kWaveTile) ? (n_words - at) : (uint32_t)kWaveTile;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint32_t i = q * 64 + lane; dst[q] = (i < len) ? seg[at + i] : 0u; }
    };
    if (pos < w_end) request(pos, tw);
    while (pos < w_end) {
        const uint32_t tlen = (n_words - pos < (uint32_t)kWaveTile) ? (n_words - pos) : (uint32_t)kWaveTile;
        const bool final_tile = (pos + tlen == n_words);
        uint32_t nh = 0, in_range = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t i = q * 64 + lane;
            tile[i] = tw[q];
            const bool is_head = (i < tlen) && (tw[q] & kHeadBit);
            const unsigned long long bal = __ballot(is_head);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (is_head && nh + before < (uint32_t)kWaveHeads) wh[nh + before] = (uint16_t)i;
            nh += (uint32_t)__popcll(bal);
            // label starts that belong to this wavefront (first word before w_end)
            in_range += (uint32_t)__popcll(__ballot(is_head && pos + i < w_end));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the tile's last label may continue past the tile: leave it for the next round
        const uint32_t n_whole = final_tile ? nh : (nh > 0 ? nh - 1 : 0);
        // at most one label per lane per round: a second, mostly idle pass over the tile would cost as much
        // as a full one -- the labels past the 64th are simply picked up by the next (overlapping) tile
        uint32_t n_proc = n_whole < in_range ? n_whole : in_range;
        if (n_proc > 64u) n_proc = 64u;
        // where the next tile starts: at the first label that is not processed now; stop once that label is
        // another wavefront's.  Labels are <= kMaxPartLabel = half a tile, so a full tile always holds >= 2
        // label starts and advances.
        uint32_t next_pos; bool more;
        if (n_proc < in_range) { const uint32_t adv = wh[n_proc]; next_pos = pos + (adv ? adv : tlen); more = true; }
        else if (n_proc < nh || final_tile) { next_pos = pos; more = false; }     // the next label start lies at or beyond w_end (or the segment ended)
        else { next_pos = pos + tlen; more = true; }                             // no label start left in this tile (cannot happen for full tiles)
        more = more && next_pos < w_end;
        uint32_t nw[4] = {0u, 0u, 0u, 0u};
        if (more) request(next_pos, nw);
        for (uint32_t l = lane; l < n_proc; l += 64) {
            const uint32_t st = wh[l];
            const uint32_t en = (l + 1 < nh) ? wh[l + 1] : tlen;
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
                    if (old == kEmpty) { new_info[atomicAdd(&s_nnew, 1u)] = s | (len << 16); atomicAdd(&lc[s], 1u); break; }
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
                    // `len` words match and the representative ends there (next word is a head or the end).
                    // Its first 4 words and the word behind it are requested together: one round trip for most labels.
                    const uint32_t* p = a.words + rep;
                    const uint64_t rep_end = (uint64_t)rep + len;
                    same = stream_label_equals(lab, w0, p, len) &&
                           (rep_end >= seg0 + n_words || (a.words[rep_end] & kHeadBit));
                }
                if (same) { atomicAdd(&lc[s], 1u); break; }
                s = (s + 1) & (kRegionSlots - 1); ++probes;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                      // all lanes are done with the tile before it is refilled
        if (!more) break;
        pos = next_pos;
#pragma unroll
        for (int q = 0; q < 4; ++q) tw[q] = nw[q];
    }
    __syncthreads();

    // ---- commit the classes this block created: ids, arena space, labels, slot re-pointing
    const uint32_t n_new = s_nnew;
    if (n_new) {
        for (uint32_t i = threadIdx.x; i < n_new; i += kPartBlock) atomicAdd(&s_newwords, entry_words(new_info[i] >> 16));
        __syncthreads();
        if (threadIdx.x == 0) {
            s_cid0 = (unsigned int)atomicAdd(&a.ctr[CTR_NEW], (unsigned long long)n_new);
            s_arena0 = atomicAdd(&a.ctr[CTR_ARENA], (unsigned long long)s_newwords);
            s_newwords = 0;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_new; i += kPartBlock) {
            const uint32_t s = new_info[i] & 0xFFFFu, len = new_info[i] >> 16;
            const unsigned long long w = lw[s];
            const uint32_t rep = (uint32_t)w;
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
