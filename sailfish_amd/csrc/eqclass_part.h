// eqclass_part.h -- radix-partitioned class construction (device kernels; included by eqclass.hip).
//
// The global table is an array of REGIONS of kRegionSlots consecutive slots; a label's home region is
// the high bits of its slot index and linear probing wraps inside the region.  One pass of a sub-batch:
//   1  k_part_route  : every label is read ONCE and hashed ONCE (bucket hash, xxh64_device.h); a block counting-sorts
//                      sub-tiles of its reads by region inside LDS and appends each region's run to ITS OWN bin of
//                      that region -- bin (region, block), fixed capacity, no histogram pass, no scan, no global
//                      atomics.  The label travels as  [id0 | head bit][H][id1] ... [id_{n-1}]  where H carries what
//                      pass 2 needs from the hash (19 tag bits, 12 slot bits): pass 2 never hashes a read.
//                      Labels that do not fit their bin (a region far above its share), over-long labels and ids
//                      >= 2^31 go to a list that the generic kernel k_insert takes.
//   2  k_part_insert : ONE block per region: the region's slots live in LDS (word u64 + count delta u32); every
//                      wavefront streams whole bins through a private LDS tile with coalesced loads (no block
//                      barriers while streaming); labels are probed and counted with LDS atomics only; new classes
//                      get their class ids / arena space with one global atomic per block and are committed by the
//                      same block (XXH64 and the full bucket hash are computed here, once per CLASS); finally the
//                      region is written back.
// HBM sees each label twice after the caller's copy (written once, read once), always streaming.
//
// Round-2 rewrite: hardware counters (profiles/r2_eq_counters_before.txt) showed the three kernels of the first
// version (histogram, scatter, insert) issue-bound, not bandwidth-bound: 287 + 262 + 503 vector and 114 + 154 +
// 419 scalar instructions per label (every pass hashed or re-derived the region, predicated loads compiled to
// branch chains).  The route pass replaces histogram + scan + scatter; the insert pass reads H instead of hashing.
#pragma once

namespace sfgpu {

constexpr int kRegionBits = 12;
constexpr uint32_t kRegionSlots = 1u << kRegionBits;          // 4096 slots: 32 KB words + 16 KB counts in LDS
constexpr uint32_t kRegionLimit = kRegionSlots / 4 * 3;       // inserts beyond this occupancy are deferred
constexpr int kPartBlock = 1024;
constexpr int kWaveTile = 256;                                // words of the label stream a wavefront handles at a time
constexpr int kPartWaves = kPartBlock / 64;
constexpr int kMaxRegions = 4096;
constexpr int kMaxRouteBlocks = 1024;                         // bins per region: <= 64 per wavefront of pass 2
constexpr uint32_t kHeadBit = 0x80000000u;
constexpr uint32_t kMaxPartLabel = kWaveTile / 2 - 8;         // ids; a label is n + 1 stream words: always < half a tile
constexpr int kWaveHeads = 68;                                // label starts a wavefront records per tile (it takes <= 64 labels per round)
constexpr int kTagBits = 19;                                  // tag bits carried in H (the table keeps 32)

__device__ __forceinline__ uint64_t region_next(uint64_t s) {
    return (s & ~(uint64_t)(kRegionSlots - 1)) | ((s + 1) & (kRegionSlots - 1));
}

// 16 bytes from a 4-byte-aligned address (global memory takes unaligned vector loads on gfx950)
struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 ld4(const uint32_t* p) { return *reinterpret_cast<const U4*>(p); }

// ---- pass 1: route every label to the bin (region, this block) ------------------------------------------------
// Writing each label straight to its region would scatter 4-byte stores over n_blocks x n_regions open cache lines
// (measured in round 1: no better than the random probes it replaces).  Instead a block counting-sorts a sub-tile of
// its reads by region inside LDS and then writes the sorted buffer out word-parallel, so HBM sees runs and each
// store instruction few lines.  A thread owns kSubPer reads of the sub-tile: it fetches the first 8 ids of each with
// two unaligned 16-byte loads (labels are packed back to back, so a wavefront's loads cover one contiguous range),
// hashes from registers, takes the label's rank inside its region with one LDS atomic, and after the block's scan
// of the region histogram writes the label into the sort buffer.  The offsets of the NEXT sub-tile are requested
// while this one is hashed.
constexpr int kSortWords = 12288;                            // 48 KB LDS sort buffer: two blocks per CU
constexpr int kSubReads = 2048;                              // reads per sub-tile: 2 per thread (their label heads stay in registers)
constexpr int kSubPer = kSubReads / kPartBlock;

struct RouteArgs {
    const uint32_t* ids; const uint32_t* off; uint32_t first, n;
    uint32_t tile;                         // reads per block
    uint32_t ids_end;                      // first id index that must not be read (end of the sub-batch's ids)
    uint64_t mask; uint32_t n_regions;
    uint32_t cap;                          // words per bin
    uint32_t* out;                         // bins: bin (r, b) starts at word (r * n_blocks + b) * cap
    uint32_t* fill;                        // fill[r * n_blocks + b] = words written to bin (r, b)
    unsigned long long* n_long; uint32_t* long_list;     // reads for the generic kernel
};

__global__ void __launch_bounds__(kPartBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_part_route(RouteArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t NR = a.n_regions;
    uint32_t* buf = reinterpret_cast<uint32_t*>(smem);                                // kSortWords (+8 slack)
    unsigned int* hs = reinterpret_cast<unsigned int*>(smem + (size_t)(kSortWords + 8) * 4);   // NR + 1: histogram, then its exclusive scan
    unsigned int* gpos = hs + NR + 1;                                                 // NR: next free word of my bin (absolute word index)
    unsigned int* cut = gpos + NR;                                                    // NR: lowered when a label does not fit the bin
    uint16_t* first_reg = reinterpret_cast<uint16_t*>(cut + NR);                      // kSortWords / 16 + 2: region of every 16th sorted word
    __shared__ unsigned int s_scan[kPartBlock / kWave];
    const uint32_t B1 = gridDim.x, blk = blockIdx.x, cap = a.cap, tid = threadIdx.x;
    for (uint32_t r = tid; r < NR; r += kPartBlock) { gpos[r] = (r * B1 + blk) * cap; cut[r] = 0xFFFFFFFFu; hs[r] = 0; }
    if (tid == 0) hs[NR] = 0;
    const uint64_t t0 = (uint64_t)blk * a.tile;
    const uint64_t t1 = (t0 + a.tile < a.n) ? t0 + a.tile : a.n;
    const uint32_t per = (NR + kPartBlock - 1) / kPartBlock;                          // regions per thread in the scans
    const uint32_t* __restrict__ off = a.off + a.first;
    const uint32_t* __restrict__ ids = a.ids;

    // offsets of the first sub-tile
    uint32_t nb[kSubPer], ne[kSubPer];
#pragma unroll
    for (int k = 0; k < kSubPer; ++k) {
        const uint64_t j = t0 + (uint64_t)k * kPartBlock + tid;
        nb[k] = 0; ne[k] = 0;
        if (j < t1) { nb[k] = off[j]; ne[k] = off[j + 1]; }
    }
    __syncthreads();
    for (uint64_t s0 = t0; s0 < t1; s0 += kSubReads) {
        const uint32_t cnt = (uint32_t)((s0 + kSubReads < t1) ? kSubReads : (t1 - s0));
        uint32_t bs[kSubPer], ln[kSubPer];
#pragma unroll
        for (int k = 0; k < kSubPer; ++k) { bs[k] = nb[k]; ln[k] = ne[k] - nb[k]; }
        // the sub-tile's words (ids + one H per read; uniform): does it fit the sort buffer at once?
        const uint32_t w_all = off[s0 + cnt] - off[s0] + cnt;
        const bool one_round = w_all <= (uint32_t)kSortWords;
        // label heads: first 8 ids of every read, zero past the label
        uint32_t w[kSubPer][kHead];
#pragma unroll
        for (int k = 0; k < kSubPer; ++k) {
            const uint32_t* lab = ids + bs[k];
            if (ln[k] != 0 && (uint64_t)bs[k] + 8u <= (uint64_t)a.ids_end) {
                const U4 x = ld4(lab), y = ld4(lab + 4);
                w[k][0] = x.x; w[k][1] = x.y; w[k][2] = x.z; w[k][3] = x.w; w[k][4] = y.x; w[k][5] = y.y; w[k][6] = y.z; w[k][7] = y.w;
            } else {
#pragma unroll
                for (int q = 0; q < kHead; ++q) w[k][q] = ((uint32_t)q < ln[k]) ? lab[q] : 0u;
            }
#pragma unroll
            for (int q = 0; q < kHead; ++q) w[k][q] = ((uint32_t)q < ln[k]) ? w[k][q] : 0u;
        }
        // the next sub-tile's offsets travel while this one is hashed
#pragma unroll
        for (int k = 0; k < kSubPer; ++k) {
            const uint64_t j = s0 + kSubReads + (uint64_t)k * kPartBlock + tid;
            nb[k] = 0; ne[k] = 0;
            if (j < t1) { nb[k] = off[j]; ne[k] = off[j + 1]; }
        }
        // hash; labels the partition stream cannot carry go to the generic kernel's list
        uint32_t rg[kSubPer], hh[kSubPer];
#pragma unroll
        for (int k = 0; k < kSubPer; ++k) {
            rg[k] = 0; hh[k] = 0;
            const uint32_t len = ln[k];
            if (len == 0) continue;
            uint32_t mx = w[k][0];
#pragma unroll
            for (int q = 1; q < kHead; ++q) mx = mx > w[k][q] ? mx : w[k][q];
            uint64_t h;
            bool generic = len > kMaxPartLabel;
            if (len <= (uint32_t)kHead) h = label_mix64_head(w[k], len);
            else {
                const uint32_t* lab = ids + bs[k];
                h = label_mix64_words([&](uint32_t q) { return lab[q]; }, len);
                if (!generic) for (uint32_t q = kHead; q < len; ++q) { const uint32_t v = lab[q]; mx = mx > v ? mx : v; }
            }
            // an id >= 2^31 would collide with the label marker of the partition stream (no real transcriptome has one)
            generic = generic || (mx & kHeadBit);
            if (generic) {
                a.long_list[atomicAdd(a.n_long, 1ull)] = a.first + (uint32_t)(s0 + (uint64_t)k * kPartBlock + tid);
                ln[k] = 0;
            } else {
                rg[k] = (uint32_t)((h & a.mask) >> kRegionBits);
                hh[k] = ((uint32_t)(h >> (64 - kTagBits)) << kRegionBits) | ((uint32_t)h & (kRegionSlots - 1));
            }
        }
        for (int round = 0; round < (one_round ? 1 : kSubPer); ++round) {
            // ---- rank inside the region (stream words: n ids + H)
            uint32_t rk[kSubPer];
            bool in_round[kSubPer];
#pragma unroll
            for (int k = 0; k < kSubPer; ++k) {
                in_round[k] = ln[k] != 0 && (one_round || k == round);
                rk[k] = 0;
                if (in_round[k]) rk[k] = atomicAdd(&hs[rg[k]], ln[k] + 1u);
            }
            __syncthreads();
            // ---- exclusive scan of the histogram, in place (thread t owns regions [t*per, (t+1)*per))
            unsigned int mine = 0;
            for (uint32_t q = 0; q < per; ++q) { const uint32_t r = tid * per + q; if (r < NR) mine += hs[r]; }
            unsigned int incl = mine;
            for (int o = 1; o < kWave; o <<= 1) { const unsigned int v = __shfl_up(incl, o, kWave); if ((int)(tid & (kWave - 1)) >= o) incl += v; }
            if ((tid & (kWave - 1)) == kWave - 1) s_scan[tid / kWave] = incl;
            __syncthreads();
            unsigned int run = 0;
            for (int q = 0; q < (int)(tid / kWave); ++q) run += s_scan[q];
            run += incl - mine;
            for (uint32_t q = 0; q < per; ++q) {
                const uint32_t r = tid * per + q;
                if (r < NR) {
                    const unsigned int h = hs[r];
                    hs[r] = run;
                    // 16-word blocks of the sorted buffer whose first word falls into region r
                    for (uint32_t bq = (run + 15u) >> 4; bq < ((run + h + 15u) >> 4); ++bq) first_reg[bq] = (uint16_t)r;
                    run += h;
                }
            }
            if (tid == kPartBlock - 1) hs[NR] = run;                                    // = words in this round (sentinel)
            __syncthreads();
            const uint32_t total = hs[NR];
            if (total > (uint32_t)kSortWords) {
                // a half sub-tile that still does not fit (labels of > 10 ids on average): the generic kernel takes it
#pragma unroll
                for (int k = 0; k < kSubPer; ++k)
                    if (in_round[k]) { a.long_list[atomicAdd(a.n_long, 1ull)] = a.first + (uint32_t)(s0 + (uint64_t)k * kPartBlock + tid); ln[k] = 0; }
            } else {
                // ---- labels -> LDS in region order: [id0 | head][H][id1] ...   (ids past the 8th come from global memory: rare)
#pragma unroll
                for (int k = 0; k < kSubPer; ++k) {
                    if (!in_round[k]) continue;
                    const uint32_t len = ln[k];
                    const uint32_t at = gpos[rg[k]] + rk[k];
                    if (at + len + 1u > (rg[k] * B1 + blk + 1u) * cap) {                  // does not fit the bin: this label and all later ones of the run
                        atomicMin(&cut[rg[k]], at);
                        a.long_list[atomicAdd(a.n_long, 1ull)] = a.first + (uint32_t)(s0 + (uint64_t)k * kPartBlock + tid);
                        ln[k] = 0;
                        continue;
                    }
                    uint32_t* dst = buf + hs[rg[k]] + rk[k];
                    dst[0] = w[k][0] | kHeadBit;
                    dst[1] = hh[k];
#pragma unroll
                    for (int q = 1; q < kHead; ++q) if ((uint32_t)q < len) dst[q + 1] = w[k][q];
                    if (len > (uint32_t)kHead) { const uint32_t* lab = ids + bs[k]; for (uint32_t q = kHead; q < len; ++q) dst[q + 1] = lab[q]; }
                    ln[k] = one_round ? len : 0u;                                       // done (the flag only matters in two-round mode)
                }
                __syncthreads();
                // ---- write-out, lane i <-> sorted word i: adjacent lanes write adjacent addresses inside a run
                for (uint32_t i = tid; i < total; i += kPartBlock) {
                    uint32_t r = first_reg[i >> 4];
                    while (hs[r + 1] <= i) ++r;
                    const uint32_t pos = gpos[r] + (i - hs[r]);
                    if (pos < cut[r]) a.out[pos] = buf[i];
                }
            }
            __syncthreads();
            for (uint32_t r = tid; r < NR; r += kPartBlock) {
                const unsigned int end = gpos[r] + (hs[r + 1] - hs[r]);
                const unsigned int c = cut[r];
                if (total <= (uint32_t)kSortWords) gpos[r] = end < c ? end : c;
                cut[r] = 0xFFFFFFFFu;
            }
            __syncthreads();
            for (uint32_t r = tid; r < NR; r += kPartBlock) hs[r] = 0;                   // (hs[r + 1] of the neighbour was read above)
            if (tid == 0) hs[NR] = 0;
            __syncthreads();
        }
    }
    for (uint32_t r = tid; r < NR; r += kPartBlock) a.fill[(uint64_t)r * B1 + blk] = gpos[r] - (r * B1 + blk) * cap;
}

struct PartArgs {
    uint64_t* table;                       // {word, count} pairs
    const uint32_t* words;                 // bins (labels: [id0 | head][H][id1] ...)
    const uint32_t* fill; uint32_t n_blocks; uint32_t cap;     // region r: bins (r, 0 .. n_blocks), fill words each
    uint64_t* cls_hash; uint64_t* cls_off; uint32_t* cls_len; uint32_t* cls_slot; uint32_t* arena;
    unsigned long long* ctr;               // CTR_* counters (classes / arena cursor / deferred)
    uint32_t* deferred;                    // (word index of the label in `words`, length) of labels that found their region full
    uint64_t base_classes;                 // classes committed before this launch
};

// ---- pass 2: one block per region
// A slot word is tag(32) | rep(32) in the table.  While a launch runs, the slot of a class CREATED by it is
// provisional: tag(19) | length(13) | position of its label in the bins (bit 31 clear); it is rewritten with the full
// tag and the arena entry when the block commits its new classes.  Probing compares the 19 tag bits H carries.
// Wavefront w streams the bins w, w + 16, ... of the region, 256 words at a time: four coalesced loads -> its private
// LDS tile; label starts are found with wave ballots (no scan, no barrier); lane j takes the tile's j-th label.  The
// only block-wide synchronisation is before and after the streaming loop.
__global__ void __launch_bounds__(kPartBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_part_insert(PartArgs a) {
    __shared__ unsigned long long lw[kRegionSlots];     // slot words
    __shared__ unsigned int lc[kRegionSlots];           // count deltas of this launch
    __shared__ uint32_t wtile[kPartWaves][kWaveTile + 12];
    __shared__ uint16_t heads[kPartWaves][kWaveHeads];  // per-wave lists of the first label starts of the tile
    __shared__ uint32_t new_info[kRegionLimit];         // classes created by this block: slot | len << 16
    __shared__ unsigned int s_occ, s_nnew, s_newwords, s_cid0;
    __shared__ unsigned long long s_arena0;
    const uint32_t region = blockIdx.x;
    const uint64_t rb = (uint64_t)region * kRegionSlots;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // this wavefront's bins: lane t holds the fill of bin wave + 16 t
    const uint32_t my_bin = wave + kPartWaves * lane;
    const uint32_t my_fill = (my_bin < a.n_blocks) ? a.fill[(uint64_t)region * a.n_blocks + my_bin] : 0u;

    unsigned int occ_local = 0;
    for (uint32_t s = threadIdx.x; s < kRegionSlots; s += kPartBlock) {
        unsigned long long w = a.table[2 * (rb + s)];
        lw[s] = w; lc[s] = 0; occ_local += (w != kEmpty);
    }
    if (threadIdx.x == 0) { s_occ = 0; s_nnew = 0; s_newwords = 0; }
    __syncthreads();
    if (occ_local) atomicAdd(&s_occ, occ_local);
    __syncthreads();

    uint32_t* tile = wtile[wave];
    uint16_t* wh = heads[wave];
    const unsigned long long have = __ballot(my_fill != 0u);
    auto next_bin = [&](int after) -> int {             // first bin t > after with words in it, -1 if none
        const unsigned long long m = (after >= 63) ? 0ull : (have & (~0ull << (after + 1)));
        return m ? (int)__builtin_ctzll(m) : -1;
    };
    int t = next_bin(-1);
    uint32_t seg0 = 0, n_words = 0, pos = 0;             // current bin: first word (index into a.words), words, position
    // word i of a tile = q * 64 + lane: each of the four loads is one coalesced 256-byte access.  The NEXT tile's
    // words (of this bin or of the wavefront's next bin) are requested as soon as this tile's label starts are known,
    // i.e. before the probe / compare of this tile's labels: its round trip hides behind theirs.
    uint32_t tw[4] = {0u, 0u, 0u, 0u};
    auto request = [&](uint32_t s0, uint32_t nw, uint32_t at, uint32_t (&dst)[4]) {
        const uint32_t len = (nw - at < (uint32_t)kWaveTile) ? (nw - at) : (uint32_t)kWaveTile;
        const uint32_t* __restrict__ p = a.words + s0 + at;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint32_t i = q * 64 + lane; dst[q] = (i < len) ? p[i] : 0u; }
    };
    if (t >= 0) {
        seg0 = ((uint32_t)region * a.n_blocks + wave + kPartWaves * (uint32_t)t) * a.cap;
        n_words = __shfl(my_fill, t, kWave);
        request(seg0, n_words, 0u, tw);
    }
    while (t >= 0) {
        const uint32_t tlen = (n_words - pos < (uint32_t)kWaveTile) ? (n_words - pos) : (uint32_t)kWaveTile;
        const bool final_tile = (pos + tlen == n_words);
        uint32_t nh = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t i = q * 64 + lane;
            tile[i] = tw[q];
            const bool is_head = (i < tlen) && (tw[q] & kHeadBit);
            const unsigned long long bal = __ballot(is_head);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (is_head && nh + before < (uint32_t)kWaveHeads) wh[nh + before] = (uint16_t)i;
            nh += (uint32_t)__popcll(bal);
        }
        if (lane < 8u) tile[kWaveTile + lane] = 0u;           // the compare reads up to 7 words past a label
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the tile's last label may continue past the tile: leave it for the next round.  At most one label per lane
        // per round: the labels past the 64th are picked up by the next (overlapping) tile.
        const uint32_t n_whole = final_tile ? nh : (nh > 0 ? nh - 1 : 0);
        const uint32_t n_proc = n_whole < 64u ? n_whole : 64u;
        // where the next tile starts: at the first label that is not processed now.  Labels are < half a tile, so a
        // full tile always holds >= 2 label starts and advances.
        int nt = t; uint32_t nseg0 = seg0, nn_words = n_words, npos = pos;
        if (n_proc < nh) npos = pos + wh[n_proc];
        else if (!final_tile) npos = pos + tlen;           // (cannot happen for full tiles: the last start is never whole)
        else {                                               // this bin is done: the wavefront's next one
            nt = next_bin(t);
            if (nt >= 0) {
                nseg0 = ((uint32_t)region * a.n_blocks + wave + kPartWaves * (uint32_t)nt) * a.cap;
                nn_words = __shfl(my_fill, nt, kWave); npos = 0;
            }
        }
        uint32_t nw[4] = {0u, 0u, 0u, 0u};
        if (nt >= 0) request(nseg0, nn_words, npos, nw);
        if (lane < n_proc) {
            const uint32_t st = wh[lane];
            const uint32_t en = (lane + 1 < nh) ? wh[lane + 1] : tlen;
            const uint32_t len = en - st - 1u;               // ids (the stream label is [id0|head][H][id1]...)
            const uint32_t* lab = tile + st;
            const uint32_t H = lab[1];
            const uint32_t tagq = H >> kRegionBits;
            uint32_t s = H & (kRegionSlots - 1);
            // first 7 ids for the 16-byte compares, zero past the label (word k of the label is lab[k + 1] for k >= 1)
            uint32_t hw8[kHead];
            hw8[0] = lab[0] & ~kHeadBit;
#pragma unroll
            for (int k = 1; k < kHead; ++k) { const uint32_t v = lab[k + 1]; hw8[k] = ((uint32_t)k < len) ? v : 0u; }
            const uint32_t here = seg0 + pos + st;           // where this label sits in the bins
            // Probe in two stages so that a wavefront pays the global round trip of the label compare ONCE:
            // (1) walk the LDS slots until an empty slot or a tag match (LDS only), (2) claim or compare.
            uint32_t probes = 0;
            for (;;) {
                unsigned long long w = lw[s];
                while (w != kEmpty && (uint32_t)(w >> (64 - kTagBits)) != tagq && probes < kRegionSlots) {
                    s = (s + 1) & (kRegionSlots - 1); ++probes; w = lw[s];
                }
                if (probes >= kRegionSlots) {                                             // cannot place: defer
                    const unsigned long long d = atomicAdd(&a.ctr[CTR_DEFER], 1ull);
                    a.deferred[2 * d] = here; a.deferred[2 * d + 1] = len;
                    break;
                }
                if (w == kEmpty) {
                    if (atomicAdd(&s_occ, 1u) >= kRegionLimit) {                              // region full: defer
                        atomicSub(&s_occ, 1u);
                        const unsigned long long d = atomicAdd(&a.ctr[CTR_DEFER], 1ull);
                        a.deferred[2 * d] = here; a.deferred[2 * d + 1] = len;
                        break;
                    }
                    const unsigned long long me = ((unsigned long long)tagq << (64 - kTagBits)) | ((unsigned long long)len << 32) | (unsigned long long)here;
                    const unsigned long long old = atomicCAS(&lw[s], (unsigned long long)kEmpty, me);
                    if (old == kEmpty) { new_info[atomicAdd(&s_nnew, 1u)] = s | (len << 16); atomicAdd(&lc[s], 1u); break; }
                    atomicSub(&s_occ, 1u);
                    w = old;
                    if ((uint32_t)(w >> (64 - kTagBits)) != tagq) { s = (s + 1) & (kRegionSlots - 1); ++probes; continue; }
                }
                // tag match: full label compare
                const uint32_t rep = (uint32_t)w;
                bool same;
                if (rep & kArenaBit) {
                    same = entry_equals(a.arena, rep & ~kArenaBit, [&](uint32_t k) { return lab[k + 1]; }, hw8, len);
                } else {
                    // a class of this launch: its label sits in the bins at `rep`, its length in the slot word
                    same = ((uint32_t)(w >> 32) & 0x1FFFu) == len;
                    if (same) {
                        const uint32_t* p = a.words + rep;
                        same = (p[0] & ~kHeadBit) == hw8[0];
                        for (uint32_t k = 1; same && k < len; ++k) same = p[k + 1] == lab[k + 1];
                    }
                }
                if (same) { atomicAdd(&lc[s], 1u); break; }
                s = (s + 1) & (kRegionSlots - 1); ++probes;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                      // all lanes are done with the tile before it is refilled
        t = nt; seg0 = nseg0; n_words = nn_words; pos = npos;
#pragma unroll
        for (int q = 0; q < 4; ++q) tw[q] = nw[q];
    }
    __syncthreads();

    // ---- commit the classes this block created: ids, arena space, labels, hashes, slot re-pointing
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
            auto word = [&](uint32_t k) { return k ? p[k + 1] : w0; };
            entry_write(a.arena, dst, word, len);
            a.cls_hash[cid] = xxh64_words(word, len);
            a.cls_off[cid] = dst + 1; a.cls_len[cid] = len; a.cls_slot[cid] = (uint32_t)(rb + s);
            uint32_t tmp[kHead];
            const uint64_t h = label_mix64(word, len, tmp);                               // the table keeps the full 32-bit tag
            lw[s] = (h & 0xFFFFFFFF00000000ull) | (unsigned long long)(kArenaBit | (uint32_t)(dst >> 2));
        }
    }
    __syncthreads();
    // ---- write the region back
    for (uint32_t s = threadIdx.x; s < kRegionSlots; s += kPartBlock) {
        a.table[2 * (rb + s)] = lw[s];
        if (lc[s]) a.table[2 * (rb + s) + 1] += lc[s];
    }
}

// deferred labels (region full): copy them out of the bins into a small CSR batch that the generic path can insert
// after the table has grown.  deferred[2 i] = word index of the label in the bins, deferred[2 i + 1] = its length.
__global__ void k_deferred_lens(uint64_t n, const uint32_t* __restrict__ deferred, uint32_t* lens) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    lens[i] = (i == n) ? 0u : deferred[2 * i + 1];
}
__global__ void k_deferred_copy(uint64_t n, const uint32_t* __restrict__ deferred, const uint32_t* __restrict__ words,
                                const uint64_t* __restrict__ off64, uint32_t* ids_out, uint32_t* off_out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    off_out[i] = (uint32_t)off64[i];
    if (i == n) return;
    uint32_t len = (uint32_t)(off64[i + 1] - off64[i]);
    const uint32_t* p = words + deferred[2 * i];
    uint32_t* q = ids_out + off64[i];
    q[0] = p[0] & ~kHeadBit;
    for (uint32_t k = 1; k < len; ++k) q[k] = p[k + 1];
}

}  // namespace sfgpu
