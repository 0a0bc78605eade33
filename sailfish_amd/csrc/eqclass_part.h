// eqclass_part.h -- radix-partitioned class construction (device kernels; included by eqclass.hip).
//
// The global table is an array of REGIONS of kRegionSlots consecutive slots; a label's home region is
// the high bits of its slot index and linear probing wraps inside the region.  One pass of a sub-batch:
//   1  k_part_route  : every label is read ONCE and hashed ONCE (bucket hash, xxh64_device.h) by one lane, which then
//                      stores it into ITS BLOCK'S bin of the label's region -- bin (region, block), fixed capacity,
//                      position from one LDS atomic; no histogram pass, no scan, no sort, no barrier, no global
//                      atomics.  The label travels as 16-byte granules  [id0 | head bit][H][id1][id2] [id3..id6] ...
//                      where H carries what pass 2 needs (length, 12 tag bits, 12 slot bits): pass 2 never hashes a
//                      read.  Labels that do not fit their bin (a region far above its share), over-long labels and
//                      ids >= 2^31 go to a list that the generic kernel k_insert takes.
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
// branch chains).  The route pass replaces histogram + scan + scatter; the insert pass reads H instead of hashing
// and takes one granule per lane instead of searching label starts in a word stream.
#pragma once

namespace sfgpu {

#ifndef SFGPU_REGION_BITS
#define SFGPU_REGION_BITS 12
#endif
constexpr int kRegionBits = SFGPU_REGION_BITS;
constexpr uint32_t kRegionSlots = 1u << kRegionBits;          // 4096 slots: 32 KB words + 16 KB counts in LDS
constexpr uint32_t kRegionLimit = kRegionSlots / 4 * 3;       // inserts beyond this occupancy are deferred
constexpr int kPartBlock = 1024;
constexpr int kPartWaves = kPartBlock / 64;
constexpr int kMaxRegions = 1 << 19;                          // the whole range of the table (2^31 slots); the passes handle kGroupRegions at a time
constexpr uint32_t kGroupRegions = 4096;                      // regions per launch pair: 2 x 4096 x 4 B of cursors next to a second block in LDS
constexpr uint32_t kHeadBit = 0x80000000u;
constexpr uint32_t kMaxPartLabel = 123;                       // ids: the length travels in 7 bits of H, a label is <= 31 granules
constexpr int kTagBits = 24 - kRegionBits;                    // tag bits carried in H next to the slot (the table keeps 32)

__device__ __forceinline__ uint64_t region_next(uint64_t s) {
    return (s & ~(uint64_t)(kRegionSlots - 1)) | ((s + 1) & (kRegionSlots - 1));
}

// The partition stream is made of 16-byte GRANULES.  A label of n ids takes ceil((n + 1) / 4) of them:
//    [id0 | head bit][H][id1][id2]   [id3][id4][id5][id6]   ...   (zero padded)
// i.e. label word k >= 1 sits k + 1 words after the label's start.  H = n << 24 | tag << 12 | slot: everything pass 2
// needs from the bucket hash (12 tag bits, the slot inside the region) and the length, so pass 2 hashes nothing and
// finds label boundaries with one ballot.
__device__ __forceinline__ uint32_t label_granules(uint32_t n) { return (n + 4u) >> 2; }

// ---- pass 1: route every label to the bin (region, this block) ------------------------------------------------
// No sort, no scan, no block barrier.  A wavefront takes 64 consecutive reads per step, copies their ids (one contiguous
// range) into its own LDS buffer with coalesced 16-byte loads, and every lane then hashes its label out of LDS, takes
// its place in the bin with ONE LDS atomic on the block's cursor of that region, and stores the label itself -- one
// 16-byte store for <= 3 ids, two for <= 7.  The stores of a wavefront go to 64 different bins; what keeps that
// affordable is their width (1.6 stores per label instead of one per id).  What round 2 measured on the way
// (profiles/r2_class_build_notes.md): counting-sorting sub-tiles in LDS first (runs instead of single labels) cost five
// barriers and ~90 LDS + ~630 vector instructions per label; lane-per-label 16-byte loads of the ids were bound by the
// texture addresser (one cycle per lane and dword); ids past the 8th read one word at a time made a chain of dependent
// loads that some lane of every wavefront had to walk (13 % of the labels hold more than 7 ids).  With those gone the
// pass is bound by its scattered stores: 0.51 ms of 1.55 ms per 67 M reads without them, 0.88 ms with all stores
// confined to 1 MB -- the rest is partial lines leaving the L2 (64 K open lines per XCD).
constexpr int kStageWords = 512;                             // ids a wavefront stages per step (64 consecutive reads): 2 KB

// ---- the RING form of pass 1 (round 3) ---------------------------------------------------------------------------------
// What bounded the pass above was measured in round 2: 1.6 scattered 16-byte stores per label, each to a different bin, whose
// 128-byte lines leave the L2 half full (512 blocks x 1024 bins = 64 K open lines per XCD): WRITE_SIZE 1.8x the payload, the
// texture addresser busy 12.2 of 12.5 cycles per read.  The ring form keeps the tail of every bin of the block in LDS -- a
// ring of two 64-byte UNITS (4 granules each) per region, 128 KB for 1024 regions, ONE block per CU -- and only whole units
// leave the CU, four lanes per unit, 16 units per store instruction: every 64-byte segment of the stream is written once, whole.
//   * cbst[r].x = front cursor (low 16 bits) | back cursor (high 16 bits), in granules.  Labels of <= 4 granules (98.7 % of
//     the benchmark's) grow the bin from its front through the ring; longer ones are stored directly, as before, from the
//     bin's BACK downwards (they are runs of >= 80 contiguous bytes anyway): pass 2 streams both segments.  One returning LDS
//     atomic gives a lane both cursors, so "does it fit" is exact.
//   * cbst[r].y = per unit parity q (16 bits each): granules written into the open unit of that parity (3 bits) | units of
//     that parity flushed so far (12 bits).  Granule p lives in unit v = p / 4, ring slot p % 8; unit v may be written when
//     v / 2 units of its parity have been flushed (the ring slot is free).  A writer adds its granules to the counts AFTER
//     writing them; the add that brings a count to 4 makes its lane the unit's COMPLETER: the wavefront flushes the completed
//     units of the step together, then the completers add 4 -- count back to 0, carry into the flushed field -- which opens
//     the slot for unit v + 2.  A lane that finds its unit closed (the ring is two units deep; it needs 3+ labels of a region
//     in flight at once) retries after the flush; no lane ever holds a completed unit while it waits, so this cannot deadlock.
//   * LDS executes a wavefront's instructions in order, which is what orders granule write -> count add -> (completer) unit
//     read -> release add between wavefronts; the fences below are for the compiler.
// What is left in the ring when the block ends (the last, partial unit of every bin) is written by one thread per region.
constexpr int kRingStageWords = 304;                          // staging per wavefront: (304 / 4 + 4) x 16 B = 1280 B (mean step: 256 ids; beyond: labels from global memory)
constexpr uint32_t kRingHotSlots = 256;                       // what 160 KB leave for the hot table (32-bit tags + counters: 2 KB)
constexpr uint32_t kRingFlushList = 32;                       // completed units a wavefront lists per flush round (128 B per wavefront)
constexpr uint32_t kRingMaxRegions = 1024;                    // 128 B of ring per region
constexpr uint32_t kRingMaxCap = 32000;                       // 16-bit cursors with room for the reservations of every lane in flight
constexpr uint32_t kRingFrontGranules = 4;                    // labels of more granules take the back of the bin

struct RouteArgs {
    const uint32_t* ids; const uint32_t* off;      // off is already advanced to the sub-batch's first read
    uint32_t first;                                // index of that read in the caller's batch (for the generic kernel's list)
    uint32_t n;                                    // reads in the sub-batch
    uint32_t tile;                                 // reads per block (a multiple of 64)
    uint32_t region_mask;                          // n_regions - 1
    uint32_t cap;                                  // granules per bin
    uint4* out;                                    // bins: bin (r, b) starts at granule (b * n_regions + r) * cap -- a block's bins are one window
                                                   // of memory (region-major, every store of a block hit a different 2 MB page)
    uint32_t* fill;                                // fill[r * n_blocks + b] = granules written to bin (r, b) (from its front)
    uint32_t* fill_back;                           // ... and from its back (ring form; 0 otherwise)
    uint32_t* cut;                                 // ring form: cut[(b * n_regions + r) * 2 + side] = first granule that did not fit
    unsigned long long* n_long; uint32_t* long_list;     // reads for the generic kernel
    // HOT classes (k_hot_select): labels that already hold so many reads that they would overflow their region's bins.  A read
    // with such a label is counted in LDS and never enters the stream; the counts go to the table when the block ends.
    const unsigned long long* hot;                 // one block: kHotSlots bucket hashes (0 = empty; indexed by hot_index(h) + short
                                                   // probing), kHotSlots x (arena granule of the label, table slot), the number of
                                                   // hot classes (0: the block skips all of this)
    const uint32_t* arena; uint64_t* table;
    unsigned long long* n_hot_reads;               // statistics: reads counted here
    uint32_t mix_mode;                             // bucket hash of long labels: kMixSampled / kMixFull (xxh64_device.h)
    // A launch handles the regions [grp_lo, grp_lo + grp_n) only: its LDS cursors and its bins are laid out for grp_n regions, and
    // a read whose label lives elsewhere is skipped altogether (not stored, not counted hot, not spilled) -- another launch over the
    // same reads takes it.  One group = the whole table up to kGroupRegions regions; larger tables (more than ~8 M classes) are
    // built in several passes over the sub-batch instead of falling back to the generic kernel (round 4).
    uint32_t grp_lo, grp_n;
    // Two tile sizes (direct form, a launch of two blocks per CU): blocks [0, half) take `tile` reads, the others `tile_lo` (0: all
    // take `tile`).  The block dispatched to a CU FIRST runs ~15 % faster than the one that joins it (its wavefronts are the older
    // ones at every issue slot), so equal tiles leave the second half of the launch running alone at the end.
    uint32_t half, tile_lo;
    // SHARED form (round 6): the blocks of one XCD share ONE bin per region -- bin (xcc, r) of `cap` granules at granule (xcc * grp_n + r) * cap,
    // front cursor gcur[xcc * grp_n + r] (a block reserves the granules of a whole step with one atomic), first granule that did not fit gcut[..]
    uint32_t* gcur; uint32_t* gcut;
};

constexpr uint32_t kCountedBit = 0x80000000u;          // in H: the label is followed by a granule [count, 0, 0, 0]
constexpr uint32_t kHotSlots = 1024;
constexpr uint32_t kHotProbes = 4;
__device__ __forceinline__ uint32_t hot_index(uint64_t h, uint32_t slots) { return (uint32_t)(h >> 40) & (slots - 1u); }   // (bits the region / slot / tag do not use alone)

// classes of the table that hold >= thr reads -> hot table of `slots` entries (a power of two; first come per slot; the table
// is rebuilt before every route pass: growth moves the slots).  One thread per table slot.
__global__ void k_hot_select(const uint64_t* __restrict__ table, uint64_t n_slots, unsigned long long thr, const uint32_t* __restrict__ arena,
                             unsigned long long* hot_h, uint2* hot_meta, unsigned int* n_hot, uint32_t slots, uint32_t mix_mode) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const uint64_t w = table[2 * s];
    if (w == kEmpty || table[2 * s + 1] < thr) return;
    const uint32_t rep = (uint32_t)w;
    if (!(rep & kArenaBit)) return;                                   // (not committed yet: cannot happen between sub-batches)
    const uint32_t* e = arena + 4ull * (rep & ~kArenaBit);
    const uint32_t len = e[0];
    if (len == 0 || len > kMaxPartLabel) return;
    uint32_t tmp[kHead];
    const uint64_t h = label_mix64([&](uint32_t k) { return e[1 + k]; }, len, tmp, mix_mode);
    if (h == 0) return;
    for (uint32_t p = 0; p < kHotProbes; ++p) {                        // short linear probing; a class that finds no place is not hot
        const uint32_t idx = (hot_index(h, slots) + p) & (slots - 1u);
        if (atomicCAS(&hot_h[idx], 0ull, (unsigned long long)h) == 0ull) { hot_meta[idx] = make_uint2(rep & ~kArenaBit, (uint32_t)s); atomicAdd(n_hot, 1u); break; }
    }
}


// A wavefront takes 64 CONSECUTIVE reads per step: their ids are one contiguous range, which it copies into its own LDS
// buffer with coalesced, 16-byte-aligned loads (lane-per-label loads from global memory cost the texture addresser one
// cycle per lane and dword: measured TA-bound, profiles/r2_class_build_notes.md); the lanes then pick their labels out
// of LDS.  The next step's offsets and ids are requested before this step's labels are hashed.
// FORM 0 (direct): the form of round 2 -- any number of regions, two blocks per CU, every granule stored where it belongs.
// FORM 1 (ring, round 3): bins written through LDS rings of two 64-byte units per region; one block per CU; off by default.
// FORM 2 (quad, round 4): since the compact stream format 92 % of the labels are ONE granule.  Those take the FRONT of their bin
//   four at a time: a region has a three-granule mailbox in LDS and a ticket (the next front position allowed to act); the
//   labels at positions 4u .. 4u + 2 leave their granule in the mailbox, the one at 4u + 3 takes all three out and writes the
//   64-byte unit -- four stores of one lane to one aligned 64-byte segment, which the L2 hands on as ONE write.  What bounds the
//   direct form is the number of partial writes leaving the L2 (3.5 of its 8.4 ms, profiles/r4_class_build_notes.md), and a
//   64-byte write costs the memory side what a 16-byte one does (tools/probes/scatter_write_probe.hip).  Labels of several granules
//   and runs are stored directly from the BACK of the bin (pass 2 streams both segments, as for the ring form).  48 KB of
//   mailboxes for 1024 regions fit next to a second block with the ring form's smaller staging buffers and hot table.
//   MEASURED SLOWER than the direct form (8.75 vs 8.0 ms per build): the ticket makes a label wait for every earlier label of its
//   region, in whatever wavefront that is -- the wavefronts of a block no longer run independently.  Off by default, kept with its
//   parity tests (builder_stress.py switches it) as the record of the attempt.
constexpr int kFormDirect = 0, kFormRing = 1, kFormQuad = 2, kFormShared = 3;
constexpr uint32_t kSharedBins = 8;                          // XCDs of the device: bins per region in the shared form
template <int FORM>
__global__ void __launch_bounds__(kPartBlock) __attribute__((amdgpu_waves_per_eu(FORM == kFormRing ? 4 : 8, FORM == kFormRing ? 4 : 8)))
k_part_route(RouteArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef SFGPU_X_EQ_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_eq_stamp[0][0][blockIdx.x] = wall_clock64();
#endif
    constexpr bool RING = FORM == kFormRing, QUAD = FORM == kFormQuad, SHARED = FORM == kFormShared, SMALL = RING || QUAD;
    constexpr int SW = SMALL ? kRingStageWords : kStageWords;
    constexpr uint32_t HS = SMALL ? kRingHotSlots : kHotSlots;
    const uint32_t NR = a.grp_n;                   // regions of this launch (= the whole table unless it is built in groups)
    // LDS: [ring: NR x 8 granules] | per region 2 words (RING: {cursors, unit state}; else {cursor, cut}) | staging | hot table
    uint4* ring4 = reinterpret_cast<uint4*>(smem);
    unsigned int* cur = reinterpret_cast<unsigned int*>(smem + (RING ? (size_t)NR * 128u : (QUAD ? (size_t)NR * 48u : 0u)));      // direct: NR: granules taken from my bin of region r
    unsigned int* cut = cur + NR;                                                     // !RING: NR: first granule of a label that did not fit (SHARED: the step's bases)
    uint2* cbst = reinterpret_cast<uint2*>(cur);                                      // RING: NR x {cursors, unit state}; QUAD: NR x {cursors, ticket}
    uint4* mail = reinterpret_cast<uint4*>(smem);                                     // QUAD: NR x 3 granules
    const uint32_t B1 = gridDim.x, blk = blockIdx.x, cap = a.cap, tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    uint4* stage4 = reinterpret_cast<uint4*>(cut + NR) + wave * (SW / 4 + 4);   // this wavefront's staging buffer (+ slack)
    const uint32_t* stage = reinterpret_cast<const uint32_t*>(stage4);
    if constexpr (RING || QUAD) {
        for (uint32_t r = tid; r < NR; r += kPartBlock) cbst[r] = make_uint2(0u, 0u);
        for (uint32_t r = tid; r < 2u * NR; r += kPartBlock) a.cut[(size_t)blk * NR * 2u + r] = 0xFFFFFFFFu;
        __threadfence();                                                              // (a later atomicMin of another wavefront finds it)
    } else {
        for (uint32_t r = tid; r < NR; r += kPartBlock) { cur[r] = 0; cut[r] = 0xFFFFFFFFu; }
    }
    // hot classes: their bucket hashes and a counter each, behind the staging buffers
    // (the ring form has LDS for 32 bits of every hot hash -- a filter; the label compare below decides either way -- and for a
    //  list of 32 completed units per wavefront, kRingFlushList)
    using HotTag = typename std::conditional<SMALL, unsigned int, unsigned long long>::type;
    HotTag* hot_hl = reinterpret_cast<HotTag*>(reinterpret_cast<uint4*>(cut + NR) + kPartWaves * (SW / 4 + 4));
    unsigned int* hot_cnt = reinterpret_cast<unsigned int*>(hot_hl + HS);
    uint32_t* flist = hot_cnt + HS + wave * kRingFlushList;                                  // RING only
    const bool have_hot = *reinterpret_cast<const unsigned int*>(a.hot + 2 * HS) != 0u;      // (uniform)
    if (have_hot) for (uint32_t q = tid; q < HS; q += kPartBlock) { hot_hl[q] = SMALL ? (HotTag)(a.hot[q] >> 32) : (HotTag)a.hot[q]; hot_cnt[q] = 0u; }
    const bool small_tile = a.tile_lo != 0u && blk >= a.half;
    const uint32_t my_tile = small_tile ? a.tile_lo : a.tile;
    const uint64_t t0w = small_tile ? (uint64_t)a.half * a.tile + (uint64_t)(blk - a.half) * a.tile_lo : (uint64_t)blk * a.tile;
    const uint32_t t0 = t0w < a.n ? (uint32_t)t0w : a.n;
    const uint32_t t1 = (t0w + my_tile < a.n) ? (uint32_t)(t0w + my_tile) : a.n;
    const uint32_t* __restrict__ off = a.off;
    const uint32_t* __restrict__ ids = a.ids;
    __syncthreads();

    // step c of this wavefront: reads [r0, r0 + 64) with r0 = t0 + 64 (wave + 16 c)
    auto offsets = [&](uint32_t r0, uint32_t& o, uint32_t& oe) {                     // o = off[r0 + lane], oe = off[end of the step]
        const uint32_t re = (r0 + 64u < t1) ? r0 + 64u : t1;
        const uint32_t j = r0 + lane;
        o = off[j < re ? j : re]; oe = off[re];
    };
    // the staged range starts at the 16-byte boundary at or below ids + w_lo: lane i fetches 16-byte words i and i + 64
    auto fetch = [&](uint32_t w_lo, uint32_t w_hi, uint4& x0, uint4& x1) {
        const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(ids + w_lo) >> 2) & 3u);
        const uint32_t n4 = (w_hi - w_lo + mis + 3u) >> 2;
        const uint4* src = reinterpret_cast<const uint4*>(ids + w_lo - mis);
        x0 = make_uint4(0u, 0u, 0u, 0u); x1 = x0;
        // non-temporal loads: the ids are read once -- they need not displace the bins' open lines from the L2 (round 4: route
        // 10.2 -> 9.6 ms per build under rocprofv3, profiles/r4_class_build_notes.md)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* srcv = reinterpret_cast<const u32x4*>(src);
        if (lane < n4 && n4 <= (uint32_t)SW / 4) { const u32x4 v = __builtin_nontemporal_load(srcv + lane); x0 = make_uint4(v.x, v.y, v.z, v.w); }
        if (lane + 64u < n4 && n4 <= (uint32_t)SW / 4) { const u32x4 v = __builtin_nontemporal_load(srcv + lane + 64u); x1 = make_uint4(v.x, v.y, v.z, v.w); }
    };
    uint32_t r0 = t0 + 64u * wave;
    uint32_t o = 0, oe = 0, no = 0, noe = 0, nno = 0, nnoe = 0;
    uint4 x0 = make_uint4(0u, 0u, 0u, 0u), x1 = x0;
    if (r0 < t1) {
        offsets(r0, o, oe);
        if (RING && r0 + 64u * kPartWaves < t1) offsets(r0 + 64u * kPartWaves, no, noe);
        fetch(__shfl(o, 0, kWave), oe, x0, x1);
    }
    // (SHARED: block barriers inside -- every wavefront takes the same number of steps; one past the tile's end finds no reads)
    const uint32_t xcc = SHARED ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & (kSharedBins - 1u)) : 0u;
    for (uint32_t rb = t0; SHARED ? (rb < t1) : (r0 < t1); r0 += 64u * kPartWaves, rb += 64u * kPartWaves) {
        const uint32_t nr0 = r0 + 64u * kPartWaves;
        // the next step's offsets travel now (ring form, one block per CU: the offsets of the step after it -- the ids of the
        // next step are then requested with offsets that arrived a whole step ago)
        if constexpr (RING) { if (nr0 + 64u * kPartWaves < t1) offsets(nr0 + 64u * kPartWaves, nno, nnoe); }
        else { if (nr0 < t1) offsets(nr0, no, noe); }
        const uint32_t re = (r0 + 64u < t1) ? r0 + 64u : t1;
        const uint32_t w_lo = __shfl(o, 0, kWave), w_hi = oe;
        const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(ids + w_lo) >> 2) & 3u);
        const bool staged = ((w_hi - w_lo + mis + 3u) >> 2) <= (uint32_t)SW / 4;
        stage4[lane] = x0;
        if (SW / 4 > 64 && lane + 64u < (uint32_t)SW / 4 + 4u) stage4[lane + 64u] = x1;
        const uint32_t bit31_here = (x0.x | x0.y | x0.z | x0.w | x1.x | x1.y | x1.z | x1.w) & (kHeadBit | kCompactBit);      // (the step's ids as fetched: a superset of its labels' ids)
        // my label: [b, e)
        const uint32_t nxt = __shfl_down(o, 1, kWave);
        const uint32_t b = o, e = (lane == 63u || r0 + lane + 1u >= re) ? oe : nxt;
        const uint32_t len = (r0 + lane < re) ? e - b : 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (nr0 < t1) fetch(__shfl(no, 0, kWave), noe, x0, x1);                         // ... and its ids (used after this step's labels)
        const uint32_t* lab_g = ids + b;
        const uint32_t* lab_s = stage + mis + (b - w_lo);
        uint32_t w[kHead];
        uint32_t mx = 0;
        if (staged) {
#pragma unroll
            for (int q = 0; q < kHead; ++q) w[q] = lab_s[q];                            // (the buffer has 16 words of slack)
        } else {                                                                        // labels of > 8 ids on average: straight from global memory
#pragma unroll
            for (int q = 0; q < kHead; ++q) w[q] = ((uint32_t)q < len) ? lab_g[q] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kHead; ++q) { w[q] = ((uint32_t)q < len) ? w[q] : 0u; mx = mx > w[q] ? mx : w[q]; }
        const bool unfit = len > kMaxPartLabel;
        // ---- the COMPACT form (eqclass.hip, label_probe): 4 .. 9 ids below 2^24 ascending in steps of 0 .. 255 -> first id + eight
        //      8-bit steps, one granule
        const uint32_t w8 = staged ? lab_s[8] : ((len > 8u && !unfit) ? lab_g[8] : 0u);            // (the staging buffer has 16 words of slack)
        bool ok8 = w[0] < (1u << 24);                           // every step among the first 9 ids is 0 .. 255
        uint32_t dlo = 0u, dhi = 0u;
#pragma unroll
        for (int k = 1; k <= 8; ++k) {
            const uint32_t d = ((k < 8) ? w[k < 8 ? k : 7] : w8) - w[k - 1];
            const bool live = (uint32_t)k < len;
            ok8 = ok8 && (!live || d <= 255u);
            const uint32_t dm = live ? (d & 255u) : 0u;
            if (k <= 4) dlo |= dm << (8 * (k - 1)); else dhi |= dm << (8 * (k - 5));
        }
        bool compact = ok8 && len >= 4u && len <= kMaxCompactLen;
        // the second compact form: 10 .. 17 ids in steps of 0 .. 15 (sixteen nibbles).  The first eight steps are at hand as bytes
        // (squeezed into nibbles); the further ones are walked by the lanes that hold such a label (<= 8 short iterations)
        bool compact4 = ok8 && len > kMaxCompactLen && len <= kMaxCompact4Len && !unfit && ((dlo | dhi) & 0xF0F0F0F0u) == 0u;
        if (__ballot(compact4)) {
            auto squeeze = [](uint32_t x) { x = (x | (x >> 4)) & 0x00FF00FFu; return (x | (x >> 8)) & 0xFFFFu; };
            const uint32_t qlo = squeeze(dlo) | (squeeze(dhi) << 16);
            uint32_t qhi = 0u, prev = w8;
            for (uint32_t k = 9u; compact4 && k < len; ++k) {
                const uint32_t cur = staged ? lab_s[k] : lab_g[k], d = cur - prev;
                prev = cur;
                compact4 = d <= 15u;
                qhi |= (d & 15u) << (4u * (k - 9u));
            }
            if (compact4) { dlo = qlo; dhi = qhi; }
        }
        const uint32_t cflag = compact4 ? (kCompactBit | kCompact4Bit) : kCompactBit;
        compact = compact || compact4;
        // ---- bucket hash (xxh64_device.h): length, the first 8 ids, and for longer labels the last and the middle id -- no walk
        //      over the tail (13 % of the labels have one: a lane walking its tail while the others wait cost this pass half of
        //      its vector instructions)
        uint32_t ha, hb;
        label_mix_head(w, len, ha, hb);
        const uint32_t ngl = label_granules(len);                 // granules of the label in the multi-granule form (compares)
        const uint32_t ng = compact ? 1u : ngl;                   // ... and in the stream
        // granule g >= 2 (ids 4g-1 .. 4g+2, zero padded) of the label that starts at id `lb` and holds `ll` ids
        auto granule_at = [&](uint32_t lb, uint32_t ll, uint32_t g) -> uint4 {
            const uint32_t q = 4u * g - 1u;
            uint32_t v0, v1, v2, v3;
            if (staged) { const uint32_t* ls = stage + mis + (lb - w_lo); v0 = ls[q]; v1 = ls[q + 1]; v2 = ls[q + 2]; v3 = ls[q + 3]; }
            else { const uint32_t* lg = ids + lb; v0 = lg[q]; v1 = q + 1 < ll ? lg[q + 1] : 0u; v2 = q + 2 < ll ? lg[q + 2] : 0u; v3 = q + 3 < ll ? lg[q + 3] : 0u; }
            return make_uint4(v0, q + 1 < ll ? v1 : 0u, q + 2 < ll ? v2 : 0u, q + 3 < ll ? v3 : 0u);
        };
        auto granule = [&](uint32_t g) -> uint4 { return granule_at(b, len, g); };
        if (len > (uint32_t)kHead && !unfit) {
            if (staged) label_mix_tail(ha, hb, [&](uint32_t k) { return lab_s[k]; }, len, a.mix_mode);
            else label_mix_tail(ha, hb, [&](uint32_t k) { return lab_g[k]; }, len, a.mix_mode);
        }
        // ids >= 2^31 in a tail: the staged words of the whole step are checked at once (they never occur in a real index: such a
        // step -- or, unstaged, such a label -- takes the generic kernel)
        if (staged) {
            if (__ballot(bit31_here != 0u)) mx |= kHeadBit;
        } else if (len > (uint32_t)kHead && !unfit) {
            for (uint32_t q = kHead; q < len; ++q) mx |= lab_g[q];
        }
        const uint64_t h = label_mix_final(ha, hb);
        // an id >= 2^31 would collide with the label marker of the partition stream (no real transcriptome has one)
        bool generic = unfit || (mx & (kHeadBit | kCompactBit));
        const uint32_t rg = (((uint32_t)h >> kRegionBits) & a.region_mask) - a.grp_lo;      // region inside this launch's group
        const bool in_grp = rg < NR;
        const uint32_t H = (len << 24) | ((uint32_t)(h >> (64 - kTagBits)) << kRegionBits) | ((uint32_t)h & (kRegionSlots - 1));
        // ---- runs: a read whose label is the previous read's (the lane below, same step) rides with it.  Reads that arrive
        //      clustered (a position-sorted file: 64 consecutive reads hold one or two labels) would otherwise pile into a few
        //      bins per block and overflow them (measured: 50 M sorted reads 22.8 ms against 2.6 ms shuffled).  The first read
        //      of a run carries the run's length: one more granule [count, bit 31, 0, 0] behind its label, announced by bit 31 of H.
        bool dup = false;
        {
            // filter: 32 bits of the bucket hash and the length (the labels themselves decide); `generic` is a property of the
            // label here (too long, bit 31), so the previous read's need not be looked at
            const uint32_t hp = __shfl_up((uint32_t)(h >> 32), 1, kWave);
            const uint32_t lp = __shfl_up(len, 1, kWave), bp = __shfl_up(b, 1, kWave);
            if (lane != 0u && len != 0u && !generic && staged && (uint32_t)(h >> 32) == hp && len == lp) {
                const uint32_t* prev_s = stage + mis + (bp - w_lo);
                dup = true;
                for (uint32_t q = 0; dup && q < len; ++q) dup = lab_s[q] == prev_s[q];
            }
        }
        const unsigned long long dm = __ballot(dup);
        uint32_t mult = 1;
        if (dm && !dup) { const unsigned long long after = lane == 63u ? 0ull : (dm >> (lane + 1u)); mult = 1u + (uint32_t)__builtin_ctzll(~after); }
        bool counted = false;
        if (have_hot && len != 0 && !generic && !dup && in_grp) {
            uint32_t hi = hot_index(h, HS);
            const HotTag hk = SMALL ? (HotTag)(h >> 32) : (HotTag)h;
            HotTag hv = hot_hl[hi];
            for (uint32_t p = 1; p < kHotProbes && hv != 0 && hv != hk; ++p) { hi = (hi + 1) & (HS - 1); hv = hot_hl[hi]; }
            if (hv == hk && hv != 0) {
                // same bucket hash: the label itself decides (arena entry [n, id0, id1, id2][id3 .. id6] ...; a label in
                // granule form is the same from the second granule on)
                const uint4* e = reinterpret_cast<const uint4*>(a.arena) + reinterpret_cast<const uint2*>(a.hot + HS)[hi].x;
                const uint4 e0 = e[0];
                bool same = e0.x == len && e0.y == w[0] && e0.z == w[1] && e0.w == w[2];
                if (same && len > 3u) { const uint4 e1 = e[1]; same = e1.x == w[3] && e1.y == w[4] && e1.z == w[5] && e1.w == w[6]; }
                for (uint32_t g = 2; same && g < ngl; ++g) {
                    const uint4 v = granule(g), eg = e[g];
                    same = eg.x == v.x && eg.y == v.y && eg.z == v.z && eg.w == v.w;
                }
                if (same) { atomicAdd(&hot_cnt[hi], mult); counted = true; }
            }
        }
        const bool place = len != 0 && !generic && !counted && !dup && in_grp;
        const uint32_t ngx = ng + (mult > 1u ? 1u : 0u);
        auto head_granule = [&]() {
            const uint32_t hx = H | (mult > 1u ? kCountedBit : 0u);
            return compact ? make_uint4(w[0] | kHeadBit | cflag, hx, dlo, dhi) : make_uint4(w[0] | kHeadBit, hx, w[1], w[2]);
        };
        auto count_granule = [&]() { return make_uint4(mult, kCountedBit, 0u, 0u); };       // (bit 31 of .y: no id has it -- such labels take the generic kernel)
        // granule j of what my label puts into the stream: head, ids 3..6, tail granules, the run's count
        auto stream_granule = [&](uint32_t j) -> uint4 {
            if (j == 0u) return head_granule();
            if (j >= ng) return count_granule();
            if (j == 1u) return make_uint4(w[3], w[4], w[5], w[6]);
            return granule(j);
        };
        if constexpr (QUAD) {
            // ---- reservation: one granule -> the front of the bin (through the mailbox), several -> its back (direct stores).  Both
            //      cursors share one LDS word, and a lane looks before it reserves (a bin that cannot take the label is left alone:
            //      the 16-bit fields cannot run over)
            bool front_ok = false;
            uint32_t pos = 0;
            if (place) {
                const uint32_t cs = cbst[rg].x;
                if ((cs & 0xFFFFu) + (cs >> 16) + ngx > cap) generic = true;
                else if (ngx == 1u) {
                    const uint32_t old = atomicAdd(&cbst[rg].x, 1u);
                    pos = old & 0xFFFFu;
                    if (pos + (old >> 16) + 1u <= cap) front_ok = true;
                    else { atomicMin(&a.cut[((size_t)blk * NR + rg) * 2u], pos); __threadfence(); generic = true; }      // (lost a race for the last room)
                } else {
                    const uint32_t old = atomicAdd(&cbst[rg].x, ngx << 16);
                    const uint32_t bk = old >> 16;
                    if ((old & 0xFFFFu) + bk + ngx <= cap) {
                        uint4* dst = a.out + (size_t)(blk * NR + rg) * cap + (cap - bk - ngx);
                        dst[0] = head_granule();
                        if (mult > 1u) dst[ng] = count_granule();
                        if (ng > 1u) dst[1] = make_uint4(w[3], w[4], w[5], w[6]);
                        for (uint32_t g = 2; g < ng; ++g) dst[g] = granule(g);
                    } else { atomicMin(&a.cut[((size_t)blk * NR + rg) * 2u + 1u], bk); __threadfence(); generic = true; }
                }
            }
            // ---- the front: cbst[rg].y is the region's TICKET, the next front position allowed to act.  Positions act strictly in
            //      order: 4u, 4u + 1, 4u + 2 leave their granule in the mailbox, 4u + 3 takes the three out (into registers -- the
            //      mailbox is free again the moment the ticket moves on) and writes the unit.  A lane only ever waits for labels
            //      that reserved BEFORE it, in its own wavefront (they act in this very iteration) or another (it runs on its own):
            //      no deadlock.  LDS executes a wavefront's instructions in order, which is what orders granule write -> ticket and
            //      mailbox reads -> ticket; the fences are for the compiler.
            bool pend = front_ok;
            const uint4 mineg = head_granule();
            while (__ballot(pend)) {
                if (pend && __hip_atomic_load(&cbst[rg].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == pos) {
                    const uint32_t sl = pos & 3u;
                    if (sl != 3u) {
                        mail[rg * 3u + sl] = mineg;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __hip_atomic_store(&cbst[rg].y, pos + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        const uint4 g0 = mail[rg * 3u], g1 = mail[rg * 3u + 1u], g2 = mail[rg * 3u + 2u];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                        __hip_atomic_store(&cbst[rg].y, pos + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        uint4* dst = a.out + (size_t)(blk * NR + rg) * cap + (pos - 3u);
                        dst[0] = g0; dst[1] = g1; dst[2] = g2; dst[3] = mineg;
                    }
                    pend = false;
                }
            }
        } else if constexpr (SHARED) {
            // ---- one reservation per (step of the block, region): the lanes rank their labels with the LDS atomic of the direct form,
            //      thread r then takes the step's granules of region r out of the XCD's bin with ONE returning atomic, and the labels
            //      are stored at base + rank.  Neighbouring granules of a bin come from the blocks of one XCD within microseconds of
            //      each other: the L2 hands whole 64-byte units on (TCC_EA0_WRREQ 0.91 -> 0.33 per read, all of them 64-byte ones).
            //      Measured (profiles/r6_class_build_notes.md): the pass is no faster for it -- 7.45 against 7.1 ms -- the two barriers
            //      and the exposed atomic cost 0.65 ms, the stores 2.9 instead of 3.15: what the stores cost is their REQUESTS into the
            //      L2 (one per label either way), not what leaves it.  A form with one barrier per step (answers used a step later,
            //      the first granules parked in LDS meanwhile) was slower still: 8.85 ms.
            uint32_t rank = 0;
            if (place) rank = atomicAdd(&cur[rg], ngx);
            __syncthreads();
            if (tid < NR) {
                const uint32_t c = cur[tid];
                if (c) { cut[tid] = atomicAdd(&a.gcur[xcc * NR + tid], c); cur[tid] = 0u; }
            }
            __syncthreads();
            if (place) {
                const uint32_t at = cut[rg] + rank;
                if (at + ngx <= cap) {
                    uint4* dst = a.out + (size_t)(xcc * NR + rg) * cap + at;
#if !defined(SFGPU_X_NOSTORE)
                    dst[0] = head_granule();
                    if (mult > 1u) dst[ng] = count_granule();
                    if (ng > 1u) dst[1] = make_uint4(w[3], w[4], w[5], w[6]);
                    for (uint32_t g = 2; g < ng; ++g) dst[g] = granule(g);
#else
                    if (at == 0xFFFFFFF0u) dst[0] = head_granule();
#endif
                } else {
                    atomicMin(&a.gcut[xcc * NR + rg], at);                                  // the bin ends before this label
                    generic = true;
                }
            }
        } else if constexpr (!RING) {
#if defined(SFGPU_X_PAIR)            // experiment (WRONG bins, timing only): the lanes of a group of SFGPU_X_PAIR store their first granules behind the
                                     // group's first lane's -- runs of that many granules at the alignment a bin's cursor happens to have: what a
                                     // block-local multisplit could at best make of the stores (requests into the L2 per label: 1 / SFGPU_X_PAIR)
            unsigned long long d64 = 0ull;
            if (place) {
                const uint32_t at = atomicAdd(&cur[rg], ngx);
                if (at + ngx <= cap) d64 = reinterpret_cast<unsigned long long>(a.out + (size_t)(blk * NR + rg) * cap + at);
                else { atomicMin(&cut[rg], at); generic = true; }
            }
            const unsigned long long l64 = __shfl(d64, (int)(lane & ~(uint32_t)(SFGPU_X_PAIR - 1)), kWave);
            if (d64 && l64) reinterpret_cast<uint4*>(l64)[lane & (uint32_t)(SFGPU_X_PAIR - 1)] = head_granule();
            if (false) {
#else
            if (place) {
#endif
                const uint32_t at = atomicAdd(&cur[rg], ngx);                               // my granules in the bin (rg, blk)
                if (at + ngx <= cap) {
#if defined(SFGPU_X_FOLD)            // experiment: every store lands in 1 MB (L2-resident): what do the stores cost WITHOUT the memory behind the L2?
                    uint4* dst = a.out + (((size_t)(blk * NR + rg) * cap + at) & 0xFFFFu);
#else
                    uint4* dst = a.out + (size_t)(blk * NR + rg) * cap + at;
#endif
#if !defined(SFGPU_X_NOSTORE)        // experiment: no stores at all
                    dst[0] = head_granule();
                    if (mult > 1u) dst[ng] = count_granule();
                    if (ng > 1u) dst[1] = make_uint4(w[3], w[4], w[5], w[6]);
                    for (uint32_t g = 2; g < ng; ++g) dst[g] = granule(g);
#else
                    if (at == 0xFFFFFFF0u) dst[0] = head_granule();
#endif
                } else {
                    atomicMin(&cut[rg], at);                                                // the bin ends before this label
                    generic = true;
                }
            }
        } else {
            // ---- reservation: front (through the ring) or back (direct stores) of the bin (rg, blk).  The back takes the labels of
            //      more than 4 granules and the labels that would have to WAIT for a ring slot: a unit that is complete but not yet
            //      released by its completer (a window of a thousand cycles or so) is closed to the lane that needs its slot, and
            //      1.8 % of the labels met one -- every other wavefront step then went through the retry loop below.  The lane looks
            //      before it reserves: if the units it would get at the cursor's present position are not open it stores its label
            //      at the back instead (scattered 16-byte stores, for those few labels).  Only a lane that loses a race between the
            //      look and the reservation still has to wait.
            bool front = false;
            uint32_t at = 0;
            uint32_t stv = 0;
            auto is_open = [&](uint32_t v, uint32_t sv) { return ((v >> 1) & 0xFFFu) == ((sv >> (16u * (v & 1u) + 3u)) & 0xFFFu); };
            bool back_ok = false;
            uint32_t back_at = 0;                                                             // granule index of my label in a.out
            if (place) {
                const uint2 cs = cbst[rg];                                                   // (a stale unit state is safe: it can only say "closed")
                stv = cs.y;
                const uint32_t pf = cs.x & 0xFFFFu;
                if (pf + (cs.x >> 16) + ngx > cap) generic = true;                           // does not fit, and never will: nothing reserved
                else if (ngx <= kRingFrontGranules && is_open(pf >> 2, stv) && is_open((pf + ngx - 1u) >> 2, stv)) {
                    const uint32_t old = atomicAdd(&cbst[rg].x, ngx);
                    at = old & 0xFFFFu;
                    if (at + ngx + (old >> 16) <= cap) front = true;
                    else {                                                                   // lost a race for the last room: the front ends before this label
                        atomicMin(&a.cut[((size_t)blk * NR + rg) * 2u], at); __threadfence();
                        generic = true;
                    }
                }
                else {
                    const uint32_t old = atomicAdd(&cbst[rg].x, ngx << 16);
                    const uint32_t bk = old >> 16;
                    if ((old & 0xFFFFu) + bk + ngx <= cap) {
                        back_ok = true; back_at = (blk * NR + rg) * cap + (cap - bk - ngx);
                        uint4* dst = a.out + (size_t)back_at;
                        dst[0] = head_granule();
                        if (mult > 1u) dst[ng] = count_granule();
                        if (ng > 1u) dst[1] = make_uint4(w[3], w[4], w[5], w[6]);
                    } else {
                        atomicMin(&a.cut[((size_t)blk * NR + rg) * 2u + 1u], bk); __threadfence();
                        generic = true;
                    }
                }
            }
            // the tail granules of the labels at the back: the wavefront copies them together, label by label -- lane i takes granule
            // 2 + i (a lane walking its own label's tail kept the other 63 waiting: 72 of this kernel's 535 vector instructions per step)
            for (unsigned long long bm = __ballot(back_ok && ng > 2u); bm; bm &= bm - 1ull) {
                const int src = __builtin_ctzll(bm);
                const uint32_t o_at = __shfl(back_at, src, kWave), o_b = __shfl(b, src, kWave), o_len = __shfl(len, src, kWave);
                const uint32_t o_ng = label_granules(o_len);
                for (uint32_t g = 2u + lane; g < o_ng; g += 64u) a.out[(size_t)o_at + g] = granule_at(o_b, o_len, g);
            }
            // ---- the ring.  A label touches at most two units: u0 (its first n0 granules) and u0 + 1 (the other n1)
            const uint32_t u0 = at >> 2, q0 = u0 & 1u;
            const uint32_t n0 = (4u - (at & 3u)) < ngx ? (4u - (at & 3u)) : ngx, n1 = ngx - n0;
            bool c0 = false, c1 = false;                                                      // completed: unit u0 / unit u0 + 1
            // The units a step completed are written out by the wavefront together: four lanes per unit -- one whole 64-byte segment
            // per four lanes -- 16 units per store instruction, listed kRingFlushList at a time; then the completers release the ring
            // slots: count back to 0, one more unit of that parity flushed.  (A completer storing its own unit's four granules one
            // after the other was measured: 10.2 instead of 7.9 ms per build -- 1.6 scattered 16-byte requests per label again.  LDS
            // executes a wavefront's instructions in order: the granules have been read when the release is performed.)
            auto flush = [&]() {
                const unsigned long long m0 = __ballot(c0), m1 = __ballot(c1);
                if (!(m0 | m1)) return;
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t k_n0 = (uint32_t)__builtin_popcountll(m0), K = k_n0 + (uint32_t)__builtin_popcountll(m1);
                const uint32_t r0 = (uint32_t)__builtin_popcountll(m0 & below), r1 = k_n0 + (uint32_t)__builtin_popcountll(m1 & below);
                for (uint32_t base = 0; base < K; base += kRingFlushList) {
                    if (c0 && r0 - base < kRingFlushList) flist[r0 - base] = rg | (u0 << 10);
                    if (c1 && r1 - base < kRingFlushList) flist[r1 - base] = rg | ((u0 + 1u) << 10);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const uint32_t n_here = (K - base < kRingFlushList) ? (K - base) : kRingFlushList;
                    // the units' granules go to registers first; as soon as the last round's reads are in the LDS queue the ring
                    // slots are released (in order behind them) -- the stores to memory follow: a unit that is complete but not
                    // yet released keeps every lane that needs its slot waiting
                    static_assert(kRingFlushList == 32, "two store rounds of 16 units per list");
                    const uint32_t ka = lane >> 2, kb = 16u + (lane >> 2);
                    uint4 da = make_uint4(0u, 0u, 0u, 0u), db = da;
                    uint32_t wa = 0xFFFFFFFFu, wb = 0xFFFFFFFFu;
                    if (ka < n_here) {
                        const uint32_t e = flist[ka], r = e & 1023u, v = e >> 10;
                        da = ring4[r * 8u + ((v & 1u) << 2) + (lane & 3u)];
                        wa = (blk * NR + r) * cap + 4u * v + (lane & 3u);
                    }
                    if (kb < n_here) {
                        const uint32_t e = flist[kb], r = e & 1023u, v = e >> 10;
                        db = ring4[r * 8u + ((v & 1u) << 2) + (lane & 3u)];
                        wb = (blk * NR + r) * cap + 4u * v + (lane & 3u);
                    }
                    if (base + kRingFlushList >= K) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        if (c0 | c1) atomicAdd(&cbst[rg].y, (c0 ? (4u << (16u * q0)) : 0u) | (c1 ? (4u << (16u * (q0 ^ 1u))) : 0u));
                    }
                    {
                        if (wa != 0xFFFFFFFFu) a.out[(size_t)wa] = da;
                        if (wb != 0xFFFFFFFFu) a.out[(size_t)wb] = db;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            };
            // the usual step: every unit a lane needs is open -- write, count, flush
            bool more = front;
            const bool fast = front && is_open(u0, stv) && (n1 == 0u || is_open(u0 + 1u, stv));
            if (fast) {
                ring4[rg * 8u + (at & 7u)] = head_granule();
                if (ngx > 1u) ring4[rg * 8u + ((at + 1u) & 7u)] = (ng > 1u) ? make_uint4(w[3], w[4], w[5], w[6]) : count_granule();
                if (mult > 1u && ng > 1u) ring4[rg * 8u + ((at + ng) & 7u)] = count_granule();
                more = false;
            }
            if (__ballot(fast && ng > 2u)) {
                if (fast && ng > 2u) ring4[rg * 8u + ((at + 2u) & 7u)] = granule(2);
                if (fast && ng > 3u) ring4[rg * 8u + ((at + 3u) & 7u)] = granule(3);
            }
            if (fast) {
                const uint32_t add = (n0 << (16u * q0)) | (n1 << (16u * (q0 ^ 1u)));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const uint32_t nw = atomicAdd(&cbst[rg].y, add) + add;
                c0 = ((nw >> (16u * q0)) & 7u) == 4u;
                c1 = n1 != 0u && ((nw >> (16u * (q0 ^ 1u))) & 7u) == 4u;
            }
            flush();
            // the rare lane: a unit it needs is closed (the ring is two units deep; a completer of another wavefront is between its
            // count and its release, or three labels of one region are in flight at once).  Granule by granule until it is through;
            // a lane never holds a completed unit while it waits, so the completers it waits for always get to flush.
            uint32_t wr = 0;
            while (__ballot(more)) {
                c0 = false; c1 = false;
                // wait (cheaply: a read and a test per poll) until the next granule of SOME waiting lane can be written
                for (;;) {
                    bool can = false;
                    if (more) { stv = __hip_atomic_load(&cbst[rg].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); can = is_open((at + wr) >> 2, stv); }
                    if (__ballot(can)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (more) {
                    uint32_t add = 0;
                    bool open = true;
                    for (uint32_t j = wr; open && j < ngx; ++j) {
                        const uint32_t p = at + j, v = p >> 2;
                        if (is_open(v, stv)) { ring4[rg * 8u + (p & 7u)] = stream_granule(j); add += 1u << (16u * (v & 1u)); wr = j + 1u; }
                        else open = false;
                    }
                    if (add) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        const uint32_t nw = atomicAdd(&cbst[rg].y, add) + add;
                        const uint32_t a0 = (add >> (16u * q0)) & 7u, a1 = (add >> (16u * (q0 ^ 1u))) & 7u;
                        c0 = a0 != 0u && ((nw >> (16u * q0)) & 7u) == 4u;
                        c1 = a1 != 0u && ((nw >> (16u * (q0 ^ 1u))) & 7u) == 4u;
                    }
                    more = wr < ngx;
                }
                flush();
            }
        }
        {
            // one cursor update per wavefront, not per read: a label that holds a large part of the reads (a highly expressed
            // gene) overflows its bins read after read, and millions of returning atomics on ONE address serialise
            // (measured: 10 % of 50 M reads on one label, 76 ms for a 2.4 ms build)
            bool spill = len != 0 && generic && in_grp;
            if (dm) {                                                                   // the reads of a run follow its first read
                const unsigned long long lead = ~dm & ((2ull << lane) - 1ull);           // (lane 63: 2 << 63 wraps to 0, - 1 = all ones)
                const int ll = 63 - __builtin_clzll(lead | 1ull);
                const bool sl = __shfl((int)spill, ll, kWave) != 0;
                if (dup) spill = sl;
            }
            const unsigned long long sm = __ballot(spill);
            if (sm) {
                unsigned long long base_l = 0;
                if (lane == (uint32_t)__builtin_ctzll(sm)) base_l = atomicAdd(a.n_long, (unsigned long long)__builtin_popcountll(sm));
                base_l = __shfl(base_l, __builtin_ctzll(sm), kWave);
                if (spill) a.long_list[base_l + __builtin_popcountll(sm & ((1ull << lane) - 1ull))] = a.first + r0 + lane;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                      // every lane is done with the staged ids before they are replaced
        o = no; oe = noe;
        if constexpr (RING) { no = nno; noe = nnoe; }
    }
    __syncthreads();
    if constexpr (QUAD) {
        // what the mailboxes still hold: the last, partial unit of every bin (1 .. 3 granules)
        for (uint32_t r = tid; r < NR; r += kPartBlock) {
            const uint32_t c = cbst[r].x;
            const uint32_t xf = a.cut[((size_t)blk * NR + r) * 2u], xb = a.cut[((size_t)blk * NR + r) * 2u + 1u];
            const uint32_t f = (c & 0xFFFFu) < xf ? (c & 0xFFFFu) : xf, bk = (c >> 16) < xb ? (c >> 16) : xb;
            for (uint32_t j = 0; j < (f & 3u); ++j) a.out[(size_t)(blk * NR + r) * cap + (f & ~3u) + j] = mail[r * 3u + j];
            a.fill[r * B1 + blk] = f; a.fill_back[r * B1 + blk] = bk;
        }
    } else if constexpr (SHARED) {
        // (the bins' fills are the XCDs' cursors: k_shared_fill, after the launch)
    } else if constexpr (!RING) {
        for (uint32_t r = tid; r < NR; r += kPartBlock) { const unsigned int c = cur[r], x = cut[r]; a.fill[r * B1 + blk] = c < x ? c : x; a.fill_back[r * B1 + blk] = 0u; }
    } else {
        // what the rings still hold: the last, partial unit of every bin
        for (uint32_t r = tid; r < NR; r += kPartBlock) {
            const uint32_t c = cbst[r].x;
            const uint32_t xf = a.cut[((size_t)blk * NR + r) * 2u], xb = a.cut[((size_t)blk * NR + r) * 2u + 1u];
            const uint32_t f = (c & 0xFFFFu) < xf ? (c & 0xFFFFu) : xf, bk = (c >> 16) < xb ? (c >> 16) : xb;
            const uint32_t v = f >> 2;
            for (uint32_t j = 0; j < (f & 3u); ++j) a.out[(size_t)(blk * NR + r) * cap + 4u * v + j] = ring4[r * 8u + ((v & 1u) << 2) + j];
            a.fill[r * B1 + blk] = f; a.fill_back[r * B1 + blk] = bk;
        }
    }
    if (have_hot) {
        unsigned int tot = 0;
        for (uint32_t q = tid; q < HS; q += kPartBlock) {
            const unsigned int c = hot_cnt[q];
            if (c) atomicAdd(reinterpret_cast<unsigned long long*>(&a.table[2ull * reinterpret_cast<const uint2*>(a.hot + HS)[q].y + 1]), (unsigned long long)c);
            tot += c;
        }
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, kWave);
        if (lane == 0 && tot) atomicAdd(a.n_hot_reads, (unsigned long long)tot);
    }
#ifdef SFGPU_X_EQ_STAMP
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_eq_stamp[0][1][blockIdx.x] = wall_clock64();
#endif
}

// shared form: cursors of the XCDs' bins before a launch, fills after it (pass 2 sees kSharedBins "blocks")
__global__ void k_shared_begin(uint32_t* gcur, uint32_t* gcut, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { gcur[i] = 0u; gcut[i] = 0xFFFFFFFFu; }
}
__global__ void k_shared_fill(const uint32_t* __restrict__ gcur, const uint32_t* __restrict__ gcut, uint32_t n_regions, uint32_t* fill, uint32_t* fill_back) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;                // = x * n_regions + r
    if (i >= n_regions * kSharedBins) return;
    const uint32_t x = i / n_regions, r = i - x * n_regions;
    const uint32_t c = gcur[i], m = gcut[i];
    fill[r * kSharedBins + x] = c < m ? c : m; fill_back[r * kSharedBins + x] = 0u;
}

struct PartArgs {
    uint64_t* table;                       // {word, count} pairs
    const uint4* bins;                     // granules
    const uint32_t* fill; const uint32_t* fill_back; uint32_t n_blocks; uint32_t cap;     // region r: bins (r, 0 .. n_blocks) of cap granules each: fill from the
                                                               // front, fill_back from the back (the ring form of pass 1 puts long labels there)
    uint64_t* cls_hash; uint64_t* cls_off; uint32_t* cls_len; uint32_t* cls_slot; uint32_t* arena;
    unsigned long long* ctr;               // this sub-batch's counters: CTR_NEW (classes it created), CTR_DEFER
    unsigned long long* arena_ctr;         // the builder's arena cursor (words)
    unsigned long long* gcls;              // the builder's class counter: class ids come from here, so a launch need not know how
                                           // many classes the launch before it created (the host may still be a sub-batch behind)
    uint32_t* deferred;                    // (granule index of the label in the bins, length) of labels that found their region full
    uint32_t mix_mode;                     // bucket hash of long labels (the full tag of a new class is computed at commit)
    uint32_t grp_lo;                       // block b handles table region grp_lo + b; the bins are laid out for gridDim.x regions
    uint32_t atomic_counts;                // 1: the counts go back with atomic adds (pipelined form: the next sub-batch's route pass may be adding
                                           // what it counted for hot classes to the same words); 0: plain read-modify-write (nothing else runs)
    uint32_t* probe;                       // slot-indexed copies of the classes' probe granules (4 words per table slot): a region's probes are
                                           // one coalesced 64 KB read (the copies in the arena, found through rep, were 1.6 M scattered reads per launch)
    uint32_t shared;                       // 1: the bins are the XCDs' (shared form of pass 1): n_blocks = kSharedBins long segments per region, which
                                           // the wavefronts of the block take in 16 equal slices (a label belongs to the slice that holds its first granule)
};

// ---- pass 2: one block per region
// A slot word is tag(32) | rep(32) in the table.  In LDS the region is kept COMPACT: `slot32[s]` = tag(12) | length(7) |
// local class index(13) (or empty), and per local class its label's FIRST GRANULE with the representative's address in
// place of the length -- `chead[idx]` = (rep, id0, id1, id2) -- plus a count delta.  A label of <= 3 ids (58 % of the reads)
// is therefore probed, compared and counted without leaving the CU; a longer one fetches only its further granules.  The
// scattered 16-byte compare loads, not the stream, bounded this pass before (texture addresser ~3.5 cycles per lane and load,
// profiles/r2_class_build_notes.md); a slot-indexed copy of the heads (64 KB) would not leave room for two blocks per CU,
// the class-indexed one does: 16 KB slots + 2304 x (16 + 4) B classes + 17 KB wavefront tiles = 79 KB.
// rep = kArenaBit | arena granule for classes committed before this launch, else the granule index of the label in the bins
// (a class CREATED by this launch; it is committed -- arena entry, XXH64, full bucket hash, table word -- after the stream).
// Wavefront w streams the bins w, w + 16, ... of the region, 64 granules at a time: ONE coalesced 16-byte load per lane;
// a lane whose granule starts a label handles that label (ids 0..2 are in its registers, ids 3..6 in its neighbour's
// granule, read back from the wavefront's LDS copy of the step).  The only block-wide synchronisation is before and
// after the streaming loop.
constexpr uint32_t kMaxRegionClasses = 2400;             // local classes per region (mean <= 2048 at the table load limit of 1/2: + 7.8 sigma); what 80 KB leave
constexpr uint32_t kSlotEmpty = 0xFFFFFFFFu;
constexpr uint32_t kSlotKeyMask = 0xFFFFE000u;            // tag(12) << 20 | length(7) << 13
constexpr uint32_t kSlotIdxMask = 0x1FFFu;
constexpr uint32_t kDeadRep = 0xFFFFFFFFu;                // a class index that lost the race for its slot

__global__ void __launch_bounds__(kPartBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_part_insert(PartArgs a) {
#ifdef SFGPU_X_EQ_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_eq_stamp[1][0][blockIdx.x] = wall_clock64();
    struct StampEnd { __device__ ~StampEnd() { if (threadIdx.x == 0 && blockIdx.x < 4096) g_eq_stamp[1][1][blockIdx.x] = wall_clock64(); } } stamp_end;
#endif
#ifdef SFGPU_X_INS_DELAY                 // experiment: the second block of every CU starts SFGPU_X_INS_DELAY us late (is the first round slow because both start in step?)
    if (blockIdx.x >= 256u && blockIdx.x < 512u) { const unsigned long long t0d = wall_clock64(); while (wall_clock64() - t0d < (unsigned long long)(SFGPU_X_INS_DELAY) * 100ull) __builtin_amdgcn_s_sleep(8); }
#endif
    __shared__ unsigned int slot32[kRegionSlots];
    __shared__ __attribute__((aligned(16))) uint4 chead[kMaxRegionClasses];
    __shared__ unsigned int ccnt[kMaxRegionClasses];
    // long labels waiting for their verification: (granule index in the bins, class | length | run), 64 per wavefront; after the
    // stream the same words list the slots of the classes this launch created
    __shared__ __attribute__((aligned(8))) uint32_t vq_words[kMaxRegionClasses];
    static_assert(kMaxRegionClasses >= 2u * 64u * kPartWaves, "the verification queues fit the new-class list");
    uint2 (*vq)[64] = reinterpret_cast<uint2 (*)[64]>(vq_words);
    __shared__ unsigned int s_occ, s_ncls, s_nold, s_nnew, s_newwords;
    __shared__ unsigned long long s_arena0, s_cid0;
    static_assert(kRegionBits == 12, "slot32 packs a 12-bit tag, a 7-bit length and a 13-bit class index");
    const uint32_t region = blockIdx.x;                                   // (inside the group: bins and fills are indexed by it)
    const uint64_t rb = (uint64_t)(a.grp_lo + region) * kRegionSlots;       // the region's slots in the table
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // this wavefront's bins: lane t holds the fills of bin wave + 16 t.  A bin is two SEGMENTS of whole labels: [0, fill) and
    // [cap - fill_back, cap); segment t < 64 is the front of bin t, segment 64 + t its back
    const bool sh = a.shared != 0u;
    const uint32_t my_bin = sh ? lane : wave + kPartWaves * lane;
    const uint32_t my_fill = (my_bin < a.n_blocks) ? a.fill[region * a.n_blocks + my_bin] : 0u;
    const uint32_t my_back = (my_bin < a.n_blocks) ? a.fill_back[region * a.n_blocks + my_bin] : 0u;

    if (threadIdx.x == 0) { s_occ = 0; s_ncls = 0; s_nnew = 0; s_newwords = 0; }
    for (uint32_t c = threadIdx.x; c < kMaxRegionClasses; c += kPartBlock) ccnt[c] = 0;
    __syncthreads();
    // ---- the region's committed classes: slot -> local class, label head from the arena (one scattered 16-byte load per
    //      CLASS and launch instead of one per READ)
    bool overfull = false;
    for (uint32_t s = threadIdx.x; s < kRegionSlots; s += kPartBlock) {
        const unsigned long long w = a.table[2 * (rb + s)];
        uint32_t e = kSlotEmpty;
        if (w != kEmpty) {
            const uint32_t idx = atomicAdd(&s_ncls, 1u);
            if (idx < kMaxRegionClasses) {
                const uint32_t rep = (uint32_t)w;
                const uint4 h = reinterpret_cast<const uint4*>(a.probe)[rb + s];                    // the class's probe granule [n, p0, p1, p2]
                chead[idx] = make_uint4(rep, h.y, h.z, h.w);
                e = ((uint32_t)(w >> 52) << 20) | ((h.x & 0x7Fu) << 13) | idx;
            } else overfull = true;                          // (a table loaded beyond 1/2 by the generic kernel: see below)
        }
        slot32[s] = e;
    }
    __syncthreads();
    if (threadIdx.x == 0) { s_nold = s_ncls < kMaxRegionClasses ? s_ncls : kMaxRegionClasses; s_occ = s_ncls; }
    const bool region_overfull = __syncthreads_or(overfull);
#ifdef SFGPU_X_INS_PROLOGUE_ONLY       // experiment: what does a launch cost before it streams anything?
    if (a.cap != 0xFFFFFFFFu) return;
#endif
    const uint32_t n_old = s_nold;

    // ---- the wavefront's bins as ONE flat stream of granules.  A wavefront owns the 128 segments {front, back} x {bin wave + 16 t}
    //      of its region; streaming them one after the other in steps of 64 granules left the last step of every segment part
    //      empty -- and a sub-batch of 67 M reads puts only ~130 granules into a segment, one of 4 M reads 8: the pass took 161 us
    //      per LAUNCH whatever it held (profiles/r4_class_build_notes.md).  Now lane i of a step takes granule gpos + i of the
    //      concatenation of all segments: the segments' exclusive prefix sums sit in the lanes (pf: fronts, pb: backs), a lane
    //      finds its segment with a 6-step binary search over them by shuffle.  Labels never span segments, so everything
    //      downstream (a label's granules in consecutive lanes, the label at lane 0 whole) holds as before.
    uint32_t tot_f = 0, tot_b = 0;
    auto excl_scan = [&](uint32_t v, uint32_t& total) -> uint32_t {
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, kWave); if ((int)lane >= o) x += y; }
        total = __shfl(x, 63, kWave);
        return x - v;
    };
    const uint32_t pf = excl_scan(my_fill, tot_f), pb = excl_scan(my_back, tot_b);
    const uint32_t T = tot_f + tot_b;                      // granules this wavefront streams
    // flat index q (clamped below T by the caller; every lane calls: shuffles inside) -> granule index in a.bins
    auto locate = [&](uint32_t q) -> uint32_t {
        const bool back = q >= tot_f;
        const uint32_t qq = back ? q - tot_f : q;
        uint32_t t = 0;
#pragma unroll
        for (uint32_t step = 32u; step; step >>= 1) {
            const uint32_t c = t + step;                   // (<= 63)
            const uint32_t vf = __shfl(pf, (int)c, kWave), vb = __shfl(pb, (int)c, kWave);
            if ((back ? vb : vf) <= qq) t = c;
        }
        const uint32_t p0f = __shfl(pf, (int)t, kWave), p0b = __shfl(pb, (int)t, kWave), fb = __shfl(my_back, (int)t, kWave);
        const uint32_t bin0 = ((sh ? t : wave + kPartWaves * t) * gridDim.x + region) * a.cap;
        return back ? bin0 + a.cap - fb + (qq - p0b) : bin0 + (qq - p0f);
    };
    // One label through the region's LDS image.  s = its home slot, key = tag | length, (w0, w1, w2) = the payload of its first
    // granule, `here` = where that granule sits in the bins.  A label of <= 3 ids or in compact form is decided by those words.
    // A longer one has further granules: with me == nullptr (the streaming loop) the first class whose key and head agree is
    // returned as a CANDIDATE (-> true, cand = its index) and verified later, 64 labels at a time (verify_batch); with me != nullptr
    // (a label whose candidate failed) every such class is compared granule by granule -- the label's granules from the bins
    // (me), the class's from the arena or, for a class this launch created, from the bins.
    auto probe = [&](uint32_t s, const uint32_t key, const uint32_t len, const uint32_t ng, const bool whole, const uint32_t w0, const uint32_t w1,
                     const uint32_t w2, const uint32_t mult, const uint32_t here, const uint32_t multi, const uint4* me, uint32_t& cand) -> bool {
        auto defer = [&]() { const unsigned long long d = atomicAdd(&a.ctr[CTR_DEFER], 1ull); a.deferred[2 * d] = here; a.deferred[2 * d + 1] = len | (multi << 31); };
        if (region_overfull) { defer(); return false; }
        uint32_t probes = 0;
        while (probes <= kRegionSlots) {
            uint32_t e = slot32[s];
            while (e != kSlotEmpty && (e & kSlotKeyMask) != key && probes < kRegionSlots) {
                s = (s + 1) & (kRegionSlots - 1); ++probes; e = slot32[s];
            }
            if (probes >= kRegionSlots) { defer(); return false; }                  // cannot place
            if (e == kSlotEmpty) {
                // claim: the class's head is written BEFORE the slot names it, so a prober that sees the slot reads a whole head
                const uint32_t idx = atomicAdd(&s_ncls, 1u);
                if (idx >= kMaxRegionClasses || atomicAdd(&s_occ, 1u) >= kRegionLimit) {      // region full: defer
                    if (idx < kMaxRegionClasses) { atomicSub(&s_occ, 1u); chead[idx].x = kDeadRep; }
                    defer(); return false;
                }
                chead[idx] = make_uint4(here, w0, w1, w2);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                const uint32_t old = atomicCAS(&slot32[s], kSlotEmpty, key | idx);
                if (old == kSlotEmpty) { atomicAdd(&ccnt[idx], mult); atomicAdd(&s_nnew, 1u); return false; }
                chead[idx].x = kDeadRep;                 // lost the race: this index stays unused
                atomicSub(&s_occ, 1u);
                e = old;
                if ((e & kSlotKeyMask) != key) { s = (s + 1) & (kRegionSlots - 1); ++probes; continue; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t idx = e & kSlotIdxMask;
            const uint4 hd = chead[idx];
            bool same = hd.y == w0 && hd.z == w1 && hd.w == w2;
#ifdef SFGPU_X_NOPHASE2            // experiment: the head granule decides (WRONG for long labels that share one: timing only)
            if (false) {
#else
            if (same && !whole) {                        // (a compact granule, or <= 3 ids, IS the label: equal words, equal labels)
#endif
                if (!me) { cand = idx; return true; }
                const uint4* r = (hd.x & kArenaBit) ? reinterpret_cast<const uint4*>(a.arena) + (hd.x & ~kArenaBit) : a.bins + hd.x;
                for (uint32_t j = 1; same && j < ng; ++j) {
                    const uint4 ej = r[j], tj = me[j];
                    same = ej.x == tj.x && ej.y == tj.y && ej.z == tj.z && ej.w == tj.w;
                }
            }
            if (same) { atomicAdd(&ccnt[idx], mult); return false; }
            s = (s + 1) & (kRegionSlots - 1); ++probes;
        }
        return false;
    };
    // ---- long labels (more granules than one, not compact: 0.75 % of the benchmark's) are VERIFIED IN BATCHES.  Comparing a label's
    //      further granules means loads from the arena, and when that happened inside the step (rounds 2 - 3: every lane compared the
    //      granule it held) the whole wavefront sat through their round trip in every third step: 0.6 of the pass's 4.1 ms (round 4,
    //      profiles/r4_class_build_notes.md).  Now a candidate (class whose tag, length and first granule agree) only puts the
    //      label into the wavefront's queue; when 64 are waiting, lane i verifies label i -- its granules from the bins (just
    //      streamed), the class's from the arena -- and the rare label whose candidate fails is probed again with full compares.
    uint32_t qn = 0;                                       // labels waiting (wavefront-uniform)
    auto verify_batch = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < qn) {
            const uint2 e = vq[wave][lane];
            const uint32_t here = e.x, idx = e.y & kSlotIdxMask, len = (e.y >> 13) & 0x7Fu, multi = (e.y >> 20) & 1u;
            const uint32_t ng = label_granules(len);
            const uint4* me = a.bins + here;
            const uint32_t rep = chead[idx].x;
            const uint4* r = (rep & kArenaBit) ? reinterpret_cast<const uint4*>(a.arena) + (rep & ~kArenaBit) : a.bins + rep;
            bool same = true;
            for (uint32_t j = 1; same && j < ng; ++j) {
                const uint4 ej = r[j], tj = me[j];
                same = ej.x == tj.x && ej.y == tj.y && ej.z == tj.z && ej.w == tj.w;
            }
            const uint32_t mult = multi ? me[ng].x : 1u;
            if (same) atomicAdd(&ccnt[idx], mult);
            else {                                          // another label with this tag, length and first granule: probe again, comparing whole labels
                const uint4 g0 = me[0];
                const uint32_t H = g0.y;
                uint32_t dummy = 0;
                probe(H & (kRegionSlots - 1), (((H >> kRegionBits) & 0xFFFu) << 20) | (len << 13), len, ng, false, g0.x & ~kHeadBit, g0.z, g0.w, mult, here, multi, me, dummy);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        qn = 0;
    };
    // (shared bins: this wavefront's slice [lo, hi) of the region's stream -- the labels whose FIRST granule lies in it; a step may read on past hi)
    const uint32_t lo = sh ? (uint32_t)((uint64_t)T * wave / kPartWaves) : 0u, hi = sh ? (uint32_t)((uint64_t)T * (wave + 1u) / kPartWaves) : T;
    uint32_t gpos = lo;                                    // flat index of the step's first granule (wavefront-uniform)
    uint32_t here = locate(lo + lane < T ? lo + lane : (T ? T - 1u : 0u));      // where this lane's granule sits in the bins (granule index)
    uint4 g = make_uint4(0u, 0u, 0u, 0u);
    if (lo + lane < T) g = a.bins[here];
    while (gpos < hi) {
        const uint32_t cnt = (T - gpos < 64u) ? (T - gpos) : 64u;
        const bool is_head = lane < cnt && (g.x & kHeadBit) && gpos + lane < hi;
        const uint32_t H = g.y;
        const uint32_t len = (H >> 24) & 0x7Fu;
        const bool compact = (g.x & kCompactBit) != 0u;       // the whole label is in this granule (first id + 8- or 4-bit steps)
        const uint32_t ng = compact ? 1u : label_granules(len);
        const uint32_t multi = H >> 31;                        // a run of identical reads: one more granule holds its length
        // labels whose granules are not all in this step wait for the next one, which starts at the first of them (a label
        // is <= 32 granules, so the label at lane 0 is always whole; the last step holds whole labels only)
        const unsigned long long inc = __ballot(is_head && lane + ng + multi > cnt);
        uint32_t adv = inc ? (uint32_t)__builtin_ctzll(inc) : cnt;
        if (adv == 0u) adv = cnt;                            // (cannot happen with well-formed bins: never spin on a corrupt one)
        // the NEXT step's granule is requested before this step's labels are probed: its round trip hides behind theirs
        const uint32_t ngpos = gpos + adv;
        const uint32_t nq = ngpos + lane;
        const uint32_t nhere = locate(nq < T ? nq : T - 1u);
        uint4 gn = make_uint4(0u, 0u, 0u, 0u);
        if (nq < T) gn = a.bins[nhere];
        // a run's length sits in the granule behind its label: in the registers of the lane ng further on (the label is whole in
        // this step)
        uint32_t mult = 1u;
        if (__ballot(is_head && multi)) { const uint32_t cx = __shfl(g.x, (int)((lane + ng) & 63u), kWave); if (is_head && multi) mult = cx; }
        uint32_t cand = 0;
        bool queue = false;
        if (is_head && lane < adv)
            queue = probe(H & (kRegionSlots - 1), (((H >> kRegionBits) & 0xFFFu) << 20) | (len << 13), len, ng, compact || len <= 3u,
                          g.x & ~kHeadBit, g.z, g.w, mult, here, multi, nullptr, cand);
        const unsigned long long qm = __ballot(queue);
        if (qm) {
            const uint32_t nq = (uint32_t)__builtin_popcountll(qm);
            if (qn + nq > 64u) verify_batch();
            if (queue) vq[wave][qn + (uint32_t)__builtin_popcountll(qm & ((1ull << lane) - 1ull))] = make_uint2(here, cand | (len << 13) | (multi << 20));
            qn += nq;
        }
        gpos = ngpos; here = nhere; g = gn;
    }
    if (qn) verify_batch();
#ifdef SFGPU_X_INS_NO_EPILOGUE        // experiment: ... and without writing anything back?
    if (a.cap != 0xFFFFFFFFu) return;
#endif
    __syncthreads();

    // ---- counts of the committed classes, and the classes this block created: ids, arena space, labels, hashes, table words
    const uint32_t n_new = s_nnew;
    uint32_t* new_slots = vq_words;                                               // (the queues are empty: every wavefront passed the barrier above)
    for (uint32_t s = threadIdx.x; s < kRegionSlots; s += kPartBlock) {
        const uint32_t e = slot32[s];
        if (e == kSlotEmpty) continue;
        const uint32_t idx = e & kSlotIdxMask;
        // (an atomic add: the route pass of the NEXT sub-batch may be running and adds what it counted for hot classes to the
        //  same words)
        if (idx < n_old) {
            const uint32_t c = ccnt[idx];
            if (c) { if (a.atomic_counts) atomicAdd(reinterpret_cast<unsigned long long*>(&a.table[2 * (rb + s) + 1]), (unsigned long long)c); else a.table[2 * (rb + s) + 1] += c; }
        }
        else { new_slots[atomicAdd(&s_newwords, 1u)] = s; }                      // s_newwords doubles as the list cursor here
    }
    __syncthreads();
    if (n_new) {
        // (n_new == s_newwords: every class that won its slot is in the list once)
        if (threadIdx.x == 0) s_newwords = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_new; i += kPartBlock) atomicAdd(&s_newwords, class_words((slot32[new_slots[i]] >> 13) & 0x7Fu));
        __syncthreads();
        if (threadIdx.x == 0) {
            s_cid0 = atomicAdd(a.gcls, (unsigned long long)n_new);
            atomicAdd(&a.ctr[CTR_NEW], (unsigned long long)n_new);
            s_arena0 = atomicAdd(a.arena_ctr, (unsigned long long)s_newwords);
            s_newwords = 0;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_new; i += kPartBlock) {
            const uint32_t s = new_slots[i];
            const uint32_t e = slot32[s];
            const uint32_t idx = e & kSlotIdxMask, len = (e >> 13) & 0x7Fu;
            const uint32_t rep = chead[idx].x;
            const uint64_t cid = s_cid0 + i;
            const uint64_t dst = s_arena0 + atomicAdd(&s_newwords, class_words(len)) + kProbeWords;      // the entry; its probe granule in front
            const uint32_t* p = reinterpret_cast<const uint32_t*>(a.bins + rep);
            const uint32_t w0 = p[0] & ~kHeadBit;
            const bool cpt = (w0 & kCompactBit) != 0u;
            const uint32_t clo = p[2], chi = p[3];
            auto word = [&](uint32_t k) { return cpt ? compact_id(w0, clo, chi, k) : (k ? p[k + 1] : w0); };
            probe_write(a.arena, dst - kProbeWords, word, len);
            reinterpret_cast<uint4*>(a.probe)[rb + s] = make_uint4(a.arena[dst - 4], a.arena[dst - 3], a.arena[dst - 2], a.arena[dst - 1]);
            entry_write(a.arena, dst, word, len);
            a.cls_hash[cid] = xxh64_words(word, len);
            a.cls_off[cid] = dst + 1; a.cls_len[cid] = len; a.cls_slot[cid] = (uint32_t)(rb + s);
            uint32_t tmp[kHead];
            const uint64_t h = label_mix64(word, len, tmp, a.mix_mode);                   // the table keeps the full 32-bit tag
            a.table[2 * (rb + s)] = (h & 0xFFFFFFFF00000000ull) | (unsigned long long)(kArenaBit | (uint32_t)(dst >> 2));
            a.table[2 * (rb + s) + 1] = ccnt[idx];                                         // (the slot was empty: count 0 before)
        }
    }
}

// deferred labels (region full): copy them out of the bins into a small CSR batch that the generic path can insert
// after the table has grown.  deferred[2 i] = granule index of the label in the bins, deferred[2 i + 1] = its length
// (| bit 31: it stands for a run of reads); the generic path takes the run lengths as weights.
__global__ void k_deferred_lens(uint64_t n, const uint32_t* __restrict__ deferred, uint32_t* lens) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    lens[i] = (i == n) ? 0u : (deferred[2 * i + 1] & 0x7Fu);
}
// (ids_out / off_out / weights_out point BEHIND what earlier calls saved; base_words = ids saved before: the offsets continue)
__global__ void k_deferred_copy(uint64_t n, const uint32_t* __restrict__ deferred, const uint4* __restrict__ bins,
                                const uint64_t* __restrict__ off64, uint32_t* ids_out, uint32_t* off_out, uint64_t* weights_out, uint64_t base_words) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    off_out[i] = (uint32_t)(base_words + off64[i]);
    if (i == n) return;
    uint32_t len = (uint32_t)(off64[i + 1] - off64[i]);
    const uint32_t* p = reinterpret_cast<const uint32_t*>(bins + deferred[2 * i]);
    const uint32_t w0 = p[0] & ~kHeadBit;
    const bool cpt = (w0 & kCompactBit) != 0u;                 // the label is one compact granule: first id + 8-bit steps
    // a run of identical reads carries its length in the granule behind the label (bit 31 of the deferred length word)
    weights_out[i] = (deferred[2 * i + 1] >> 31) ? (uint64_t)bins[deferred[2 * i] + (cpt ? 1u : label_granules(len))].x : 1ull;
    uint32_t* q = ids_out + off64[i];
    if (cpt) { for (uint32_t k = 0; k < len; ++k) q[k] = compact_id(w0, p[2], p[3], k); return; }
    q[0] = w0;
    for (uint32_t k = 1; k < len; ++k) q[k] = p[k + 1];
}

}  // namespace sfgpu
