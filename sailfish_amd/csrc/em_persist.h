// em_persist.h -- the whole EM / VBEM loop of optimize() as ONE launch (round 5).  Included by em.hip (inside namespace sfgpu,
// behind k_sweep_lds: it uses TileDesc, sweep_x, vb_x_head and the tile constants).
//
// Replaces the loop of src/CollapsedEMOptimizer.cpp:818-861 (while (itNum < minIter or (itNum < maxIter and !converged))
// { EMUpdate_ / VBEMUpdate_; convergence test; swap; ++itNum }) for plans that fit the chip in one round of blocks.
//
// Why: with one kernel per iteration (k_sweep_lds<.., .., FUSED>) every iteration still pays a kernel boundary (~1.4 us), a cold L2
// behind it (every dependent round trip at the head of a launch costs 1.2 - 2 us) and the wait for the launch's slowest tile.
// Here a block keeps its tile for ALL iterations; what changes hands between iterations are the window sums of the <= 6 tiles whose
// windows overlap a tile's own (TileDesc::e, k_nb_table) -- point to point, tile to tile, no grid barrier:
//   * a sum travels as a 16-byte GRANULE {lo, tag, hi, tag} written by ONE write-through store (buffer_store_dwordx4 sc1) and read
//     with sc1 loads (L1 bypassed; the per-XCD L2s are not coherent, the write-through store drops the line there): the data is its
//     own flag.  tag = EPOCH << 24 | the sweep's number + 1 (never 0: the arrays are zeroed before every launch; the epoch is a
//     process-wide launch counter, 1..255 in rotation -- round 6 -- so that a granule some EARLIER launch left behind with the
//     same sweep number can never validate, wherever a copy of it may have survived); each 8-byte half carries the tag, so a
//     torn granule cannot pass.  Two arrays by tag parity: tile T overwrites the granules of sweep s - 1 with those of
//     sweep s + 1 at the end of step s + 1, whose head needed its neighbours' sums of sweep s, which they publish after their
//     step s has read T's sums of sweep s - 1 (the overlap relation is symmetric);
//   * far members (escapes): tile V adds what it hands a far transcript t into a dense LDS accumulator (one FAR SLOT per distinct
//     far transcript of the tile, numbered by the plan) and publishes the slots as granules like the window; t's home thread (in
//     the lowest tile H whose window holds t) adds t's far slots in tile order in front of the window sums -- the sum k_update
//     forms as alphaOut[t] + partials, without the atomics -- and publishes x_t as a granule of its own, which V's threads read
//     in place of the x vector.  V -> H -> V: each hand-over names an EARLIER step or an earlier phase of the same step: no cycle;
//   * the stop test needs every tile's verdict on update u: thread 0 of every block posts "update u done" on one of 8 counters
//     (by block mod 8: the XCD) behind an atomicMax of u on "the last update some transcript of mine moved in"; wave 0 reads them
//     at the head of step u + 1, one sweep later -- by then they are complete unless a tile trails by a whole sweep.  converged_u
//     <=> all maxima < u (a tile only reaches update u + 1 if the loop did not stop at u).
// alpha', the gate and the relative change are formed as in the fused kernel (same additions, same order); x by vb_x_head / a
// reciprocal (em.hip): the same recurrence, series and exp as the other loops' vb_x_lean / vb_x_fast with the divisions as a reciprocal
// + two Newton steps -- agreement ~1e-13 relative, not bit identity: a run that falls back to one kernel per iteration (a give-up,
// several bootstrap lanes) differs from the persistent one at that level, both within 1e-9 of the CPU restatement the tests check against.  A transcript's alpha
// is read and written by its home thread only and goes to memory at every final update.
//
// Every spin is bounded: a tile that waits ~1 s (a block that never became resident: another process holds the CUs) raises the
// abort word, every other tile sees it in its own spins, the launch ends with status != 0 and em_run repeats the run with one
// kernel per iteration.  Two persistent launches of one process never overlap (em_run holds a per-device mutex).

typedef unsigned int gr4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kShards = 8;                      // arrival counters / convergence words, by block mod 8
constexpr uint32_t kCtlStride = 16;                  // 64-bit words between two control words (128 bytes: a line of their own)
constexpr uint32_t kCtlArrive = 0;                   // [4][kShards] update u, slot u % 4, monotonic over the run: low word arrivals, high word those that saw a change > tol
constexpr uint32_t kCtlAbort = 5 * kShards;          // != 0: some tile gave up waiting
constexpr uint32_t kCtlWords = (5 * kShards + 2) * kCtlStride;       // (+ a line for who gave up: tile + 1, thread, wait, step)
constexpr uint32_t kSpinLimit = 1u << 18;            // polls of one wait (0.2 - 1 us each: 50 - 250 ms) before a tile gives up
constexpr uint32_t kFanInMax = 16;                   // far slots one transcript may collect (its home thread reads them one by one)
// The block: 1024 threads, one window slot per thread, two blocks per CU.  The kernel is written for kPS window slots per thread
// (`-DSFGPU_PERSIST_BLOCK=512`: two slots, four wavefronts per SIMD, 128 VGPRs / 102 SGPRs instead of 64 / 80) because a step costs a
// wavefront ~570 instructions whatever its tile holds and half the wavefronts looked like half of that cost -- measured, it is the
// other way round: cfg3 15.9 -> 20.2 us per step, cfg2 8.7 -> 10.6, a 2 k-nonzero problem 7.4 -> 8.5.  A step is chains of dependent
// instructions (granule -> sum -> division -> LDS -> barrier ...), and what hides them is other wavefronts (profiles/r5_em_notes.md 4b).
#if !defined(SFGPU_PERSIST_BLOCK)
#define SFGPU_PERSIST_BLOCK 1024
#endif
constexpr int kPB = SFGPU_PERSIST_BLOCK;                // threads of a block of the persistent loop
constexpr int kPS = (kWin + kPB - 1) / kPB;             // window slots per thread: slot j of thread t is t + j * kPB
constexpr int kPWaves = kPB / kWave;
constexpr int kPAhead = kCntAhead * (kSweepBlock / kPB);      // class chunks (and counts) a thread requests ahead of phase A
#ifndef SFGPU_P_CAHEAD
#define SFGPU_P_CAHEAD 2
#endif
constexpr int kPCAhead = SFGPU_P_CAHEAD * (kSweepBlock / kPB);       // transcript-major chunks a thread requests ahead of phase C
static_assert(kPB % kWave == 0 && kPB <= kSweepBlock && kSweepBlock % kPB == 0, "block of the persistent loop");

// ---- the plan's far-slot tables (sfgpu_em_create -> em_persist_plan) ----------------------------------------------------------
// An escape is (tile, class, far transcript).  The distinct far transcripts of a tile are its FAR SLOTS, numbered in position
// order; global far slot g = TileDesc::f0 + f.  ft_list lists, per target position, the far slots that feed it in tile order
// (ftgt[pos] = its range).  All of it from two sorts of E keys.
__global__ void __launch_bounds__(kEmBlock)
k_far_keys(const TileDesc* __restrict__ td, const uint32_t* __restrict__ esc_where, uint64_t* keys, uint32_t* vals) {
    const uint32_t T = blockIdx.x, n = td[T].n_esc;
    const uint64_t e0 = td[T].e0;
    for (uint32_t i = threadIdx.x; i < n; i += kEmBlock) { keys[e0 + i] = ((uint64_t)T << 32) | esc_where[e0 + i]; vals[e0 + i] = (uint32_t)(e0 + i); }
}
__global__ void k_far_heads(uint64_t E, const uint64_t* __restrict__ ks, uint32_t* head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) head[i] = (i == 0 || ks[i] != ks[i - 1]) ? 1u : 0u; else if (i == E) head[i] = 0u;
}
// sorted entry i belongs to far slot g = (heads before i) + head[i] - 1
__global__ void k_far_assign(uint64_t E, const uint64_t* __restrict__ ks, const uint32_t* __restrict__ vs, const uint32_t* __restrict__ head,
                             const uint64_t* __restrict__ gsum, uint32_t* esc_g, uint32_t* far_pos, uint64_t* keys2) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const uint32_t g = (uint32_t)gsum[i] + head[i] - 1u;
    esc_g[vs[i]] = g;
    if (head[i]) { const uint32_t p = (uint32_t)ks[i]; far_pos[g] = p; keys2[g] = ((uint64_t)p << 32) | g; }
    const uint64_t F = gsum[E];
    if (i >= F) keys2[i] = ~0ull;                                        // (the second sort runs over E entries: the surplus goes last)
}
__global__ void k_far_tiles(uint32_t n_tiles, uint64_t E, const uint64_t* __restrict__ ks, const uint64_t* __restrict__ gsum, TileDesc* td, uint32_t* pflags) {
    const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
    if (T >= n_tiles) return;
    auto first_of = [&](uint64_t tile) -> uint32_t {                     // far slots in front of `tile`'s
        uint64_t lo = 0, hi = E;
        const uint64_t key = tile << 32;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (ks[mid] >= key) hi = mid; else lo = mid + 1; }
        return (uint32_t)gsum[lo];
    };
    const uint32_t a = first_of(T), b = first_of((uint64_t)T + 1u);
    td[T].f0 = a; td[T].nf = b - a;
    atomicMax(&pflags[1], b - a);
    atomicMax(&pflags[2], td[T].n_esc);                                  // (sizes the LDS copy of the tiles' far members)
}
__global__ void __launch_bounds__(kEmBlock)
k_far_local(const TileDesc* __restrict__ td, const uint32_t* __restrict__ esc_g, uint32_t* esc_far) {
    const uint32_t T = blockIdx.x, n = td[T].n_esc, f0 = td[T].f0;
    const uint64_t e0 = td[T].e0;
    for (uint32_t i = threadIdx.x; i < n; i += kEmBlock) esc_far[e0 + i] = esc_g[e0 + i] - f0;
}
// the far slots sorted by (target position, slot): ranges per target
__global__ void k_ft_ranges(uint64_t E, const uint64_t* __restrict__ gsum, const uint64_t* __restrict__ k2, uint2* ftgt, uint32_t* ft_list) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t F = gsum[E];
    if (k >= F) return;
    const uint32_t p = (uint32_t)(k2[k] >> 32);
    ft_list[k] = (uint32_t)k2[k];
    if (k == 0 || (uint32_t)(k2[k - 1] >> 32) != p) ftgt[p].x = (uint32_t)k;
    if (k + 1 == F || (uint32_t)(k2[k + 1] >> 32) != p) ftgt[p].y = (uint32_t)k + 1u;
}
// where a far slot's x is published (the first list entry of its target), and the checks: a target that no window holds has no
// home thread (flag 2), one that collects more than kFanInMax slots would stall its home thread (flag 4)
__global__ void k_far_xi(uint64_t E, const uint64_t* __restrict__ gsum, const uint32_t* __restrict__ far_pos, const uint2* __restrict__ ftgt,
                         const uint2* __restrict__ cov2, uint32_t* far_xi, uint32_t* pflags) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= gsum[E]) return;
    const uint32_t p = far_pos[g];
    const uint2 r = ftgt[p];
    far_xi[g] = r.x;
    if (cov2[p].x == cov2[p].y) atomicOr(&pflags[0], 2u);
    if (r.y - r.x > kFanInMax) atomicOr(&pflags[0], 4u);
}

// What the kernel reads in rare paths only (far members, the first sweep, the end of the loop) sits in DEVICE MEMORY and is fetched where
// it is used: as by-value arguments these words would be loop invariants held in SGPRs for the whole run -- 80 are to be had at 8 waves
// per SIMD, and what does not fit is spilled into VGPR lanes and fetched back with a v_readlane apiece all over the hot phases.
// (Measured the other way round too, profiles/r5_em_notes.md: EVERYTHING read through memory at each phase's start costs a dependent scalar
//  round trip per phase, 16 -> 21 us per step on cfg3.)
// ---- phase A's stream for the persistent loop (round 6): one RECORD per class, in four sizes ---------------------------------------
// Round 5 gave every class one 16-byte chunk (its first eight window slots, 16 bits each) and sent the members behind the eighth through
// an overflow list with atomics.  cfg3's class sizes are not the pool's geometric law (small labels collide in the pool and merge): of
// its 1.62 M classes 36 % hold <= 3 members, 29 % 4 - 6, 27 % 7 - 12 and 8 % more -- 22 % were "long" under the eight-slot chunk, 45 % of
// the nonzeros went through overflow chunks, and a tile streamed 51 KB of chunks + 13 KB of counts + ~10 KB of overflow per step.
// Now a window slot is a 10-bit field (kWin = 1023: slots 0 .. 1022, 1023 = the null slot whose x is 0) and a class is a record of
//   4 bytes (<= 3 in-window members), 8 bytes (<= 6), 16 bytes (<= 12), or 16 bytes + overflow chunks of 12 (more, or a far member),
// three slots to a dword.  The classes of a tile are PERMUTED so that the records of one size are contiguous (B1 | B2 | B3 | B4):
// lane per record, fixed-size loads, no run detection, and only B4 -- classes of more than 12 in-window members or with a far member,
// whose x arrives by atomic -- adds with atomics and needs phase B.  den[] is indexed by the permuted position; the transcript-major
// copy of phase C is remapped once at plan time (k_csc_remap), the count words are scattered into permuted order before every launch
// (k_persist_init) and live in the block's LDS for the whole run.  cfg3: 25 KB of records + 3 KB of overflow per tile and step.
struct TilePack {
    uint32_t a16, n1, n2, n3, n4, n_ov, ov0;              // a16: the tile's first 16-byte unit in `recs`: [B3 | B4 | overflow | B2 | B1]
    uint32_t nq;                                          // phase C: the tile's chunks in `cscp` (k_cscp_build), from unit q16
    uint32_t q16, pad[7];
};
static_assert(sizeof(TilePack) == 64, "one 64-byte line");
constexpr uint32_t kRecSlots3 = 12;                                   // slots of a 16-byte record / overflow chunk
__device__ __forceinline__ uint32_t pack3(uint32_t a, uint32_t b, uint32_t c) { return a | (b << 10) | (c << 20); }
// One block per tile.  recs: the tile's records start at unit a16 = c0 + s0 / 8 + 4 T (16 bytes per class bound the records, 2 bytes per
// nonzero the overflow chunks: no scan over the tiles); ovc: class (permuted, in the tile) of every overflow chunk, at s0 / 8 + T.
__global__ void __launch_bounds__(kSweepBlock)
k_pack_build(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ tile_c0, const uint64_t* __restrict__ tile_s0,
             const uint16_t* __restrict__ slot16, const TileDesc* __restrict__ td, const uint32_t* __restrict__ esc_cls,
             uint4* recs, uint16_t* ovc, TilePack* tp, uint32_t* cpos, uint32_t* esc_cls_p, uint32_t lds_slots) {
    __shared__ uint16_t perm_l[8192];                                 // class in the tile -> its permuted position (pass 1: its rank in the thread's run)
    __shared__ uint32_t ext_l[8192];                                  // bucket << 30 | overflow chunks in front of the class's
    extern __shared__ uint16_t slots_l[];                             // the tile's 16-bit slots, when they fit (lds_slots entries): the passes below read every
                                                                      // slot twice, a lane walking its class -- from memory that was 82 us of cfg3's plan
    __shared__ unsigned long long wsum[kSweepBlock / kWave];
    __shared__ uint32_t wsum2[kSweepBlock / kWave];
    __shared__ uint32_t tot_s[5];
    const uint32_t T = blockIdx.x, c0 = tile_c0[T], c1 = tile_c0[T + 1], nc = c1 - c0, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint64_t s0 = tile_s0[T];
    const uint32_t j0 = rowptr[c0], nnz = rowptr[c1] - j0;
    const bool in_lds = nnz <= lds_slots;
    if (in_lds) {                                                     // coalesced: 16 bytes per lane (s0 is a multiple of 8 slots)
        const uint4* src = reinterpret_cast<const uint4*>(slot16 + s0); uint4* dst = reinterpret_cast<uint4*>(slots_l);
        for (uint32_t i = tid; i < (nnz + 7u) / 8u; i += kSweepBlock) dst[i] = src[i];
        __syncthreads();
    }
    const uint16_t* const slots_base = in_lds ? slots_l : slot16 + s0;
    auto bucket_of = [&](uint32_t c, uint32_t& n_in) -> uint32_t {
        const uint32_t b = rowptr[c0 + c], k = rowptr[c0 + c + 1] - b;
        const uint16_t* sl = slots_base + (b - j0);
        n_in = 0; bool far = false;
        for (uint32_t m = 0; m < k; ++m) { if (sl[m] == (uint16_t)kWin) far = true; else ++n_in; }
        return (far || n_in > kRecSlots3) ? 3u : (n_in <= 3u ? 0u : (n_in <= 6u ? 1u : 2u));
    };
    // pass 1: a thread's run of classes [q0, q1): bucket, rank inside the run, overflow chunks in front
    const uint32_t per = (nc + kSweepBlock - 1u) / kSweepBlock, q0 = tid * per < nc ? tid * per : nc, q1 = q0 + per < nc ? q0 + per : nc;
    unsigned long long mine = 0ull; uint32_t mine_ov = 0u;             // four 16-bit counters (a tile holds < 8192 classes)
    for (uint32_t c = q0; c < q1; ++c) {
        uint32_t n_in; const uint32_t bk = bucket_of(c, n_in);
        perm_l[c] = (uint16_t)((mine >> (16u * bk)) & 0xFFFFull);
        ext_l[c] = (bk << 30) | mine_ov;
        mine += 1ull << (16u * bk);
        if (bk == 3u && n_in > kRecSlots3) mine_ov += (n_in - 1u) / kRecSlots3;
    }
    unsigned long long incl = mine; uint32_t incl_ov = mine_ov;
    for (int o = 1; o < kWave; o <<= 1) {
        const unsigned long long v = __shfl_up(incl, o, kWave); const uint32_t v2 = __shfl_up(incl_ov, o, kWave);
        if ((int)lane >= o) { incl += v; incl_ov += v2; }
    }
    if (lane == kWave - 1) { wsum[wave] = incl; wsum2[wave] = incl_ov; }
    __syncthreads();
    unsigned long long base = incl - mine; uint32_t base_ov = incl_ov - mine_ov;
    for (uint32_t w = 0; w < wave; ++w) { base += wsum[w]; base_ov += wsum2[w]; }
    if (tid == kSweepBlock - 1u) {
        const unsigned long long t = base + mine;
        tot_s[0] = (uint32_t)(t & 0xFFFFull); tot_s[1] = (uint32_t)((t >> 16) & 0xFFFFull); tot_s[2] = (uint32_t)((t >> 32) & 0xFFFFull); tot_s[3] = (uint32_t)((t >> 48) & 0xFFFFull);
        tot_s[4] = base_ov + mine_ov;
    }
    __syncthreads();
    const uint32_t n1 = tot_s[0], n2 = tot_s[1], n3 = tot_s[2], n4 = tot_s[3], n_ov = tot_s[4];
    const uint32_t first[4] = {0u, n1, n1 + n2, n1 + n2 + n3};
    const uint32_t a16 = c0 + (uint32_t)(s0 / 8u) + 4u * T, ov0 = (uint32_t)(s0 / 8u) + T;
    if (tid == 0u) { tp[T].a16 = a16; tp[T].n1 = n1; tp[T].n2 = n2; tp[T].n3 = n3; tp[T].n4 = n4; tp[T].n_ov = n_ov; tp[T].ov0 = ov0; }      // (nq / q16: k_cscp_build)
    // pass 2: the permuted position
    for (uint32_t c = q0; c < q1; ++c) {
        const uint32_t bk = ext_l[c] >> 30;
        const uint32_t pos = first[bk] + (uint32_t)((base >> (16u * bk)) & 0xFFFFull) + perm_l[c];
        perm_l[c] = (uint16_t)pos;
        ext_l[c] = (bk << 30) | (base_ov + (ext_l[c] & 0x3FFFFFFFu));
        cpos[c0 + c] = c0 + pos;
    }
    __syncthreads();
    // pass 3: the records
    uint4* const r3 = recs + a16; uint4* const r4 = r3 + n3; uint4* const rov = r4 + n4;
    uint2* const r2 = reinterpret_cast<uint2*>(rov + n_ov); uint32_t* const r1 = reinterpret_cast<uint32_t*>(r2 + n2);
    for (uint32_t c = tid; c < nc; c += kSweepBlock) {
        const uint32_t b = rowptr[c0 + c], k = rowptr[c0 + c + 1] - b;
        const uint16_t* sl = slots_base + (b - j0);
        const uint32_t bk = ext_l[c] >> 30, pos = perm_l[c];
        uint32_t w[kRecSlots3], n = 0;
        uint32_t at = ov0 + (ext_l[c] & 0x3FFFFFFFu);
        bool head = true;
        auto flush = [&]() {
            for (uint32_t m = n; m < kRecSlots3; ++m) w[m] = (uint32_t)kWin;
            const uint4 v = make_uint4(pack3(w[0], w[1], w[2]), pack3(w[3], w[4], w[5]), pack3(w[6], w[7], w[8]), pack3(w[9], w[10], w[11]));
            if (head) {
                if (bk == 0u) r1[pos] = v.x;
                else if (bk == 1u) r2[pos - first[1]] = make_uint2(v.x, v.y);
                else if (bk == 2u) r3[pos - first[2]] = v;
                else r4[pos - first[3]] = v;
                head = false;
            } else { ovc[at] = (uint16_t)pos; rov[at - ov0] = v; ++at; }
            n = 0;
        };
        const bool single = k == 1u;                                  // (a singleton's denominator is never used: nothing to read)
        for (uint32_t m = 0; m < k; ++m) {
            const uint32_t s = sl[m];
            if (s == (uint32_t)kWin || single) continue;              // a far member: its x arrives through the tile's far slots
            if (n == kRecSlots3) flush();
            w[n++] = s;
        }
        flush();                                                      // (the head record always exists; a full last chunk is written here)
    }
    // pass 4: the far members' classes as permuted positions
    const uint32_t n_esc = td[T].n_esc; const uint64_t e0 = td[T].e0;
    for (uint32_t i = tid; i < n_esc; i += kSweepBlock) { const uint32_t t = esc_cls[e0 + i]; esc_cls_p[e0 + i] = ((uint32_t)perm_l[(t >> 16) & 0x1FFFu] << 16) | (t & kSingle); }
}
// ---- phase C's stream for the persistent loop (round 6): every chunk PURE -------------------------------------------------------------
// The transcript-major copy of k_sweep_lds keeps a tile's nonzeros sorted by window slot in chunks of 8 whatever the slots' runs look
// like: a chunk that straddles two slots is MIXED (8 classes + 8 slots, 32 bytes, walked run by run: ~100 instructions, two loads, up
// to 8 atomics) -- a fifth of cfg3's chunks, and they cost the persistent loop 1.0 - 2.2 us of a step (`-DSFGPU_P_NOMIXED`: cfg3 VBEM
// 12.9 -> 10.6 us without them, which also drops their 20 % of the work).  Here a slot's run is padded to whole chunks with the null class
// (count / denominator 0), so every chunk is one slot: eight 13-bit class positions (permuted: k_pack_build) and the slot + singleton
// bit in the dwords' spare bits -- 16 bytes, one load, eight LDS reads, one atomic; ~10 % more chunks than pure + mixed before, no second
// loop, no slot array.  Built from the tile's sorted nonzeros (kv: key << 16 | class in the tile, key = singleton << 11 | slot, stable: the
// order of k_tile_build), so the sums are formed in the same order in every plan.
//   chunk = { c0 | c1 << 13 | slot[5:0] << 26,  c2 | c3 << 13 | slot[9:6] << 26 | single << 31,  c4 | c5 << 13,  c6 | c7 << 13 }
__global__ void __launch_bounds__(kSweepBlock)
k_cscp_build(const TileDesc* __restrict__ td, TilePack* tp, const uint32_t* __restrict__ cpos, const uint32_t* __restrict__ kv,
             const uint32_t* __restrict__ idx, const uint32_t* __restrict__ tile_in, const uint64_t* __restrict__ tile_s0, uint4* cscp, uint32_t null_cls) {
    __shared__ uint32_t start[kEscBin + 1];                              // first entry of key b among the tile's sorted nonzeros
    __shared__ uint32_t coff[kEscBin + 1];                               // first chunk of key b
    __shared__ uint32_t wsum[kSweepBlock / kWave];
    const uint32_t T = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t c0 = td[T].c0, j0 = idx[T], n = tile_in[T];
    const uint32_t* __restrict__ e = kv + j0;
    for (uint32_t b = tid; b <= kEscBin; b += kSweepBlock) {              // lower bound of key b (the escape bin's start = n)
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((e[mid] >> 16) >= b) hi = mid; else lo = mid + 1u; }
        start[b] = lo;
    }
    __syncthreads();
    constexpr uint32_t kPer = (kEscBin + kSweepBlock - 1) / kSweepBlock;  // bins per thread in the scan (3)
    uint32_t mine[kPer], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) { const uint32_t b = tid * kPer + k; mine[k] = b < kEscBin ? (start[b + 1] - start[b] + 7u) >> 3 : 0u; sum += mine[k]; }
    uint32_t incl = sum;
    for (int o = 1; o < kWave; o <<= 1) { const uint32_t v = __shfl_up(incl, o, kWave); if ((int)lane >= o) incl += v; }
    if (lane == kWave - 1) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = incl - sum;
    for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) { const uint32_t b = tid * kPer + k; if (b < kEscBin) coff[b] = base; base += mine[k]; }
    if (tid == kSweepBlock - 1u) coff[kEscBin] = base;
    __syncthreads();
    const uint32_t nq = coff[kEscBin];
    const uint32_t q16 = (uint32_t)(tile_s0[T] / 8u) + 2112u * T;          // (a tile of n8 padded nonzeros has at most n8 / 8 + 2046 chunks: one partial chunk per key; + 63 of the transposition below)
    // The chunks of a slot follow each other, and thread t of the loop takes chunk t: the lanes of a wavefront would add to ONE accumulator,
    // and same-address LDS atomics of one instruction serialise (an ablation with plain stores in their place: cfg3 - 1.9 us of a step).  So
    // the chunks are stored TRANSPOSED: position row * 64 + col holds chunk col * rows + row (rows = ceil(nq / 64)) -- the 64 lanes of a load
    // get chunks `rows` apart, i.e. of different slots, and the loads stay contiguous.  Positions without a chunk hold null chunks.
    const uint32_t rows = (nq + 63u) / 64u, nq_pad = rows * 64u;
    if (tid == 0u) { tp[T].nq = nq_pad; tp[T].q16 = q16; }
    for (uint32_t j = nq + tid; j < nq_pad; j += kSweepBlock) {
        const uint32_t n2 = null_cls | (null_cls << 13);
        cscp[q16 + (j % rows) * 64u + j / rows] = make_uint4(n2, n2 | (1u << 31), n2, n2);      // (singleton bit: adds the sum of its zeros, whatever x is)
    }
    for (uint32_t j = tid; j < nq; j += kSweepBlock) {
        uint32_t lo = 0, hi = kEscBin;                                   // the last key whose first chunk is <= j (keys without entries share their successor's offset)
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (coff[mid] <= j) lo = mid; else hi = mid; }
        while (coff[lo + 1u] <= j) ++lo;                                  // (cannot run past kEscBin - 1: j < nq = coff[kEscBin])
        const uint32_t b = lo, first = start[b] + 8u * (j - coff[b]), end = start[b + 1];
        uint32_t c[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) c[k] = first + k < end ? cpos[c0 + (e[first + k] & 0xFFFFu)] - c0 : null_cls;
        const uint32_t slot = b & 0x3FFu, single = (b & 0x800u) ? 1u : 0u;
        cscp[q16 + (j % rows) * 64u + j / rows] = make_uint4(c[0] | (c[1] << 13) | ((slot & 63u) << 26), c[2] | (c[3] << 13) | ((slot >> 6) << 26) | (single << 31), c[4] | (c[5] << 13), c[6] | (c[7] << 13));
    }
}

struct PersistCold {
    EmState* st; const double* x; const uint32_t* inv;    // x of sweep 0 (init made it, in the caller's order); position -> transcript
    const uint32_t* esc_cls; const uint32_t* esc_far;     // per escape: class << 16 | single ; far slot in the tile
    const uint32_t* far_pos; const uint32_t* far_xi;      // per far slot (TileDesc::f0 + f): target position ; index of its x granule
    const uint32_t* ft_list;                              // the far slots that feed a position, in tile order (ranges: PersistArgs::ftgt)
    const uint32_t* unc; const uint32_t* unc_n;           // positions no window holds (inactive transcripts)
    double* tmax; uint32_t* status;                       // status: 0 = ran to the stop, 1 = gave up (see above)
    unsigned long long* dbg;                              // SFGPU_P_STAMP builds: [tile][8] time spent per phase, summed over the steps (10 ns units)
};
struct PersistArgs {
    const TileDesc* tiles; const PersistCold* cold;
    uint32_t min_iter, max_iter, n_tiles; int check_mode;
    const TilePack* tp; const uint4* recs; const uint16_t* ovc;    // phase A: a record per class in four sizes, the overflow chunks' classes (k_pack_build)
    const uint32_t* counts;                               // cnt8: count | singleton << 31, in the tiles' permuted class order
    const uint4* cscp;                                    // phase C: the transcript-major copy, every chunk one slot's (k_cscp_build)
    const double* lenc; double* alpha;                    // by position of the plan's order
    const uint2* ftgt;                                    // per position: [k0, k1) of ft_list (null: the plan has no far members)
    // the exchange buffer: ONE buffer descriptor, the pieces by byte offset; part_off: granules of the window sums, slot-major
    // (TileDesc::off + slot), by tag parity
    void* xbuf; uint32_t xbuf_bytes, part_off[2];
    unsigned long long* ctl;                              // kCtlWords control words (zeroed before the launch)
    double tol, log_norm;
    uint32_t den_cap, far_cap;                            // LDS: den[den_cap + 1] (den_cap = the plan's null class), facc[far_cap]
    uint32_t esc_ln;                                      // LDS: far members of the tile kept on chip (the plan's most, or what fits)
    uint32_t far_off0, far_stride;                        // exchange buffer: far slots' granules at far_off0 + parity * far_stride, the far targets' x at + 2 * far_stride
    uint32_t tag0;                                        // the launch's EPOCH in the top byte of every tag (see the tags below)
    int ablate;                                           // dev: 1 = no tag checks (timing only: wrong results); 3 = tile 0 gives up in step 2, 4 = tile 0 starts 100 us late (tests)
};

// before every launch: tags, counters, abort word and status zeroed, the cold block written -- ONE kernel, and its stores are
// WRITE-THROUGH (sc1) like every later store to the exchange buffer: zeros written with plain stores stay behind as clean lines in the
// L2 of the XCD that wrote them, a write-through store from another XCD does not reach those copies, and a tile that runs on that XCD
// then polls (sc1 loads are L2-served) a line of zeros for ever.  Seen with three bootstrap lanes (other streams' kernels between the
// launches: blocks land on other XCDs than b mod 8): tiles of one or two XCDs waited in step 1 for sums their neighbours had published.
// ... and cnt8, the count words the loop reads: the handle's CURRENT counts (the bootstrap resamples them before every run) in the
// tiles' permuted class order (cpos: class of the plan -> its position).
__global__ void __launch_bounds__(256)
k_persist_init(void* xbuf, uint32_t bytes, uint32_t cold_first16, uint32_t cold_n16, PersistCold* d_cold, PersistCold cold,
               uint64_t C, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ cpos, uint32_t* __restrict__ cnt8) {
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < C; c += (uint64_t)gridDim.x * blockDim.x)
        cnt8[cpos[c]] = counts[c];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, bytes, 0x00020000);
    const uint32_t n16 = bytes / 16u;
    const gr4 z = {0u, 0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x)
        if (i - cold_first16 >= cold_n16) __builtin_amdgcn_raw_buffer_store_b128(z, rx, i * 16u, 0, 16);
    if (blockIdx.x == 0 && threadIdx.x == 0) *d_cold = cold;
}

// An LDS read by BYTE OFFSET.  The kernel's only LDS is its dynamic block, which starts at offset 0 (checked when the kernel starts), and
// xs / den sit at compile-time offsets in it: addressed through the `plds` symbol every gather of phases A and C carried a `v_add_u32 v, 0, v`
// (the symbol's address is a link-time constant the compiler cannot fold) -- one of the ~5 instructions a member costs.  From an integer the
// offset goes into the instruction's immediate field.
typedef const double __attribute__((address_space(3))) lds_cdouble;
__device__ __forceinline__ double lds_f64(uint32_t byte_off) { return *reinterpret_cast<lds_cdouble*>(static_cast<uintptr_t>(byte_off)); }
constexpr uint32_t kLdsXs = 0u, kLdsAcc = (uint32_t)(kWin + 2) * 8u, kLdsDen = 2u * (uint32_t)(kWin + 2) * 8u;      // byte offsets of xs / acc / den (see the carve below)

__device__ __forceinline__ double gr_value(const gr4& g) { return __hiloint2double((int)g.z, (int)g.x); }
__device__ __forceinline__ bool gr_ok(const gr4& g, uint32_t tag) { return g.y == tag && g.w == tag; }

#ifdef SFGPU_P_STAMP
#if SFGPU_P_STAMP == 2                                                     // (2: no drain of the memory queue in front of a stamp -- issue times; the stamps behind barriers are exact)
#define SFP_DRAIN() do { } while (0)
#else
#define SFP_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
#define SFP_STAMP(k) do { if (tid == 0u) { SFP_DRAIN(); const unsigned long long t_ = wall_clock64(); pst[k] += t_ - pst[7]; pst[7] = t_; } } while (0)
#else
#define SFP_STAMP(k) do { } while (0)
#endif
// the cold block / the tile's record, looked at from a rare path: the pointer is made opaque THERE, so no load of it is hoisted out
#ifdef SFGPU_P_NOCONF                                                      // dev, timing only: every LDS gather of phases A and C without bank conflicts (wrong addresses)
#define SFP_BANK(v, m) (((v) & ~31u) | ((lane + (m)) & 31u))
#else
#define SFP_BANK(v, m) (v)
#endif
#ifdef SFGPU_P_L2HIT                                                       // dev, timing only: every stream load of a tile falls into its first 1024 entries (all L2 hits)
#define SFP_IX(i) ((i) & 1023u)
#else
#define SFP_IX(i) (i)
#endif
#if defined(SFGPU_P_NT_C)                                                  // dev: phase C's chunks with non-temporal loads (do the records then stay in the L2?)
#define SFP_LDC(p) ([&]() { typedef uint32_t u32x4_ __attribute__((ext_vector_type(4))); const u32x4_ v_ = __builtin_nontemporal_load(reinterpret_cast<const u32x4_*>(p)); return make_uint4(v_.x, v_.y, v_.z, v_.w); }())
#else
#define SFP_LDC(p) (*(p))
#endif
#define SFP_COLD(name) const PersistCold* name = a.cold; asm volatile("" : "+s"(name))
#define SFP_TILE(name) const TileDesc* name = a.tiles + blockIdx.x; asm volatile("" : "+s"(name))
template <bool VB>
__global__ void __launch_bounds__(kPB) __attribute__((amdgpu_waves_per_eu(kPB / 128, kPB / 128)))
k_em_persist(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) double plds[];
    double* const xs = plds;                               // [kWin + 1]  (+ the slot of the null words: x = 0)
    double* const acc = xs + (kWin + 2);                   // [kWin + 1]
    double* const den = acc + (kWin + 2);                  // [den_cap + 1]  denominators, then count / denom (+ the null class at den_cap)
    double* const facc = den + (a.den_cap + 2);            // [far_cap]
    double* const fxs = facc + a.far_cap;                  // [far_cap]  x of the far slots' transcripts, fetched once per step (head)
    double* const wmax = fxs + a.far_cap;                  // [2][waves]: the wavefronts' largest relative change, by update parity
    uint32_t* const sctl = reinterpret_cast<uint32_t*>(wmax + 2 * kPWaves);     // [0] stop, [1] abort (heads), [2..3] block saw a change > tol (by step parity), [4..5] abort (phases, by step parity)
    uint32_t* const hprev = sctl + 8;                      // [4][kShards] the counters' high words at wave 0's last visit
    // What the far paths need of the plan, on chip (round 5, second pass): read through the cold block these were two or three DEPENDENT
    // round trips to memory in the head, in phase A and again in phase C of every step -- in the tiles that have far members, and those
    // set the pace of all (cfg2's tile 0: x + update 2.8 us against 1.45, C 3.1 against 2.3, tools/r5_pstamp.sh).
    uint32_t* const cntl = hprev + 4 * kShards;            // [den_cap + 2] the tile's count words (bit 31: singleton), permuted order: read from memory ONCE per launch (round 6)
    uint32_t* const far_xi_l = cntl + (a.den_cap + 2);     // [far_cap] per far slot of the tile: index of its transcript's x granule
    uint32_t* const ftg_l = far_xi_l + a.far_cap;          // [kWin] per window slot: the first far slot that feeds it (plans with far members)
    uint2* const esc_l = reinterpret_cast<uint2*>(ftg_l + (a.ftgt ? kWin : 0) + ((a.far_cap + (a.ftgt ? kWin : 0)) & 1u));      // [esc_ln] far members: {class << 16 | single, far slot}
#ifdef SFGPU_P_STAMP
    unsigned long long* const pst = reinterpret_cast<unsigned long long*>(esc_l + a.esc_ln);       // [0..6] phase sums, [7] the last stamp
    if (threadIdx.x == 0) { for (int k = 0; k < 7; ++k) pst[k] = 0ull; pst[7] = wall_clock64(); }
#endif
    const uint32_t tid0 = threadIdx.x;
    // ---- the tile: what the hot phases need, as scalars; the rest of the record (e0, f0) is read where a far member needs it
    uint32_t lo, nc, np, n_esc, nf, off, nb_n, nb_before, delta[kNbMax];
    uint32_t n1, n2, n3, n4, n_ov;                                       // the tile's records by size (k_pack_build): B1 | B2 | B3 | B4 in den[]'s order
    const uint4* __restrict__ recs; const uint16_t* __restrict__ ovcp;   // recs: [B3 | B4 | overflow | B2 | B1]
    const uint4* __restrict__ pure;                                      // the tile's chunks of phase C (all of one slot each: k_cscp_build)
    uint32_t flags[kPS];
    // ---- what a thread keeps for the whole run: which of the overlapping tiles hold the position of its window slot (bits 0..5), whether
    //      it has a slot (bit 30) and whether this tile is the position's home (bit 31).  Everything else it needs per step (effLen,
    //      alpha, the far slots that feed it) is read again every step: registers are what this kernel is short of (64 per thread at
    //      two blocks per CU), and those words sit in the L2.
    {
        const TileDesc t = a.tiles[blockIdx.x];
        lo = t.lo; nc = t.nc; n_esc = t.n_esc; nf = t.nf; off = (uint32_t)t.off; nb_n = t.nb_n; nb_before = t.nb_before;
#ifdef SFGPU_P_NOFAR                                                       // dev, timing only: no far members (is the tile that has them the one everybody waits for?)
        n_esc = 0u; nf = 0u;
#endif
        {
            const TilePack pk = a.tp[blockIdx.x];
            n1 = pk.n1; n2 = pk.n2; n3 = pk.n3; n4 = pk.n4; n_ov = pk.n_ov;
#ifdef SFGPU_P_NOB4                                                        // dev, timing only: no long classes
            n4 = 0u; n_ov = 0u;
#endif
#ifdef SFGPU_P_NOA                                                         // dev, timing only: phase A without any record
            n1 = n2 = n3 = n4 = n_ov = 0u;
#endif
            recs = a.recs + pk.a16; ovcp = a.ovc + pk.ov0;
            np = pk.nq; pure = a.cscp + pk.q16;
#ifdef SFGPU_P_NOC                                                         // dev, timing only: phase C without any chunk
            np = 0u;
#endif
        }
        {   // the count words: on chip for the whole run
            const uint32_t* __restrict__ cnt = a.counts + t.c0;
            for (uint32_t c = threadIdx.x; c < nc; c += kPB) cntl[c] = cnt[c];
        }
        const uint32_t span = t.span;
        bool home[kPS];
#pragma unroll
        for (int q = 0; q < kPS; ++q) { home[q] = tid0 + q * kPB < span; flags[q] = home[q] ? 0x40000000u : 0u; }
#pragma unroll
        for (int j = 0; j < kNbMax; ++j) {
            const uint4 e = t.e[j];                                      // {lo', span', off', tile'}
            delta[j] = e.z - e.x;                                        // the slot of position p in that tile's piece: p + delta
#pragma unroll
            for (int q = 0; q < kPS; ++q) {
                const uint32_t sidx = tid0 + q * kPB;
                const bool in = sidx < span && (uint32_t)j < nb_n && (lo + sidx - e.x) < e.y;
                if (in) { flags[q] |= 1u << j; if ((uint32_t)j < nb_before) home[q] = false; }
            }
        }
#pragma unroll
        for (int q = 0; q < kPS; ++q) if (home[q]) flags[q] |= 0x80000000u;
    }
    const double* __restrict__ lenc = a.lenc + lo; double* __restrict__ alpha = a.alpha + lo;
    const uint2* __restrict__ ftgt = a.ftgt ? a.ftgt + lo : nullptr;
    unsigned long long* const ctl = a.ctl;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, a.xbuf_bytes, 0x00020000);
    auto gr_load = [&](uint32_t piece_off, uint32_t idx) -> gr4 { return __builtin_amdgcn_raw_buffer_load_b128(rx, idx * 16u, piece_off, 16); };      // sc1: past the L1
    auto gr_store = [&](uint32_t piece_off, uint32_t idx, double v, uint32_t tag) {                                                                    // sc1: write-through
        gr4 w; w.x = (uint32_t)__double2loint(v); w.y = tag; w.z = (uint32_t)__double2hiint(v); w.w = tag;
        __builtin_amdgcn_raw_buffer_store_b128(w, rx, idx * 16u, piece_off, 16);
    };
    // a wait that does not end: look at the abort word now and then, give up after ~1 s (returns true when the wait must be left)
    // (`word`: the block's abort word this wait reports to -- sctl[1] for the waits of a head, sctl[4 + step parity] for those inside
    //  the phases: every word is read by ALL threads behind ONE barrier that no writer of it can have passed, so the block leaves as one)
#ifdef SFGPU_P_PROGRESS
    uint32_t why = 0u;                                                   // (dev builds: what the current wait is for -- 1 arrivals, 2 a window sum, 3 a far slot, 4 a far member's x)
#define SFP_WHY(k) why = (k)
#else
    constexpr uint32_t why = 0u;
#define SFP_WHY(k) do { } while (0)
#endif
    auto spin_check = [&](uint32_t& spins, uint32_t word) -> bool {
#ifndef SFGPU_P_SLEEP
#define SFGPU_P_SLEEP 2
#endif
        if (SFGPU_P_SLEEP) __builtin_amdgcn_s_sleep(SFGPU_P_SLEEP);
        if ((++spins & 63u) != 0u) return false;
        asm volatile("buffer_inv sc1" ::: "memory");                         // (agent-scope invalidate now and then: nothing this CU caches may stand between a poll and memory)
        if (__hip_atomic_load(&ctl[kCtlAbort * kCtlStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
#ifdef SFGPU_P_PROGRESS
            { SFP_COLD(cq); cq->dbg[gridDim.x + blockIdx.x] = (unsigned long long)why * 1000ull + sctl[6]; }
#endif
            sctl[word] = 1u; return true;
        }
        if (spins >= kSpinLimit) {
#ifdef SFGPU_P_PROGRESS
            { SFP_COLD(cq); cq->dbg[gridDim.x + blockIdx.x] = 900000ull + (unsigned long long)why * 1000ull + sctl[6]; }      // (9xxxxx: the tile that gave up)
#endif
            // who gave up, for the log: [1] tile + 1, [2] thread, [3] what it waited for (the abort word of the wait: 1 = a head, 4 / 5 = a phase)
            const unsigned long long first = atomicCAS(&ctl[(kCtlAbort + 1) * kCtlStride], 0ull, (unsigned long long)blockIdx.x + 1ull);
            if (first == 0ull) { ctl[(kCtlAbort + 1) * kCtlStride + 1] = threadIdx.x; ctl[(kCtlAbort + 1) * kCtlStride + 2] = why ? why : word; ctl[(kCtlAbort + 1) * kCtlStride + 3] = sctl[6]; }
            __hip_atomic_store(&ctl[kCtlAbort * kCtlStride], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sctl[word] = 1u;
            return true;
        }
        return false;
    };
    if (tid0 < 8u + 4u * kShards) sctl[tid0] = 0u;                       // (and hprev)
    // (lds_f64 addresses xs / den by byte offset from 0: this kernel has no static LDS in front of its dynamic block.  Should a toolchain
    //  ever place it elsewhere the launch reports "gave up" at once and the run is repeated with one kernel per iteration.)
    if (__builtin_amdgcn_groupstaticsize() != 0u) { if (tid0 == 0u) { SFP_COLD(cq); *cq->status = 1u; } return; }
#pragma unroll
    for (int q = 0; q < kPS; ++q) acc[tid0 + q * kPB] = 0.0;

    auto x_of = [&](double ap_, double l) -> double {
#ifdef SFGPU_P_VBCHEAP                                                     // dev, timing only: VBEM with EM's x (what the digamma / exp chain of the head costs)
        if (VB) return (ap_ > kTiny) ? sweep_x<true>(ap_ / (l * a.log_norm)) : 0.0;
#endif
#ifdef SFGPU_P_OLDHEAD                                                     // dev: the head's arithmetic of round 5 (vb_x_fast from constant memory, IEEE divisions)
        if (VB) return (ap_ > kTiny) ? sweep_x<true>(vb_x_fast(ap_, a.log_norm, l)) : 0.0;
        return sweep_x<false>(ap_ / l);
#endif
        if (VB) return (ap_ > kTiny) ? sweep_x<true>(vb_x_head(ap_, a.log_norm, l)) : 0.0;       // :300-320
        return sweep_x<false>(ap_ * fast_rcp(l));
    };
    // x of the far member behind far slot f of this tile (f0: the tile's first), for sweep s: sweep 0 reads the x vector, later sweeps
    // the granule the transcript's home thread published at the head of its step s
    auto far_x = [&](const PersistCold* cp, uint32_t f0, uint32_t f, uint32_t s, uint32_t word) -> double {
        if (s == 0u) { const uint32_t p = cp->far_pos[f0 + f]; const uint32_t* inv = cp->inv; return cp->x[inv ? inv[p] : p]; }
        const uint32_t xi = far_xi_l[f], xo = a.far_off0 + 2u * a.far_stride;
        gr4 g = gr_load(xo, xi);
        SFP_WHY(4u);
        if (a.ablate != 1) for (uint32_t spins = 0; !gr_ok(g, a.tag0 + s);) { if (spin_check(spins, word)) break; g = gr_load(xo, xi); }
        return gr_value(g);
    };
    {   // the far paths' tables into the LDS, once
        SFP_COLD(cp); SFP_TILE(tp);
        if (nf) { const uint32_t f0 = tp->f0; for (uint32_t f = tid0; f < nf; f += kPB) far_xi_l[f] = cp->far_xi[f0 + f]; }
        if (ftgt) {
#pragma unroll
            for (int q = 0; q < kPS; ++q) {
                const uint32_t sidx = tid0 + q * kPB;
                if (flags[q] & 0x40000000u) { const uint2 r = ftgt[sidx]; ftg_l[sidx] = r.y > r.x ? cp->ft_list[r.x] : 0u; }
            }
        }
        if (n_esc) {
            const uint64_t e0 = tp->e0;
            const uint32_t n = n_esc < a.esc_ln ? n_esc : a.esc_ln;
            for (uint32_t i = tid0; i < n; i += kPB) esc_l[i] = make_uint2(cp->esc_cls[e0 + i], cp->esc_far[e0 + i]);
        }
    }
    if (a.ablate == 4 && blockIdx.x == 0u) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < 10000ull) __builtin_amdgcn_s_sleep(8); }      // tests: tile 0 starts 100 us late
    __syncthreads();

    uint32_t k_done = 0;                                                 // updates done when the loop ends
    bool conv_last = false;
    double lm_keep = -1.0;                                               // the thread's largest relative change in the last FINAL update
    for (uint32_t s = 0;; ++s) {
        // (the thread's index, opaque to the compiler once per step: left alone it hoists every per-thread address of the loop body --
        //  a dozen 64-bit pointers -- out of the loop and spills them)
        uint32_t tid = tid0;
        asm volatile("" : "+v"(tid));
        const uint32_t lane = tid & (kWave - 1), wave = tid / kWave;
        if (tid == 0u) sctl[6] = s;                                          // (for the log of a give-up)
#ifdef SFGPU_P_PROGRESS
        if (tid == 0u) {
            SFP_COLD(cq); __hip_atomic_store(&cq->dbg[blockIdx.x], (unsigned long long)s + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s == 0u) {                                                   // when and where the block started (its first step)
                __hip_atomic_store(&cq->dbg[2 * gridDim.x + blockIdx.x], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20), hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_XCC_ID, HW_REG_HW_ID
                __hip_atomic_store(&cq->dbg[3 * gridDim.x + blockIdx.x], ((unsigned long long)xcc << 32) | hwid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#endif
        // ================= head of step s: [the stop test of update s - 1] update s, x of sweep s =================
        bool has[kPS], home[kPS];
#pragma unroll
        for (int q = 0; q < kPS; ++q) { has[q] = (flags[q] & 0x40000000u) != 0u; home[q] = (flags[q] & 0x80000000u) != 0u; }
        double ap_v[kPS], xv[kPS];
#pragma unroll
        for (int q = 0; q < kPS; ++q) { ap_v[q] = 0.0; xv[q] = 0.0; }
        double lm = -1.0; unsigned ncv = 0u;
        // phase A's stream chunks and phase B's class counts: requested in the head once the operands are in (they miss the L2 -- a tile's
        // stream is read once per step and 64 tiles share 4 MB --, and the x arithmetic and the head's barrier hide the round trip);
        // phase C's first chunks are requested at the start of phase A
        // the thread's records (see k_pack_build): two of B1, one each of B2, B3, B4 -- the lanes are dealt differently per size so that
        // the wavefronts' work evens out (B3 starts at the block's middle, B4 at its end)
        uint32_t ra[2]; uint2 rb; uint4 rc, rd;
#define SFP_I3 ((tid + kPB / 2) & (kPB - 1u))
#define SFP_I4 (kPB - 1u - tid)
        const uint2* __restrict__ r2p = reinterpret_cast<const uint2*>(recs + (n3 + n4 + n_ov));
        const uint32_t* __restrict__ r1p = reinterpret_cast<const uint32_t*>(r2p + n2);
        auto request_stream = [&]() {
            ra[0] = ra[1] = 0u; rb = make_uint2(0u, 0u); rc = rd = make_uint4(0u, 0u, 0u, 0u);
            if (tid < n1) ra[0] = r1p[SFP_IX(tid)];
            if (tid + kPB < n1) ra[1] = r1p[SFP_IX(tid + kPB)];
            if (tid < n2) rb = r2p[SFP_IX(tid)];
            if (SFP_I3 < n3) rc = recs[SFP_IX(SFP_I3)];
            if (SFP_I4 < n4) rd = recs[n3 + SFP_IX(SFP_I4)];
        };
        const uint32_t tg = a.tag0 + s;                                      // the tag of what sweep s - 1 published: epoch | s
        if (s > 0u) {
            const uint32_t rd_off = (s & 1u) ? a.part_off[1] : a.part_off[0];       // sums of sweep s - 1 carry tag s
            const uint32_t pos = lo + tid;                                   // (slot q of the thread: position pos + q * kPB)
            gr4 gq[kPS][3];
#pragma unroll
            for (int q = 0; q < kPS; ++q)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    gq[q][j] = gr4{0u, tg, 0u, tg};
                    if (flags[q] & (1u << j)) gq[q][j] = gr_load(rd_off, pos + q * kPB + delta[j]);
                }
            double len[kPS], av[kPS], own[kPS]; uint2 ft[kPS];
#pragma unroll
            for (int q = 0; q < kPS; ++q) {
                len[q] = 1.0; av[q] = 0.0; ft[q] = make_uint2(0u, 0u);
                if (has[q]) { len[q] = lenc[tid + q * kPB]; if (ftgt) ft[q] = ftgt[tid + q * kPB]; }
                if (home[q]) av[q] = alpha[tid + q * kPB];
                own[q] = acc[tid + q * kPB];                               // what this tile's last sweep handed the slot (cleared below)
            }
            // wave 0: has every tile finished update s - 1, and did the loop end there?  (:820)
            if (wave == 0u && s >= 2u) {
                const uint32_t u = s - 1u;
                const uint32_t visits = (u & 3u) ? (u >> 2) + 1u : (u >> 2);
                bool mv = false;
                if (lane < kShards) {
                    const unsigned long long want = (unsigned long long)visits * ((a.n_tiles + (kShards - 1u) - lane) / kShards);
                    const unsigned long long* w = &ctl[(kCtlArrive + (u & 3u) * kShards + lane) * kCtlStride];
                    unsigned long long got = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    SFP_WHY(1u);
                    if (a.ablate != 1) for (uint32_t spins = 0; (got & 0xFFFFFFFFull) < want;) { if (spin_check(spins, 1u)) break; got = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    // (ONE word says both: every arrival of a visit is in, and how many of them moved -- against the count of the visit before)
                    const uint32_t hi = (uint32_t)(got >> 32), k = (u & 3u) * kShards + lane;
                    mv = hi != hprev[k]; hprev[k] = hi;
                }
                const bool moved = __any(mv);
                if (lane == 0u) {
                    const bool conv = !moved;
                    const bool stop = u >= a.min_iter && (u >= a.max_iter || conv);
                    sctl[0] = stop ? (conv ? 3u : 1u) : 0u;
                }
            } else if (wave == 0u && lane == 0u) {
                const uint32_t u = s - 1u;                                    // (u = 0: nothing to wait for)
                sctl[0] = (u >= a.min_iter && u >= a.max_iter) ? 1u : 0u;
            }
            // the far slots that feed a transcript, in tile order: what alphaOut held in the other loops
#pragma unroll
            for (int q = 0; q < kPS; ++q) {
                if (has[q] && ft[q].y > ft[q].x) {
                    const uint32_t frd_off = a.far_off0 + ((s & 1u) ? a.far_stride : 0u);
                    for (uint32_t k = ft[q].x; k < ft[q].y; ++k) {
                        uint32_t g = ftg_l[tid + q * kPB];                 // (the first from the LDS: most targets have one)
                        if (k != ft[q].x) { SFP_COLD(cp); g = cp->ft_list[k]; }
                        gr4 w = gr_load(frd_off, g);
                        SFP_WHY(3u);
                        if (a.ablate != 1) for (uint32_t spins = 0; !gr_ok(w, tg);) { if (spin_check(spins, 1u)) break; w = gr_load(frd_off, g); }
                        ap_v[q] += gr_value(w);
                    }
                }
            }
            // the overlapping tiles' sums, in tile order with this tile's own in its place (as the cover list); three at a time
            // (a slot the thread does not have, or a tile that does not hold it, keeps the preset granule: its tag is right)
            auto wait3 = [&](uint32_t first) {
                if (a.ablate == 1) return;
                SFP_WHY(2u);
                for (uint32_t spins = 0;;) {
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < kPS; ++q) ok = ok && gr_ok(gq[q][0], tg) && gr_ok(gq[q][1], tg) && gr_ok(gq[q][2], tg);
                    if (ok || spin_check(spins, 1u)) break;
#pragma unroll
                    for (int q = 0; q < kPS; ++q)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            if (((flags[q] >> (first + j)) & 1u) && !gr_ok(gq[q][j], tg)) gq[q][j] = gr_load(rd_off, pos + q * kPB + delta[first + j]);
                }
            };
            wait3(0u);
#pragma unroll
            for (int q = 0; q < kPS; ++q)
#pragma unroll
                for (int j = 0; j < 3; ++j) { if ((uint32_t)j == nb_before) ap_v[q] += own[q]; if (flags[q] & (1u << j)) ap_v[q] += gr_value(gq[q][j]); }
            if (nb_n > 3u) {                                              // (block-uniform; most tiles overlap two or three others)
#pragma unroll
                for (int q = 0; q < kPS; ++q)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        gq[q][j] = gr4{0u, tg, 0u, tg};
                        if (flags[q] & (8u << j)) gq[q][j] = gr_load(rd_off, pos + q * kPB + delta[3 + j]);
                    }
                wait3(3u);
#pragma unroll
                for (int q = 0; q < kPS; ++q) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) { if ((uint32_t)(3 + j) == nb_before) ap_v[q] += own[q]; if (flags[q] & (8u << j)) ap_v[q] += gr_value(gq[q][j]); }
                    if (nb_before >= 6u) ap_v[q] += own[q];
                }
            } else if (nb_before >= 3u) {
#pragma unroll
                for (int q = 0; q < kPS; ++q) ap_v[q] += own[q];
            }
            SFP_STAMP(0);                                                 // operands here
#ifndef SFGPU_P_EARLY
            request_stream();
#endif
#pragma unroll
            for (int q = 0; q < kPS; ++q) {
                if (!has[q]) continue;
                if (VB) ap_v[q] += kPriorAlpha;
                xv[q] = x_of(ap_v[q], len[q]);
                if (home[q]) {
                    const double gate = a.check_mode ? av[q] : ap_v[q];   // :852 vs :499
                    if (gate > kCheckCutoff) {
#ifdef SFGPU_P_OLDHEAD
                        const double rel = fabs(av[q] - ap_v[q]) / ap_v[q];
#else
                        // (:852's gate makes ap > 1e-2: a reciprocal + Newton will do; :499's gate -- the bootstrap's -- is on the OLD alpha and
                        //  leaves ap anything >= 0, 0 included: the IEEE division, whose infinity counts as "moved" like the reference's)
                        const double rel = a.check_mode ? fabs(av[q] - ap_v[q]) / ap_v[q] : fabs(av[q] - ap_v[q]) * fast_rcp(ap_v[q]);
#endif
                        if (rel > lm) lm = rel;                            // NaN never wins, as in the reference (:854)
                        if (rel > a.tol) ncv = 1u;
                        if (lm < 0.0) lm = 0.0;                            // gated at least once
                    }
                    if (ft[q].y > ft[q].x) gr_store(a.far_off0 + 2u * a.far_stride, ft[q].x, xv[q], tg);     // a far target: its x for the tiles that hold it as a far member
                }
            }
            // what the wavefront saw of update s (tentative until the stop test of update s - 1 is known, behind the barrier); the largest
            // relative change stays in the thread's register and is reduced ONCE, when the loop has ended (round 6: six shuffle steps of a
            // double in every wavefront of every step before)
#ifdef SFGPU_P_OLDHEAD
            for (int o = kWave / 2; o > 0; o >>= 1) { const double m = __shfl_down(lm, o, kWave); if (m > lm) lm = m; }
            if (lane == 0u) wmax[(s & 1u) * kPWaves + wave] = lm;
#endif
            if (__any(ncv != 0u) && lane == 0u) sctl[2u + (s & 1u)] = 1u;
        } else {
#ifndef SFGPU_P_EARLY
            request_stream();
#endif
#pragma unroll
            for (int q = 0; q < kPS; ++q)
                if (has[q]) { SFP_COLD(cp); const uint32_t* inv = cp->inv; const uint32_t p = lo + tid + q * kPB; xv[q] = cp->x[inv ? inv[p] : p]; }
        }
        if (a.ablate == 3 && s == 2u && blockIdx.x == 0u && tid == 0u) {      // tests: tile 0 gives up here -- every tile must leave, the host repeats the run
            __hip_atomic_store(&ctl[kCtlAbort * kCtlStride], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sctl[1] = 1u;
        }
        for (uint32_t i = n1 + n2 + n3 + tid; i < nc; i += kPB) den[i] = 0.0;      // (B4 adds with atomics; every other class's word is written whole in phase A)
        if (nf) {                                                         // far members: the x of every far slot's transcript, once per step
            SFP_COLD(cp); SFP_TILE(tp);
            const uint32_t f0 = tp->f0;
            for (uint32_t f = tid; f < nf; f += kPB) { fxs[f] = far_x(cp, f0, f, s, 1u); facc[f] = 0.0; }
        }
        if (tid == 0u) { xs[kWin] = 0.0; acc[kWin] = 0.0; den[a.den_cap] = 0.0; }      // (den_cap: the plan's null class)
#pragma unroll
        for (int q = 0; q < kPS; ++q) if (has[q]) { xs[tid + q * kPB] = xv[q]; acc[tid + q * kPB] = 0.0; }
        SFP_STAMP(1);                                                     // x, the update, LDS cleared
        __syncthreads();
        SFP_STAMP(2);                                                     // head barrier
        if ((sctl[1] | sctl[4u + ((s + 1u) & 1u)]) != 0u) { k_done = 0xFFFFFFFFu; break; }      // some tile gave up (seen in this head, or in the phases of the step before): leave as one
        if (s > 0u) {
            const uint32_t sv = sctl[0];
            if (sv != 0u) { k_done = s - 1u; conv_last = (sv & 2u) != 0u; break; }
            // update s is final: alpha <- alpha' (home), and the block's word on it
#pragma unroll
            for (int q = 0; q < kPS; ++q) if (home[q]) alpha[tid + q * kPB] = ap_v[q];
            lm_keep = lm;
            if (tid == 0u) {
                const uint32_t shard = blockIdx.x & (kShards - 1u);
                unsigned long long inc = 1ull;
                if (sctl[2u + (s & 1u)] != 0u) { inc |= 1ull << 32; sctl[2u + (s & 1u)] = 0u; }
                __hip_atomic_fetch_add(&ctl[(kCtlArrive + (s & 3u) * kShards + shard) * kCtlStride], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ================= A: denominators =================
        uint4 pc_e[kPCAhead];                                            // phase C's first chunks: half of them requested here, half at the end of A
        {
#pragma unroll
            for (int i = 0; i < kPCAhead; ++i) pc_e[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int i = 0; i < kPCAhead / 2; ++i) { const uint32_t ch = tid + i * kPB; if (ch < np) pc_e[i] = SFP_LDC(pure + SFP_IX(ch)); }
            // (x of three window slots of a dword; the null slot kWin reads 0)
            auto sum3 = [&](uint32_t w) -> double { return (lds_f64(kLdsXs + ((w << 3) & 0x1FF8u)) + lds_f64(kLdsXs + ((w >> 7) & 0x1FF8u))) + lds_f64(kLdsXs + ((w >> 17) & 0x1FF8u)); };
            auto finish = [&](uint32_t c, double sum) {                          // :260-264; singletons carry the full count :275 / :364
                const uint32_t cwc = cntl[c];
                const double cn = (double)(cwc & 0x7FFFFFFFu);
#ifdef SFGPU_P_NOFIN                                                       // dev, timing only: no division per class
                den[c] = (cwc >> 31) ? cn : sum;
#else
                den[c] = (cwc >> 31) ? cn : ((sum > kTiny) ? cn / sum : 0.0);
#endif
            };
            if (tid < n1) finish(tid, sum3(ra[0]));
            if (tid + kPB < n1) finish(tid + kPB, sum3(ra[1]));
            for (uint32_t c = tid + 2u * kPB; c < n1; c += kPB) finish(c, sum3(r1p[SFP_IX(c)]));
            if (tid < n2) finish(n1 + tid, sum3(rb.x) + sum3(rb.y));
            for (uint32_t c = tid + kPB; c < n2; c += kPB) { const uint2 w = r2p[SFP_IX(c)]; finish(n1 + c, sum3(w.x) + sum3(w.y)); }
            auto sum12 = [&](const uint4& w) -> double { return (sum3(w.x) + sum3(w.y)) + (sum3(w.z) + sum3(w.w)); };
            if (SFP_I3 < n3) finish(n1 + n2 + SFP_I3, sum12(rc));
            for (uint32_t c = SFP_I3 + kPB; c < n3; c += kPB) finish(n1 + n2 + c, sum12(recs[SFP_IX(c)]));
            // B4: more than twelve in-window members (overflow chunks), or a far member: everything adds with atomics, phase B divides
            const uint32_t b4 = n1 + n2 + n3;
            if (SFP_I4 < n4) atomicAdd(&den[b4 + SFP_I4], sum12(rd));
            for (uint32_t c = SFP_I4 + kPB; c < n4; c += kPB) atomicAdd(&den[b4 + c], sum12(recs[n3 + SFP_IX(c)]));
            for (uint32_t j = tid; j < n_ov; j += kPB) atomicAdd(&den[ovcp[j]], sum12(recs[n3 + n4 + SFP_IX(j)]));
            // far members (few tiles have any): class and far slot from the plan, x from the LDS copy the head made
            if (n_esc) {
                for (uint32_t i = tid; i < n_esc; i += kPB) {
                    uint2 e;
                    if (i < a.esc_ln) e = esc_l[i]; else { SFP_COLD(cp); SFP_TILE(tp); const uint64_t e0 = tp->e0; e = make_uint2(cp->esc_cls[e0 + i], cp->esc_far[e0 + i]); }
                    const uint32_t tag = e.x, f = e.y;
                    const double v = (tag & kSingle) ? 0.0 : fxs[f];
                    if (v != 0.0) atomicAdd(&den[(tag >> 16) & 0x1FFFu], v);
                }
            }
#pragma unroll
            for (int i = kPCAhead / 2; i < kPCAhead; ++i) { const uint32_t ch = tid + i * kPB; if (ch < np) pc_e[i] = SFP_LDC(pure + SFP_IX(ch)); }
        }
        __syncthreads();
        SFP_STAMP(3);                                                     // phase A + its barrier
        // ================= B: count / denom per class (:260-264; singletons carry the full count :275 / :364) =================
        {
            // B4 only (a singleton among them: its one member is a far one)
            for (uint32_t c = n1 + n2 + n3 + tid; c < nc; c += kPB) {
                const uint32_t cwc = cntl[c];
                const double cn = (double)(cwc & 0x7FFFFFFFu);
                const double d = den[c];
                den[c] = (cwc >> 31) ? cn : ((d > kTiny) ? cn / d : 0.0);
            }
        }
        __syncthreads();
        SFP_STAMP(4);                                                     // phase B + its barrier
        // ================= C: the window (a gather over the transcript-major copy) =================
        {
            auto pure_chunk = [&](const uint4& e4) {                              // eight classes of ONE slot (k_cscp_build): slot and singleton bit in the spare bits
                const double q0 = lds_f64(kLdsDen + ((e4.x << 3) & 0xFFF8u)), q1_ = lds_f64(kLdsDen + ((e4.x >> 10) & 0xFFF8u)), q2_ = lds_f64(kLdsDen + ((e4.y << 3) & 0xFFF8u)), q3 = lds_f64(kLdsDen + ((e4.y >> 10) & 0xFFF8u));
                const double q4 = lds_f64(kLdsDen + ((e4.z << 3) & 0xFFF8u)), q5 = lds_f64(kLdsDen + ((e4.z >> 10) & 0xFFF8u)), q6 = lds_f64(kLdsDen + ((e4.w << 3) & 0xFFF8u)), q7 = lds_f64(kLdsDen + ((e4.w >> 10) & 0xFFF8u));
                const double sum = ((q0 + q1_) + (q2_ + q3)) + ((q4 + q5) + (q6 + q7));
                const uint32_t slot = (e4.x >> 26) | (((e4.y >> 26) & 15u) << 6);
                const double v = (e4.y >> 31) ? sum : xs[slot] * sum;
#ifdef SFGPU_P_NOATOMC                                                     // dev, timing only: a plain store where phase C adds
                if (v != 0.0) acc[slot] = v;
#else
                if (v != 0.0) atomicAdd(&acc[slot], v);
#endif
            };
#pragma unroll
            for (int i = 0; i < kPCAhead; ++i) if (tid + i * kPB < np) pure_chunk(pc_e[i]);
            for (uint32_t ch = tid + kPCAhead * kPB; ch < np; ch += kPB) pure_chunk(SFP_LDC(pure + SFP_IX(ch)));
            if (n_esc) {                                                     // far members: into the tile's far slots
                for (uint32_t i = tid; i < n_esc; i += kPB) {
                    uint2 e;
                    if (i < a.esc_ln) e = esc_l[i]; else { SFP_COLD(cp); SFP_TILE(tp); const uint64_t e0 = tp->e0; e = make_uint2(cp->esc_cls[e0 + i], cp->esc_far[e0 + i]); }
                    const uint32_t tag = e.x, f = e.y;
                    const double q = den[(tag >> 16) & 0x1FFFu];
                    const double contrib = (tag & kSingle) ? q : fxs[f] * q;
                    if (contrib != 0.0) atomicAdd(&facc[f], contrib);
                }
            }
        }
        __syncthreads();
        SFP_STAMP(5);                                                     // phase C + its barrier
        // ================= D: publish the window and the far slots: granules with tag s + 1 =================
#pragma unroll
        for (int q = 0; q < kPS; ++q) if (has[q]) gr_store((s & 1u) ? a.part_off[0] : a.part_off[1], off + tid + q * kPB, acc[tid + q * kPB], tg + 1u);
        if (nf) {
            SFP_TILE(tp);
            const uint32_t fo = a.far_off0 + ((s & 1u) ? 0u : a.far_stride), f0 = tp->f0;
            for (uint32_t f = tid; f < nf; f += kPB) gr_store(fo, f0 + f, facc[f], tg + 1u);
        }
        SFP_STAMP(6);                                                     // phase D (stores drained)
        // (no barrier here: what the next head clears or writes before its own barrier -- xs[tid], acc[tid], den, facc[f] -- was last read
        //  in phase C, behind the barrier above, or is this thread's own slot of phase D)
    }

    // ================= the loop has ended =================
    const uint32_t tid = tid0, lane = tid & (kWave - 1), wave = tid / kWave;
    SFP_COLD(cp);
#ifdef SFGPU_P_STAMP
    if (cp->dbg && tid == 0u) for (int k = 0; k < 7; ++k) cp->dbg[blockIdx.x * 8 + k] = pst[k];
#endif
    if (k_done == 0xFFFFFFFFu) { if (tid == 0u) *cp->status = 1u; return; }
    if (k_done > 0u) {
#ifdef SFGPU_P_OLDHEAD
        const double wm = wmax[(k_done & 1u) * kPWaves + wave];
#else
        double wm = lm_keep;                                             // (update k_done was the last one declared final: its values are what the threads kept)
        for (int o = kWave / 2; o > 0; o >>= 1) { const double m = __shfl_down(wm, o, kWave); if (m > wm) wm = m; }
#endif
        // (the table has a row of kSweepBlock / kWave entries per tile, as the other loops fill it: a wavefront writes its maximum into every
        //  kPWaves-th of them)
        if (lane == 0u) for (uint32_t w2 = wave; w2 < (uint32_t)(kSweepBlock / kWave); w2 += kPWaves)
            cp->tmax[((uint64_t)((k_done - 1u) & 1u) * gridDim.x + blockIdx.x) * (kSweepBlock / kWave) + w2] = wm;
        // positions no window holds are inactive transcripts here (a plan with far-only transcripts does not run persistent):
        // every update leaves them at the prior (VBEM, :318) or at 0
        const uint32_t n_unc = *cp->unc_n;
        for (uint32_t j = tid * gridDim.x + blockIdx.x; j < n_unc; j += kPB * gridDim.x) a.alpha[cp->unc[j]] = VB ? kPriorAlpha : 0.0;
    }
    if (blockIdx.x == 0u && tid == 0u) {
        EmState* st = cp->st;
        st->it_a = k_done; st->itv[0] = st->itv[1] = k_done;
        if (k_done > 0u) st->notconv3[(k_done - 1u) % 3u] = conv_last ? 0u : 1u;
    }
}
