// eqclass.hip -- EquivalenceClassBuilder on the device (rows a1-a5 of SURVEY.md section 8).
//
// Replaces  include/EquivalenceClassBuilder.hpp:53-119  (addGroup -> libcuckoo upsert,
//           include/cuckoohash_map.hh:667-698), include/TranscriptGroup.hpp and
//           src/TranscriptGroup.cpp (key = ordered id list, hash = XXH64, == is vector ==).
//
// Design (not the reference's: that is a lock-striped CPU cuckoo table fed one read at a time):
//   * reads arrive as packed batches (ids[], offsets[]) resident in HBM (host batches are accumulated in
//     pinned memory and staged, see sfgpu_eq_add_batch_host);
//   * the table is open addressing with linear probing inside REGIONS of 4096 slots; a slot is one
//     16-byte pair {word, count}: word = tag(32) | rep(32).  Where a label lands (region, slot, tag) is
//     decided by a cheap bucket hash (xxh64_device.h); tag only filters probes -- class identity is
//     ALWAYS decided by a full label compare against the slot's representative label, exactly like
//     operator== (src/TranscriptGroup.cpp:53-55); XXH64 (TranscriptGroup::hash) is computed once per
//     class when it is committed and is what the export and the canonical order use;
//   * big unweighted batches go through the radix-partitioned kernels of eqclass_part.h (one block per
//     region, region image in LDS); weighted merges, small batches, over-long labels and overflow take
//     the generic kernel k_insert below: one lane per read probing the table in HBM, claiming a slot
//     with a single 64-bit CAS that publishes tag and representative together, so no lane ever waits on
//     another lane (no spin, no fence): rep is either a read index of the running batch (label still in
//     the batch buffers, written before the launch) or 0x80000000 | arena entry / 4 (label in the arena,
//     written by an earlier launch or by the committing block);
//   * committed labels live in the arena as 16-byte-aligned entries [len, ids..., 0 pad]: one 16-byte
//     load decides a compare for labels of <= 3 ids;
//   * the table is sized from the number of classes actually seen (load <= 1/2 partitioned, <= 1/2
//     generic); inserts that would exceed the budget are deferred, the table is doubled, and the
//     deferred reads are replayed;
//   * finish() orders classes canonically (first id, XXH64, length, label) with a radix sort.
#include <mutex>
#include <vector>

#include "common.h"
#include "primitives.h"
#include <atomic>
#include <thread>
#include "xxh64_device.h"

namespace sfgpu {

constexpr uint64_t kEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t kArenaBit = 0x80000000u;
constexpr uint64_t kSlack = 1ull << 20;      // >= lanes in flight that can pass the budget guard together
constexpr uint64_t kMinHeadroom = 1ull << 16;
constexpr int kBlock = 256;

// ctr[0] = classes created by this launch, ctr[1] = deferred reads, ctr[2] = arena cursor (words),
// ctr[3] = scratch (long-label count / read total), ctr[4] = nnz read back at finish
enum { CTR_NEW = 0, CTR_DEFER = 1, CTR_ARENA = 2, CTR_TMP = 3, CTR_NNZ = 4, CTR_PEEK = 5, CTR_HOT = 8, CTR_GCLS = 9, CTR_N = 12 };     // CTR_PEEK..+2: offsets read ahead for the host; CTR_GCLS: classes committed (the partition passes number new classes from it)

// ---- label arena ------------------------------------------------------------------------------
// A committed class keeps its label in the arena as one ENTRY: [len, id0, id1, ...], zero-padded to a
// multiple of 4 words and 16-byte aligned.  A table slot of a committed class holds kArenaBit | entry/4,
// so deciding "is this read's label the slot's label" is ONE 16-byte load for labels of <= 3 ids and two
// for <= 7 -- no class-id indirection, no per-word loads (they bound the insert kernels: a wavefront's 64
// lanes hit 64 different cache lines per load instruction).  cls_off[c] = entry + 1 (the first id).
__host__ __device__ __forceinline__ uint32_t entry_words(uint32_t len) { return (len + 4u) & ~3u; }

// label (n words; first 8 in hw[], zero past n; word(k) for the rest) == arena entry e?
template <typename WordFn>
__device__ __forceinline__ bool entry_equals(const uint32_t* __restrict__ arena, uint32_t e, WordFn word,
                                             const uint32_t (&hw)[kHead], uint32_t n) {
    const uint32_t* p = arena + ((uint64_t)e << 2);
    // both 16-byte words are requested together and compared without short-circuits: one round trip
    // (a 4-word entry is followed by the next entry or by the arena's slack, so p[4..7] is readable)
    const uint4 a = reinterpret_cast<const uint4*>(p)[0];
#ifdef SFGPU_EQ_COND_SECOND
    uint32_t d = (a.x ^ n) | (a.y ^ hw[0]) | (a.z ^ hw[1]) | (a.w ^ hw[2]);
    if (d) return false;
    if (n > 3u) {
        const uint4 b = reinterpret_cast<const uint4*>(p)[1];
        if ((b.x ^ hw[3]) | (b.y ^ hw[4]) | (b.z ^ hw[5]) | (b.w ^ hw[6])) return false;
    }
#else
    const uint4 b = reinterpret_cast<const uint4*>(p)[1];
    uint32_t d = (a.x ^ n) | (a.y ^ hw[0]) | (a.z ^ hw[1]) | (a.w ^ hw[2]);
    const uint32_t d2 = (b.x ^ hw[3]) | (b.y ^ hw[4]) | (b.z ^ hw[5]) | (b.w ^ hw[6]);
    d |= (n > 3u) ? d2 : 0u;
    if (d) return false;
#endif
    for (uint32_t k = 7; k < n; ++k) if (p[1 + k] != word(k)) return false;
    return true;
}
// write entry [len, words..., 0 pad] at arena word offset dst (a multiple of 4)
template <typename WordFn>
__device__ __forceinline__ void entry_write(uint32_t* arena, uint64_t dst, WordFn word, uint32_t len) {
    arena[dst] = len;
    for (uint32_t k = 0; k < len; ++k) arena[dst + 1 + k] = word(k);
    for (uint32_t k = len + 1; k < entry_words(len); ++k) arena[dst + k] = 0u;
}

// ---- the PROBE granule of a class (round 4) ------------------------------------------------------------------------------
// The partition stream carries a label of 4 .. 9 ids in ONE 16-byte granule when it can: ids below 2^24 that ascend in steps of
// 0 .. 255 (the isoforms of a gene in annotation order) travel as  [head | compact | id0][H][d1 d2 d3 d4][d5 d6 d7 d8]  -- the
// first id and eight 8-bit steps.  92 % of the benchmark's labels are then one granule (58 % with <= 3 ids, which always were),
// pass 1 writes one 16-byte store for them and pass 2 decides their identity in LDS without touching the arena: with every label
// cut to 3 ids the two passes took 8.7 ms instead of 14.6 (profiles/r4_class_build_notes.md).  The encoding is a function of the
// label alone and injective among labels of one length, so comparing encodings compares labels; everything else (more ids, wider
// steps, unsorted ids, ids >= 2^24) keeps the multi-granule form [id0][H][id1][id2] [id3 ..] ...
// Pass 2 keeps, per class, the three payload words of the label's FIRST stream granule.  For committed classes they sit in the
// arena in a granule of their own right in front of the entry:  [len, p0, p1, p2] [len, id0, id1, id2] [id3 ..] ... ; the table's
// rep still addresses the entry, so every other reader of the arena is unchanged.
constexpr uint32_t kCompactBit = 0x40000000u;                 // in word 0 of a head granule (ids >= 2^30 take the generic kernel)
constexpr uint32_t kMaxCompactLen = 9;
// ... and a second compact form for 10 .. 17 ids: sixteen 4-bit steps (0 .. 15).  With the first form alone 7.5 % of the
// benchmark's labels kept their tails, and since some lane of nearly every 64-read step then holds one, they cost the two passes
// 2.6 of 12.2 ms (profiles/r4_class_build_notes.md: labels cut to 9 ids).  Bit 29 of word 0 tells the forms apart (a compact
// id0 is below 2^24).
constexpr uint32_t kCompact4Bit = 0x20000000u;
constexpr uint32_t kMaxCompact4Len = 17;
constexpr uint32_t kProbeWords = 4;                           // arena words in front of an entry
template <typename WordFn>
__device__ __forceinline__ void label_probe(WordFn word, uint32_t n, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    const uint32_t id0 = word(0);
    if (n >= 4u && n <= kMaxCompactLen && id0 < (1u << 24)) {
        uint32_t prev = id0, lo = 0, hi = 0;
        bool ok = true;
        for (uint32_t k = 1; k < n; ++k) {
            const uint32_t v = word(k), d = v - prev;
            ok = ok && d <= 255u;
            prev = v;
            if (k <= 4u) lo |= (d & 255u) << (8u * (k - 1u)); else hi |= (d & 255u) << (8u * (k - 5u));
        }
        if (ok) { p0 = kCompactBit | id0; p1 = lo; p2 = hi; return; }
    } else if (n > kMaxCompactLen && n <= kMaxCompact4Len && id0 < (1u << 24)) {
        uint32_t prev = id0, lo = 0, hi = 0;
        bool ok = true;
        for (uint32_t k = 1; k < n; ++k) {
            const uint32_t v = word(k), d = v - prev;
            ok = ok && d <= 15u;
            prev = v;
            if (k <= 8u) lo |= (d & 15u) << (4u * (k - 1u)); else hi |= (d & 15u) << (4u * (k - 9u));
        }
        if (ok) { p0 = kCompactBit | kCompact4Bit | id0; p1 = lo; p2 = hi; return; }
    }
    // (ids >= 2^30 never enter the stream -- such reads take the generic kernel -- and must not look like a compact granule here)
    p0 = id0 < kCompactBit ? id0 : 0xFFFFFFFFu; p1 = n > 1u ? word(1) : 0u; p2 = n > 2u ? word(2) : 0u;
}
// id k of a label held as ONE compact granule (c0 = word 0 without the head bit, lo / hi = words 2 and 3)
__device__ __forceinline__ uint32_t compact_id(uint32_t c0, uint32_t lo, uint32_t hi, uint32_t k) {
    uint32_t v = c0 & 0xFFFFFFu;
    if (c0 & kCompact4Bit) { for (uint32_t j = 1; j <= k; ++j) v += (j <= 8u ? (lo >> (4u * (j - 1u))) : (hi >> (4u * (j - 9u)))) & 15u; }
    else for (uint32_t j = 1; j <= k; ++j) v += (j <= 4u ? (lo >> (8u * (j - 1u))) : (hi >> (8u * (j - 5u)))) & 255u;
    return v;
}
// arena words a class takes: probe granule + entry
__host__ __device__ __forceinline__ uint32_t class_words(uint32_t len) { return kProbeWords + entry_words(len); }
template <typename WordFn>
__device__ __forceinline__ void probe_write(uint32_t* arena, uint64_t dst, WordFn word, uint32_t len) {
    uint32_t p0, p1, p2;
    label_probe(word, len, p0, p1, p2);
    arena[dst] = len; arena[dst + 1] = p0; arena[dst + 2] = p1; arena[dst + 3] = p2;
}

__global__ void k_table_init(uint64_t* table, uint64_t cap) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < cap; i += stride) { table[2 * i] = kEmpty; table[2 * i + 1] = 0; }
}

__global__ void k_rebase(uint32_t* off, uint64_t n, uint32_t base) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) off[i] -= base;
}

__device__ __forceinline__ bool labels_equal(const uint32_t* a, const uint32_t* b, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i)
        if (a[i] != b[i]) return false;
    return true;
}

}  // namespace sfgpu
#ifdef SFGPU_X_EQ_STAMP
namespace sfgpu { __device__ unsigned long long g_eq_stamp[2][2][4096]; }      // dev: [route | insert][start | end][block], 100 MHz clock
#endif
#include "eqclass_part.h"
namespace sfgpu {

// a1: standalone label hash (sfgpu_xxh64_labels)
__global__ void k_hash_labels(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ off, uint32_t n,
                              uint64_t* __restrict__ out) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint32_t b = off[r], len = off[r + 1] - b;
    const uint32_t* lab = ids + b;
    uint32_t hw[kHead];
    out[r] = xxh64_label([&](uint32_t k) { return lab[k]; }, len, hw);
}

// addGroup for a range of reads (list == nullptr: reads [first, first+n); else list[0..n)).
__global__ void __launch_bounds__(kBlock)
k_insert(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ off, uint32_t first, uint32_t n,
         const uint32_t* __restrict__ list, uint64_t* table, uint64_t mask,
         const uint64_t* __restrict__ cls_off, const uint32_t* __restrict__ cls_len,
         const uint32_t* __restrict__ arena, unsigned long long* ctr, uint32_t* newlist,
         unsigned long long limit_new, uint32_t* deferred, const uint64_t* __restrict__ weights, uint32_t mix_mode) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = i < n;
    uint32_t r = in_range ? (list ? list[i] : first + i) : 0u;
    unsigned long long inc = (in_range && weights) ? (unsigned long long)weights[r] : 1ull;
    uint32_t b = 0, len = 0;
    if (in_range) { b = off[r]; len = off[r + 1] - b; }
    const bool act = in_range && len != 0;             // call-site guard: empty hit lists never reach addGroup
    const uint32_t* lab = ids + b;
    uint32_t hw[kHead];
    uint64_t h = 0;
    if (act) h = label_mix64([&](uint32_t k) { return lab[k]; }, len, hw, mix_mode);     // bucket hash (xxh64_device.h)
    // Identical labels of one wavefront are added once, with their counts summed: the reads this kernel sees are often
    // dominated by a few labels (the overflow of a hot label's bins, see k_part_route), and 64 atomics on one slot are 64
    // serialised L2 round trips.  Two rounds: the first two distinct labels of the wavefront collect their duplicates.
    bool absorbed = false, probing = false;
    int leader_of = 0;
    {
        const uint32_t lane = threadIdx.x & (kWave - 1);
        unsigned long long todo = __ballot(act);
        for (int round = 0; round < 2 && todo; ++round) {
            const int leader = __builtin_ctzll(todo);
            const uint64_t hl = __shfl(h, leader, kWave);
            const uint32_t ll = __shfl(len, leader, kWave), bl = __shfl(b, leader, kWave);
            const bool cand = act && !absorbed && (int)lane != leader && ((todo >> lane) & 1ull) && h == hl && len == ll;
            const bool same = cand && labels_equal(ids + bl, lab, len);
            const unsigned long long m = __ballot(same);
            if (m) {
                unsigned long long add = 0;
                if (!weights) add = (unsigned long long)__builtin_popcountll(m);
                else for (unsigned long long q = m; q; q &= q - 1) add += __shfl(inc, __builtin_ctzll(q), kWave);
                if ((int)lane == leader) inc += add;
                if (same) { absorbed = true; leader_of = leader; }
            }
            todo &= ~(m | (1ull << leader));
        }
        // (no lane leaves before the end: an absorbed read follows its leader if the leader is deferred)
        probing = act && !absorbed;
    }
    uint64_t tag = h >> 32;
    uint64_t s = h & mask;
    uint32_t probes = 0;
    bool was_deferred = false;
    while (probing) {
        uint64_t w = __hip_atomic_load(&table[2 * s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == kEmpty) {
            if (__hip_atomic_load(&ctr[CTR_NEW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= limit_new) { was_deferred = true; break; }
            uint64_t mine = (tag << 32) | (uint64_t)r;
            unsigned long long old = atomicCAS((unsigned long long*)&table[2 * s], (unsigned long long)kEmpty,
                                               (unsigned long long)mine);
            if (old == kEmpty) {
                unsigned long long idx = atomicAdd(&ctr[CTR_NEW], 1ull);
                newlist[idx] = (uint32_t)s;
                atomicAdd((unsigned long long*)&table[2 * s + 1], inc);
                break;
            }
            w = old;
        }
        if ((w >> 32) == tag) {
            uint32_t rep = (uint32_t)w;
            bool same;
            if (rep & kArenaBit) same = entry_equals(arena, rep & ~kArenaBit, [&](uint32_t k) { return lab[k]; }, hw, len);
            else { uint32_t rb = off[rep]; same = (off[rep + 1] - rb == len) && labels_equal(ids + rb, lab, len); }
            if (same) {
                atomicAdd((unsigned long long*)&table[2 * s + 1], inc);
                break;
            }
        }
        if (++probes >= kRegionSlots) { was_deferred = true; break; }   // home region full: replay after the table has grown
        s = region_next(s);
    }
    // deferred reads are replayed one by one with their own weights: a deferred leader takes the reads it absorbed along
    // (the shuffle runs in EVERY lane, before the test: inside `absorbed && shfl(...)` it would execute with the leader's lane
    //  masked off -- leaders are never absorbed -- and read 0 from it: the followers of a deferred leader were then dropped.
    //  Found in round 4 by test_builder_long_labels_that_agree_in_every_sampled_id: 11 of 40 000 reads lost.)
    const bool leader_deferred = __shfl((int)was_deferred, leader_of, kWave) != 0;
    const bool follow = absorbed && leader_deferred;
    if (was_deferred || follow) {
        unsigned long long d = atomicAdd(&ctr[CTR_DEFER], 1ull);
        deferred[d] = r;
    }
}

// move the labels of the classes created by the last k_insert into the arena
__global__ void k_commit(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ off,
                         uint64_t* table, const uint32_t* __restrict__ newlist, uint64_t n_new, uint64_t base_cid,
                         uint64_t* cls_hash, uint64_t* cls_off, uint32_t* cls_len, uint32_t* cls_slot,
                         uint32_t* arena, unsigned long long* ctr, uint32_t* probe) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_new) return;
    uint32_t s = newlist[i];
    uint64_t w = table[2 * (uint64_t)s];
    uint32_t r = (uint32_t)w;
    uint32_t b = off[r], len = off[r + 1] - b;
    const uint32_t* lab = ids + b;
    unsigned long long dst = atomicAdd(&ctr[CTR_ARENA], (unsigned long long)class_words(len));
    probe_write(arena, dst, [&](uint32_t k) { return lab[k]; }, len);
    reinterpret_cast<uint4*>(probe)[s] = make_uint4(arena[dst], arena[dst + 1], arena[dst + 2], arena[dst + 3]);
    dst += kProbeWords;
    entry_write(arena, dst, [&](uint32_t k) { return lab[k]; }, len);
    uint64_t cid = base_cid + i;
    cls_hash[cid] = xxh64_words([&](uint32_t k) { return lab[k]; }, len);
    cls_off[cid] = dst + 1; cls_len[cid] = len; cls_slot[cid] = s;
    table[2 * (uint64_t)s] = (w & 0xFFFFFFFF00000000ull) | (uint64_t)(kArenaBit | (uint32_t)(dst >> 2));
}

// re-insert every class into a larger table, carrying its count
__global__ void k_rehash(const uint64_t* __restrict__ old_table, uint64_t* table, uint64_t mask, uint64_t n_cls,
                         const uint64_t* __restrict__ cls_off, const uint32_t* __restrict__ cls_len,
                         const uint32_t* __restrict__ arena, uint32_t* cls_slot, uint32_t mix_mode, uint32_t* probe) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cls) return;
    const uint32_t* lab = arena + cls_off[c];
    uint32_t hw[kHead];
    uint64_t h = label_mix64([&](uint32_t k) { return lab[k]; }, cls_len[c], hw, mix_mode);
    uint64_t mine = ((h >> 32) << 32) | (uint64_t)(kArenaBit | (uint32_t)((cls_off[c] - 1) >> 2));
    uint64_t cnt = old_table[2 * (uint64_t)cls_slot[c] + 1];
    uint64_t s = h & mask;
    for (;;) {
        unsigned long long old = atomicCAS((unsigned long long*)&table[2 * s], (unsigned long long)kEmpty,
                                           (unsigned long long)mine);
        if (old == kEmpty) break;
        s = region_next(s);
    }
    table[2 * s + 1] = cnt;
    cls_slot[c] = (uint32_t)s;
    reinterpret_cast<uint4*>(probe)[s] = *reinterpret_cast<const uint4*>(arena + cls_off[c] - 1 - kProbeWords);      // the class's probe granule moves with its slot
}

// ---- finish(): canonical order ---------------------------------------------------------------
// sort key = first id << 14 | the top 14 bits of XXH64: with the ids of a transcriptome (< 2^18) that is a 32-bit key -- four
// radix passes instead of the eight of (first id << 32 | hash >> 32); classes that share a key (same first id, same 14 bits:
// a few per thousand) are put into (hash, length, label) order by k_tie_fix.  max_first: the largest first id (sizes the sort).
constexpr int kSortHashBits = 14;
// The same pass adds up the classes' counts (total reads: one more scattered read per class next to the arena's, instead of a
// pass of its own over the table later).
__global__ void k_sort_keys(uint64_t n, const uint64_t* __restrict__ cls_hash, const uint64_t* __restrict__ cls_off,
                            const uint32_t* __restrict__ arena, uint64_t* keys, uint32_t* vals, unsigned long long* max_first,
                            const uint64_t* __restrict__ table, const uint32_t* __restrict__ cls_slot, unsigned long long* total) {
    // grid-stride over a capped grid: at most one atomic per wavefront of a few thousand blocks, not one per 64 classes (25 k atomics
    // on one address took 250 us)
    uint32_t mx = 0;
    unsigned long long sum = 0;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t first = arena[cls_off[c]];
        const unsigned long long cnt = table[2 * (uint64_t)cls_slot[c] + 1];
        keys[c] = ((uint64_t)first << kSortHashBits) | (cls_hash[c] >> (64 - kSortHashBits));
        vals[c] = (uint32_t)c;
        mx = first > mx ? first : mx;
        sum += cnt;
    }
    for (int o = kWave / 2; o > 0; o >>= 1) { const uint32_t v = __shfl_down(mx, o, kWave); mx = v > mx ? v : mx; sum += __shfl_down(sum, o, kWave); }
    // one atomic per BLOCK for the sum (every block has one to add: thousands of same-address atomics are served one after the other)
    __shared__ unsigned long long wsum[kBlock / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if ((unsigned long long)mx > __hip_atomic_load(max_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(max_first, (unsigned long long)mx);   // (looks first: the running maximum is soon above most wavefronts')
        wsum[threadIdx.x / kWave] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < kBlock / kWave; ++i) t += wsum[i];
        if (t) atomicAdd(total, t);
    }
}

__device__ bool class_less(uint32_t a, uint32_t b, const uint64_t* cls_hash, const uint64_t* cls_off,
                           const uint32_t* cls_len, const uint32_t* arena) {
    if (cls_hash[a] != cls_hash[b]) return cls_hash[a] < cls_hash[b];
    if (cls_len[a] != cls_len[b]) return cls_len[a] < cls_len[b];
    const uint32_t* p = arena + cls_off[a]; const uint32_t* q = arena + cls_off[b];
    for (uint32_t i = 0; i < cls_len[a]; ++i)
        if (p[i] != q[i]) return p[i] < q[i];
    return false;
}

// runs of equal sort keys (first id, top bits of the hash) are ordered by (hash, len, label); such runs are
// short, so the thread at the head of a run insertion-sorts it.
__global__ void k_tie_fix(uint64_t n, const uint64_t* __restrict__ keys, uint32_t* order,
                          const uint64_t* __restrict__ cls_hash, const uint64_t* __restrict__ cls_off,
                          const uint32_t* __restrict__ cls_len, const uint32_t* __restrict__ arena) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i > 0 && keys[i - 1] == keys[i]) return;        // not a run head
    uint64_t e = i + 1;
    while (e < n && keys[e] == keys[i]) ++e;
    for (uint64_t a = i + 1; a < e; ++a) {
        uint32_t v = order[a]; uint64_t b = a;
        while (b > i && class_less(v, order[b - 1], cls_hash, cls_off, cls_len, arena)) { order[b] = order[b - 1]; --b; }
        order[b] = v;
    }
}

__global__ void k_sorted_lens(uint64_t n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ cls_len,
                              uint32_t* lens) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lens[i] = cls_len[order[i]];
    else if (i == n) lens[i] = 0;
}

// grid-stride: a few hundred same-address atomics in all (one per wavefront of a capped grid), not one per 64 classes
__global__ void k_sum_counts(uint64_t n, const uint64_t* __restrict__ table, const uint32_t* __restrict__ cls_slot,
                             unsigned long long* total) {
    unsigned long long v = 0;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (uint64_t)gridDim.x * blockDim.x)
        v += table[2 * (uint64_t)cls_slot[c] + 1];
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && v) atomicAdd(total, v);
}

// export, per class: rowptr, count, hash in the canonical order
__global__ void k_export_classes(uint64_t n, const uint32_t* __restrict__ order, const uint64_t* __restrict__ rowptr64,
                                 const uint64_t* __restrict__ table, const uint32_t* __restrict__ cls_slot,
                                 const uint64_t* __restrict__ cls_hash, uint32_t* rowptr, uint64_t* counts, uint64_t* hashes) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    rowptr[i] = (uint32_t)rowptr64[i];
    if (i == n) return;
    const uint32_t c = order[i];
    counts[i] = table[2 * (uint64_t)cls_slot[c] + 1];
    if (hashes) hashes[i] = cls_hash[c];
}
// export, per ID: a wavefront writes kExportPerWave consecutive positions of `ids`, 64 per step, coalesced; the class of a
// position comes from the class starts the lanes hold side by side (64 consecutive positions meet at most 64 of them: searched
// with shuffles), its label from the arena (consecutive lanes of a class read consecutive words).  A lane per class that copies
// its whole label wrote 4 bytes here and 4 bytes there: 240 us on cfg3, this form ~60.
constexpr uint32_t kExportPerWave = 1024;
__global__ void __launch_bounds__(kBlock)
k_export_ids(uint64_t n, uint64_t nnz, const uint32_t* __restrict__ order, const uint64_t* __restrict__ rowptr64,
             const uint64_t* __restrict__ cls_off, const uint32_t* __restrict__ arena, uint32_t* ids) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint64_t p_begin = wave * kExportPerWave, p_end = p_begin + kExportPerWave < nnz ? p_begin + kExportPerWave : nnz;
    if (p_begin >= nnz) return;
    uint64_t cf;                                                // class of position p0: the last one with rowptr64[c] <= p0
    { uint64_t a = 0, b = n; while (b - a > 1) { const uint64_t mid = (a + b) >> 1; if (rowptr64[mid] <= p_begin) a = mid; else b = mid; } cf = a; }
    for (uint64_t p0 = p_begin; p0 < p_end; p0 += kWave) {
        if (cf + 1 < n && rowptr64[cf + 1] <= p0) ++cf;         // (position p0 - 1 was of class cf or cf - 1 ... see below)
        const uint64_t pos = p0 + lane;
        const uint64_t ia = cf + lane;
        const uint64_t A = ia <= n ? rowptr64[ia] : ~0ull, B = ia + 1 <= n ? rowptr64[ia + 1] : ~0ull;
        uint32_t r = 0;                                         // how many of the starts B_0 <= B_1 <= ... are <= pos (at most 63)
#pragma unroll
        for (uint32_t step = kWave / 2; step; step >>= 1) {
            const uint32_t q = r + step - 1u;
            const uint64_t probe = ((uint64_t)__shfl((uint32_t)(B >> 32), (int)q, kWave) << 32) | __shfl((uint32_t)B, (int)q, kWave);
            if (probe <= pos) r += step;
        }
        const uint64_t a_r = ((uint64_t)__shfl((uint32_t)(A >> 32), (int)r, kWave) << 32) | __shfl((uint32_t)A, (int)r, kWave);
        if (pos < p_end) {
            const uint32_t c = order[cf + r];
            ids[pos] = arena[cls_off[c] + (pos - a_r)];
        }
        const uint32_t r_last = __shfl(r, kWave - 1, kWave);    // lane 63's class becomes the next step's first guess
        cf += r_last;
        if (cf >= n) cf = n - 1;
    }
}

static inline unsigned grid_for(uint64_t n, int block = kBlock) { return (unsigned)((n + block - 1) / block); }

}  // namespace sfgpu

using namespace sfgpu;

struct sfgpu_eq {
    hipStream_t stream = nullptr;
    std::mutex mu;
    bool finished = false;
    uint64_t expected = 0;
    uint64_t cap = 0;                     // slots (power of two)
    DevBuf<uint64_t> table;               // 2*cap
    DevBuf<uint32_t> probe;               // 4*cap: the probe granule [len, p0, p1, p2] of the class in each occupied slot, slot-indexed, so that
                                          // pass 2 loads a region's probes with coalesced reads (the arena keeps a copy in front of every entry)
    uint64_t n_classes = 0;
    DevBuf<uint64_t> cls_hash, cls_off;
    DevBuf<uint32_t> cls_len, cls_slot;
    DevBuf<uint32_t> arena; uint64_t arena_used = 0;
    DevBuf<uint32_t> newlist, deferred_a, deferred_b;
    unsigned long long* d_ctr = nullptr;
    unsigned long long* h_ctr = nullptr;  // pinned
    DevBuf<uint32_t> stage_ids, stage_off;
    DevBuf<uint32_t> stage2_ids[2], stage2_off[2];          // large host batches: double-buffered chunks
    hipStream_t copy_stream = nullptr; hipEvent_t ev_copy[2] = {nullptr, nullptr};
    // sfgpu_eq_add_batch_host: small host batches (the mapper threads hand over ~1000 reads at a time) are
    // appended to ONE pinned CSR and built kAccReads at a time -- a launch per 1000 reads would cap the
    // builder at a few million reads/s; the copy is one large pinned H2D transfer per flush
    uint32_t* acc_ids = nullptr; uint32_t* acc_off = nullptr;      // pinned
    uint64_t acc_n_ids = 0; uint32_t acc_n_reads = 0;
    std::atomic<int> acc_writers{0};                               // threads still copying into a reserved range
    // finish() products
    DevBuf<uint32_t> order; DevBuf<uint64_t> rowptr64;
    uint64_t nnz = 0, total_reads = 0;
    uint32_t sub_batch = 1u << 22;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    sfgpu_eq_stats stats{};
    // radix-partitioned path (eqclass_part.h)
    bool use_part = true; uint32_t part_sub_batch = 1u << 24;
    DevBuf<uint32_t> part_words, part_hist, part_cursor, part_long, def_lens, def_ids, def_off;
    DevBuf<uint64_t> def_w;                 // run lengths of the deferred labels (weights of their replay)
    uint64_t def_n = 0, def_words = 0;      // deferred labels saved out of the bins and not yet replayed (eq_deferred_save), their ids
    // small DEVICE batches are gathered here and built together (a sub-batch costs ~0.25 ms of launches and round trips whatever
    // its size: 100 M reads in 1 M-read batches took 34 ms instead of 5 ms)
    DevBuf<uint32_t> dacc_ids, dacc_off;
    uint32_t dacc_n_reads = 0; uint64_t dacc_n_ids = 0;
    // hot classes for k_part_route: kHotSlots hashes, kHotSlots (arena granule, slot) pairs, the count.  Two copies: the pipelined
    // path (eq_pipeline) rebuilds the table for sub-batch k + 1 while the route pass of sub-batch k still reads the other one
    DevBuf<unsigned long long> hot_bufs[2]; int hot_cur = 0;
    // ---- pipelined partition passes (eq_pipeline): route(k + 1) runs next to insert(k) on a second stream; two sets of bins
    struct PartSet {
        DevBuf<uint32_t> words, hist, longl, deferred;
        unsigned long long* d_ctr = nullptr;                  // CTR_N counters of the sub-batch in this set
        unsigned long long* h_ctr = nullptr;                  // pinned: [0, CTR_N) this set's counters, [CTR_N, 2 CTR_N) the builder's (arena cursor, classes)
        hipEvent_t ev_route = nullptr, ev_ins = nullptr;
        uint32_t first = 0, cnt = 0, n_blocks = 0; uint64_t n_words = 0, cap = 0;
        bool in_flight = false, fix = false;                  // launched and not yet collected / has deferred or spilled reads to replay
        uint64_t n_def = 0, n_long = 0;
    } pset[2];
    hipStream_t ins_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_hot = nullptr, ev_join = nullptr;
    DevBuf<uint32_t> grid_dev; uint32_t* grid_host = nullptr; uint64_t grid_cap = 0;     // offsets at the sub-batch grid (pinned copy)
    bool settled = false;       // the last partitioned sub-batch returned with the stream waited for and arena_used current
    // dev builds (SFGPU_EQ_PIPE): how the sub-batches of a large batch are queued (eq_pipeline).  0 (the product): one sub-batch, one host round
    // trip.  1: route(k + 1) on a second stream NEXT TO insert(k), the host one sub-batch behind (round 4: bit-exact, no faster -- the two
    // kernels take each other's block slots, profiles/r4_class_build_notes.md).  2: the same queueing on ONE stream -- no kernel overlap, no
    // idle device between sub-batches (round 6: the 8 round trips of cfg3 are 0.24 ms, and the form is 0.2 ms SLOWER: deciding the
    // sub-batch sizes one sub-batch late makes one launch pair more, profiles/r6_class_build_notes.md section 4)
    int pipe_mode = 0;
    uint64_t reads_seen = 0;                // reads added since start()
    uint64_t hot_cap = 0, hot_reads = 0;    // table size and reads_seen when the hot table was last rebuilt
    uint32_t hot_slots = 0;                 // ... and the number of slots it was laid out for
    uint32_t mix_mode = kMixSampled;        // how the bucket hash treats the tail of a long label (xxh64_device.h): sampled, or walked
                                            // once a cluster of long labels that agree in every sample has been seen (eq_generic)
    DevBuf<uint64_t> part_off, def_off64;
};

static int eq_alloc_table(sfgpu_eq* eq, uint64_t cap) {
    eq->table.p = nullptr; eq->table.cap = 0;
    int rc = eq->table.reserve(2 * cap, eq->stream, false);
    if (rc) return rc;
    eq->probe.p = nullptr; eq->probe.cap = 0;
    if ((rc = eq->probe.reserve(4 * cap, eq->stream, false))) return rc;
    hipLaunchKernelGGL(k_table_init, dim3(2048), dim3(kBlock), 0, eq->stream, eq->table.p, cap);
    SF_CHECK_LAUNCH();
    eq->cap = cap;
    return SFGPU_OK;
}

static int eq_grow(sfgpu_eq* eq, uint64_t new_cap) {
    SF_REQUIRE(new_cap <= (1ull << 31), SFGPU_ERR_RANGE, "equivalence-class table would exceed 2^31 slots");
    uint64_t* old = eq->table.p;
    uint32_t* old_probe = eq->probe.p;
    int rc = eq_alloc_table(eq, new_cap);
    if (rc) return rc;
    if (eq->n_classes) {
        hipLaunchKernelGGL(k_rehash, dim3(grid_for(eq->n_classes)), dim3(kBlock), 0, eq->stream, old, eq->table.p,
                           new_cap - 1, eq->n_classes, eq->cls_off.p, eq->cls_len.p, eq->arena.p, eq->cls_slot.p, eq->mix_mode, eq->probe.p);
        SF_CHECK_LAUNCH();
    }
    SF_HIP(hipStreamSynchronize(eq->stream));
    if (old) pool_free(old);
    if (old_probe) pool_free(old_probe);
    eq->stats.table_grows++;
    log_msg(0, "eq: table grown to %llu slots (%llu classes)", (unsigned long long)new_cap,
            (unsigned long long)eq->n_classes);
    return SFGPU_OK;
}

constexpr uint32_t kHostChunkReads = 1u << 24;   // reads per chunk of a large host batch (copy of chunk k + 1 overlaps the build of chunk k)
constexpr uint64_t kHostChunkIds = 1ull << 27;    // ... and ids (512 MB)
constexpr uint32_t kAccReads = 1u << 21;         // reads per accumulated host batch
#ifndef SFGPU_MAX_SUBBATCH_LOG2
#define SFGPU_MAX_SUBBATCH_LOG2 26
#endif
constexpr uint64_t kMaxSubBatch = 1ull << SFGPU_MAX_SUBBATCH_LOG2;   // reads per partitioned sub-batch (they grow x4 up to this)
constexpr uint64_t kAccIds = 1ull << 24;          // ids per accumulated host batch (64 MB pinned)

static uint64_t pow2_at_least(uint64_t x) { uint64_t p = 1; while (p < x) p <<= 1; return p; }

static int eq_reset(sfgpu_eq* eq) {
    eq->stats = sfgpu_eq_stats{};
    eq->finished = false; eq->n_classes = 0; eq->arena_used = 0; eq->nnz = 0; eq->total_reads = 0;
    eq->acc_n_ids = 0; eq->acc_n_reads = 0; eq->dacc_n_reads = 0; eq->dacc_n_ids = 0;
    eq->reads_seen = 0; eq->hot_cap = 0; eq->hot_reads = 0; eq->mix_mode = kMixSampled;
    for (auto& S : eq->pset) { S.in_flight = false; S.fix = false; }
    eq->def_n = 0; eq->def_words = 0;
    uint64_t want = pow2_at_least(2 * (eq->expected ? eq->expected : 1000000ull) + 2 * kSlack);
    if (eq->table.p && eq->cap == want) {
        hipLaunchKernelGGL(k_table_init, dim3(2048), dim3(kBlock), 0, eq->stream, eq->table.p, eq->cap);
        SF_CHECK_LAUNCH();
    } else {
        if (eq->table.p) { SF_HIP(hipStreamSynchronize(eq->stream)); pool_free(eq->table.p); if (eq->probe.p) pool_free(eq->probe.p); }
        int rc = eq_alloc_table(eq, want);
        if (rc) return rc;
    }
    SF_HIP(hipMemsetAsync(eq->d_ctr, 0, CTR_N * sizeof(unsigned long long), eq->stream));
    return SFGPU_OK;
}

extern "C" {

int sfgpu_xxh64_labels(const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t n_reads, uint64_t* d_hashes,
                       sfgpu_stream stream) {
    SF_REQUIRE(d_offsets && d_hashes, SFGPU_ERR_INVALID, "sfgpu_xxh64_labels: null pointer");
    if (n_reads == 0) return SFGPU_OK;
    hipLaunchKernelGGL(k_hash_labels, dim3(grid_for(n_reads)), dim3(kBlock), 0, as_stream(stream), d_ids, d_offsets,
                       n_reads, d_hashes);
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

int sfgpu_eq_create(sfgpu_eq** out, uint64_t expected_classes, sfgpu_stream stream) {
    SF_REQUIRE(out, SFGPU_ERR_INVALID, "sfgpu_eq_create: null out");
    sfgpu_eq* eq = new sfgpu_eq();
    eq->stream = as_stream(stream);
    eq->expected = expected_classes;
    if (const char* e = getenv("SFGPU_EQ_SUBBATCH")) { long v = atol(e); if (v >= 1024) { eq->sub_batch = (uint32_t)v; eq->part_sub_batch = (uint32_t)v; } }
    if (const char* e = SF_DEV_ENV("SFGPU_EQ_PARTITION")) eq->use_part = atoi(e) != 0;
    if (const char* e = SF_DEV_ENV("SFGPU_EQ_PIPE")) { const int v = atoi(e); eq->pipe_mode = (v >= 0 && v <= 2) ? v : 0; }
    hipError_t e1 = pool_malloc(&eq->d_ctr, CTR_N * sizeof(unsigned long long));
    hipError_t e2 = pinned_malloc(&eq->h_ctr, (CTR_N + 4) * sizeof(unsigned long long));      // (+ 4: scratch for small readbacks)
    if (e1 == hipSuccess) e1 = hipEventCreate(&eq->ev0);
    if (e1 == hipSuccess) e1 = hipEventCreate(&eq->ev1);
    if (e1 != hipSuccess || e2 != hipSuccess) {
        set_error("sfgpu_eq_create: allocation failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        delete eq; return SFGPU_ERR_HIP;
    }
    int rc = eq_reset(eq);
    if (rc) { sfgpu_eq_destroy(eq); return rc; }
    *out = eq;
    return SFGPU_OK;
}

int sfgpu_eq_destroy(sfgpu_eq* eq) {
    if (!eq) return SFGPU_OK;
    (void)hipStreamSynchronize(eq->stream);
    if (eq->d_ctr) pool_free(eq->d_ctr);
    if (eq->h_ctr) pinned_free(eq->h_ctr);
    if (eq->acc_ids) pinned_free(eq->acc_ids);
    if (eq->acc_off) pinned_free(eq->acc_off);
    if (eq->ev0) (void)hipEventDestroy(eq->ev0);
    if (eq->ev1) (void)hipEventDestroy(eq->ev1);
    if (eq->copy_stream) { (void)hipStreamSynchronize(eq->copy_stream); stream_release(eq->copy_stream); }
    for (int i = 0; i < 2; ++i) if (eq->ev_copy[i]) (void)hipEventDestroy(eq->ev_copy[i]);
    if (eq->ins_stream) { (void)hipStreamSynchronize(eq->ins_stream); stream_release(eq->ins_stream); }
    for (hipEvent_t ev : {eq->ev_fork, eq->ev_hot, eq->ev_join}) if (ev) (void)hipEventDestroy(ev);
    for (auto& S : eq->pset) {
        if (S.d_ctr) pool_free(S.d_ctr);
        if (S.h_ctr) pinned_free(S.h_ctr);
        if (S.ev_route) (void)hipEventDestroy(S.ev_route);
        if (S.ev_ins) (void)hipEventDestroy(S.ev_ins);
    }
    if (eq->grid_host) pinned_free(eq->grid_host);
    delete eq;
    return SFGPU_OK;
}

int sfgpu_eq_start(sfgpu_eq* eq) {
    SF_REQUIRE(eq, SFGPU_ERR_INVALID, "sfgpu_eq_start: null handle");
    std::lock_guard<std::mutex> lk(eq->mu);
    return eq_reset(eq);
}

// generic path: one lane per read probing the table in HBM (any batch size, weights, replays).
// Handles reads [first, first+todo) or, with `list`, the reads list[0..todo).
static int eq_generic(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t first, uint32_t todo,
                      const uint32_t* list, const uint64_t* d_weights) {
    hipStream_t st = eq->stream;
    int rc;
    bool flip = false;
    uint64_t prev_def = 0;      // reads the previous launch of this call deferred (0: this is the first launch)
    int stalls = 0;
    while (todo) {
        // new-class budget of this launch: keep load <= 1/2 with room for the guard's slack
        int64_t free_budget = (int64_t)(eq->cap / 2) - (int64_t)eq->n_classes - (int64_t)kSlack;
        uint64_t need = todo < kMinHeadroom ? todo : kMinHeadroom;
        if (free_budget < (int64_t)need) { if ((rc = eq_grow(eq, eq->cap * 2))) return rc; continue; }
        uint64_t limit_new = (uint64_t)free_budget < todo ? (uint64_t)free_budget : todo;
        if ((rc = eq->newlist.reserve(limit_new + kSlack, st, false))) return rc;
        DevBuf<uint32_t>& dout = flip ? eq->deferred_b : eq->deferred_a;
        if ((rc = dout.reserve(todo, st, false))) return rc;
        uint64_t cls_need = eq->n_classes + limit_new + kSlack;
        SF_REQUIRE(cls_need < kArenaBit, SFGPU_ERR_RANGE, "more than 2^31 equivalence classes");
        if ((rc = eq->cls_hash.reserve(cls_need, st, true, eq->n_classes))) return rc;
        if ((rc = eq->cls_off.reserve(cls_need, st, true, eq->n_classes))) return rc;
        if ((rc = eq->cls_len.reserve(cls_need, st, true, eq->n_classes))) return rc;
        if ((rc = eq->cls_slot.reserve(cls_need, st, true, eq->n_classes))) return rc;
        SF_HIP(hipMemsetAsync(eq->d_ctr, 0, 2 * sizeof(unsigned long long), st));
        SF_HIP(hipEventRecord(eq->ev0, st));
        hipLaunchKernelGGL(k_insert, dim3(grid_for(todo)), dim3(kBlock), 0, st, d_ids, d_offsets, first, todo, list,
                           eq->table.p, eq->cap - 1, eq->cls_off.p, eq->cls_len.p, eq->arena.p, eq->d_ctr,
                           eq->newlist.p, (unsigned long long)limit_new, dout.p, d_weights, eq->mix_mode);
        SF_CHECK_LAUNCH();
        SF_HIP(hipEventRecord(eq->ev1, st));
        SF_HIP(hipMemcpyAsync(eq->h_ctr, eq->d_ctr, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        SF_HIP(hipStreamSynchronize(st));
        { float ms = 0.f; if (hipEventElapsedTime(&ms, eq->ev0, eq->ev1) == hipSuccess) eq->stats.insert_ms += ms; }
        eq->stats.insert_launches++;
        uint64_t n_new = eq->h_ctr[CTR_NEW], n_def = eq->h_ctr[CTR_DEFER];
        if (n_new) {
            hipLaunchKernelGGL(k_commit, dim3(grid_for(n_new)), dim3(kBlock), 0, st, d_ids, d_offsets, eq->table.p,
                               eq->newlist.p, n_new, eq->n_classes, eq->cls_hash.p, eq->cls_off.p, eq->cls_len.p,
                               eq->cls_slot.p, eq->arena.p, eq->d_ctr, eq->probe.p);
            SF_CHECK_LAUNCH();
            eq->n_classes += n_new;
        }
        if (n_def) {
            eq->stats.deferred_reads += n_def;
            // A replay after growth that defers (nearly) as many reads as before, in a table that is far from full, is not
            // short of room: its labels share a home slot at every table size -- distinct long labels that agree in every field
            // the sampled bucket hash looks at (xxh64_device.h).  The builder then hashes whole labels for the rest of its
            // life (the table is rehashed in place); should even that not spread them, the loop gives up instead of doubling
            // the table until an allocation fails.
            const bool no_progress = prev_def != 0 && n_def * 2 > prev_def && eq->n_classes * 8 < eq->cap;
            if (no_progress && eq->mix_mode == kMixSampled) {
                eq->mix_mode = kMixFull; eq->hot_cap = 0;
                log_msg(0, "eq: %llu reads deferred again after growth at load %.3f: long labels share their sampled ids -- hashing whole labels from here on",
                        (unsigned long long)n_def, (double)eq->n_classes / (double)eq->cap);
                if ((rc = eq_grow(eq, eq->cap))) return rc;       // same size, new hash (synchronises: commit has finished)
                eq->stats.table_grows--;                          // (a rehash, not a growth)
                stalls = 0;
            } else {
                if (no_progress) ++stalls;
                SF_REQUIRE(stalls < 3, SFGPU_ERR_RANGE, "equivalence-class table: labels cannot be placed (one region keeps overflowing)");
                if ((rc = eq_grow(eq, eq->cap * 2))) return rc;   // synchronises: commit has finished
            }
            prev_def = n_def;
            list = dout.p; todo = (uint32_t)n_def; flip = !flip;
        } else {
            todo = 0;
        }
    }
    return SFGPU_OK;
}


// geometry of one partitioned sub-batch (direct form of pass 1): blocks of pass 1, reads per block, granules per bin.
// Bin capacity in 16-byte granules: a label of n ids takes ceil((n + 1) / 4) <= (n + 4) / 4 granules, so (ids + 4 reads) / 4 bounds
// the stream from above; a bin gets its share of that bound plus 25 % and a constant -- the share is a sum of ~mean / 1.6
// independent labels, so this is > 6 standard deviations for hashed labels.  A region far above its share (one label holding a
// large part of the reads) overflows into the generic kernel's list.  Rounded to whole 128-byte lines.
struct PartGeom { uint32_t n_blocks, tile; uint64_t cap, n_bins;
                  uint32_t tile_hi = 0, tile_lo = 0; };         // two tile sizes (tile_lo = 0: all blocks take `tile`)
static PartGeom part_geometry(uint32_t cnt, uint64_t n_words, uint32_t n_regions, uint32_t mb) {
    PartGeom g;
    if (mb > 1024u) mb = 1024u;
    g.n_blocks = (cnt + 2047u) / 2048u; if (g.n_blocks > mb) g.n_blocks = mb; if (g.n_blocks == 0) g.n_blocks = 1;     // (>= 2 steps per wavefront)
    g.tile = (uint32_t)((((uint64_t)cnt + g.n_blocks - 1) / g.n_blocks + 63) & ~63ull);     // whole wavefront steps
    g.n_bins = (uint64_t)n_regions * g.n_blocks;
    const uint64_t stream_gr = (n_words + 4ull * cnt) / 4 + 1;
    g.cap = (stream_gr + g.n_bins - 1) / g.n_bins;
    // two tile sizes for a full launch (see RouteArgs::half): the first half of the blocks takes (1000 + skew) / 1000 of the mean
    g.tile_hi = g.tile; g.tile_lo = 0;
    // (150: measured on cfg3 -- equal tiles: first half of the blocks done after 441 us, second after 510; 15 % skew: 486 / 491, the launch
    //  3.4 % shorter, the step 16.59 -> 16.44 ms; cfg2 and the sorted / clustered / long-label probes: no worse.  SFGPU_EQ_SKEW=0: equal tiles)
    static const uint32_t skew = []() { const char* e = SF_DEV_ENV("SFGPU_EQ_SKEW"); long v = e ? atol(e) : 150; return (uint32_t)(v > 0 && v <= 300 ? v : 0); }();
    if (skew && g.n_blocks == mb && (g.n_blocks & 1u) == 0u && g.n_blocks >= 64u && cnt >= (1u << 20)) {
        const uint64_t half = g.n_blocks / 2;
        const uint64_t hi = (((uint64_t)cnt * (1000u + skew) / 1000u + g.n_blocks - 1) / g.n_blocks + 63) & ~63ull;
        const uint64_t rest = (uint64_t)cnt > half * hi ? (uint64_t)cnt - half * hi : 0;
        const uint64_t lo = ((rest + half - 1) / half + 63) & ~63ull;
        if (lo >= 64) { g.tile_hi = (uint32_t)hi; g.tile_lo = (uint32_t)lo; g.cap = g.cap * (1000u + skew) / 1000u + 1; }
    }
    g.cap = g.cap + g.cap / 4 + 48;
    g.cap = (g.cap + 7) & ~7ull;
    return g;
}
static uint32_t part_max_blocks() {
    static const uint32_t v = []() { const char* e = SF_DEV_ENV("SFGPU_EQ_BLOCKS"); long x = e ? atol(e) : 512; return (uint32_t)(x >= 1 && x <= 1024 ? x : 512); }();
    return v;
}
static uint64_t part_load_div() {
    static const uint64_t v = []() { const char* e = SF_DEV_ENV("SFGPU_EQ_LOAD_DIV"); long x = e ? atol(e) : 2; return (uint64_t)(x >= 2 && x <= 8 ? x : 2); }();
    return v;
}

// Hot classes: a label that already holds more than 1/8 of a region's fair share of the reads (1.25x the share is what a
// region's bins hold) is counted in the route pass itself.  Real RNA-seq is skewed like that -- a highly expressed gene
// holds percents of the reads -- and without this its region overflows into the generic kernel read after read (measured
// on 50 M reads: 10 % on one label 76 ms, 50 % on 100 labels 22 ms, for a 2.4 ms build).  The table is rebuilt when the
// class table has grown (slots moved) and each time the reads seen have quadrupled; the first sub-batch of a builder is
// kept small (see eq_add_locked) so that the hot classes are known before the bulk of the reads arrives.
// `into`: which of the two buffers to (re)build when a rebuild is due; `reads`: reads the table holds when the kernel runs;
// `have_classes`: the table may hold classes.  *rebuilt = true when `into` was written (the caller then makes it current).
static int eq_hot_refresh(sfgpu_eq* eq, uint32_t n_regions, uint32_t hs, hipStream_t st, int into, uint64_t reads, bool have_classes, bool* rebuilt) {
    int rc;
    *rebuilt = false;
    for (auto& b : eq->hot_bufs)
        if (!b.p) {
            if ((rc = b.reserve(2ull * kHotSlots + 2, st, false))) return rc;
            SF_HIP(hipMemsetAsync(b.p, 0, (2ull * kHotSlots + 2) * 8, st));
            eq->hot_cap = 0; eq->hot_reads = 0;
        }
    // (the ring form's LDS leaves room for kRingHotSlots entries: the table is laid out for the form the sub-batch takes)
    unsigned long long* hot_h = eq->hot_bufs[into].p;
    uint2* hot_meta = reinterpret_cast<uint2*>(hot_h + hs);
    unsigned int* n_hot = reinterpret_cast<unsigned int*>(hot_h + 2ull * hs);
    if (have_classes && reads && (eq->hot_cap != eq->cap || eq->hot_slots != hs || reads >= 4 * eq->hot_reads)) {
        const unsigned long long thr = std::max<unsigned long long>(64ull, reads / (8ull * n_regions));
        SF_HIP(hipMemsetAsync(hot_h, 0, (2ull * kHotSlots + 2) * 8, st));
        hipLaunchKernelGGL(k_hot_select, dim3(grid_for(eq->cap)), dim3(kBlock), 0, st, eq->table.p, eq->cap, thr, eq->arena.p, hot_h, hot_meta, n_hot, hs, eq->mix_mode);
        SF_CHECK_LAUNCH();
        eq->hot_cap = eq->cap; eq->hot_reads = reads; eq->hot_slots = hs;
        *rebuilt = true;
    } else if (eq->hot_cap != eq->cap || eq->hot_slots != hs) {                 // (an empty table: nothing is hot)
        SF_HIP(hipMemsetAsync(hot_h, 0, (2ull * kHotSlots + 2) * 8, st));
        eq->hot_cap = eq->cap; eq->hot_reads = reads; eq->hot_slots = hs;
        *rebuilt = true;
    }
    return SFGPU_OK;
}

static int eq_generic(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t first, uint32_t todo,
                      const uint32_t* list, const uint64_t* d_weights);
// Labels whose home region was full stay in the bins as (granule index, length) pairs: before the bins are reused they are
// copied out -- APPENDED to a small CSR batch (def_ids / def_off / def_w) that eq_part_fixups inserts the generic way once the
// table may grow.  `deferred` = the n_new pairs to save.  Synchronises eq->stream (the size of the copy is read back).
static int eq_deferred_save(sfgpu_eq* eq, const uint32_t* bins_words, const uint32_t* deferred, uint64_t n_new) {
    if (!n_new) return SFGPU_OK;
    hipStream_t st = eq->stream;
    int rc;
    if ((rc = eq->def_lens.reserve(n_new + 1, st, false)) || (rc = eq->def_off64.reserve(n_new + 2, st, false))) return rc;
    hipLaunchKernelGGL(k_deferred_lens, dim3(grid_for(n_new + 1)), dim3(kBlock), 0, st, n_new, deferred, eq->def_lens.p);
    SF_CHECK_LAUNCH();
    if ((rc = exclusive_scan_u32(eq->def_lens.p, eq->def_off64.p, n_new, st))) return rc;
    uint64_t tot = 0;
    SF_HIP(hipMemcpyAsync(&tot, eq->def_off64.p + n_new, 8, hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    SF_REQUIRE(eq->def_words + tot < (1ull << 32), SFGPU_ERR_RANGE, "deferred labels exceed 2^32 ids");
    if ((rc = eq->def_ids.reserve(eq->def_words + tot + 1, st, true, eq->def_words)) ||
        (rc = eq->def_off.reserve(eq->def_n + n_new + 1, st, true, eq->def_n + 1)) ||
        (rc = eq->def_w.reserve(eq->def_n + n_new + 1, st, true, eq->def_n))) return rc;
    hipLaunchKernelGGL(k_deferred_copy, dim3(grid_for(n_new + 1)), dim3(kBlock), 0, st, n_new, deferred,
                       reinterpret_cast<const uint4*>(bins_words), eq->def_off64.p, eq->def_ids.p + eq->def_words, eq->def_off.p + eq->def_n,
                       eq->def_w.p + eq->def_n, eq->def_words);
    SF_CHECK_LAUNCH();
    eq->def_n += n_new; eq->def_words += tot;
    eq->stats.deferred_reads += n_new;
    return SFGPU_OK;
}

// what partitioned sub-batches leave for the generic kernel: the deferred labels saved by eq_deferred_save (the table is grown,
// they are inserted the generic way) and reads that never entered the stream (bin overflow, over-long labels: `long_list`).
// The table must be quiescent (no partition pass in flight); synchronises eq->stream.
static int eq_part_fixups(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets, const uint32_t* long_list, uint64_t n_long) {
    hipStream_t st = eq->stream;
    int rc;
    eq->stats.spilled_reads += n_long;
    const uint64_t n_def = eq->def_n, tot = eq->def_words;
    if (n_def) {
        eq->def_n = 0; eq->def_words = 0;
        // worst case every deferred label opens a class
        const uint64_t need = eq->arena_used + tot + 8 * n_def + 4;
        SF_REQUIRE((need >> 2) < kArenaBit, SFGPU_ERR_RANGE, "sfgpu_eq_add_batch: label arena would exceed 2^33 words");
        if ((rc = eq->arena.reserve(need, st, true, eq->arena_used))) return rc;
        if ((rc = eq_grow(eq, eq->cap * 2))) return rc;
        if ((rc = eq_generic(eq, eq->def_ids.p, eq->def_off.p, 0, (uint32_t)n_def, nullptr, eq->def_w.p))) return rc;
    }
    if (n_long) {        // labels too long for an LDS tile, reads that did not fit their bin
        if ((rc = eq_generic(eq, d_ids, d_offsets, 0, (uint32_t)n_long, long_list, nullptr))) return rc;
    }
    if (n_def || n_long) {      // the generic kernel's commits moved the arena cursor
        SF_HIP(hipMemcpyAsync(eq->h_ctr + CTR_ARENA, eq->d_ctr + CTR_ARENA, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        SF_HIP(hipStreamSynchronize(st));
        eq->arena_used = eq->h_ctr[CTR_ARENA];
    }
    return SFGPU_OK;
}

// start of a partitioned sub-batch: zero its counters and fetch the offsets the host will want for the NEXT sub-batch (its
// possible ends), so that they come back with this sub-batch's counters instead of in a round trip of their own
__global__ void k_sub_batch_begin(unsigned long long* ctr, const uint32_t* __restrict__ offsets, uint64_t p0, uint64_t p1, uint64_t p2, unsigned long long n_classes) {
    if (threadIdx.x == 0) { ctr[CTR_NEW] = 0; ctr[CTR_DEFER] = 0; ctr[CTR_TMP] = 0; ctr[CTR_HOT] = 0; ctr[CTR_GCLS] = n_classes; }
    if (threadIdx.x == 1) ctr[CTR_PEEK + 0] = offsets[p0];
    if (threadIdx.x == 2) ctr[CTR_PEEK + 1] = offsets[p1];
    if (threadIdx.x == 3) ctr[CTR_PEEK + 2] = offsets[p2];
}

static int device_cus() {
    static const int n = []() {
        int dev = 0, n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        return n_cu > 0 ? n_cu : 256;
    }();
    return n;
}

// radix-partitioned path for one sub-batch (see eqclass_part.h), SERIAL form: route, insert, one host round trip.  Small batches,
// batches with a caller-chosen sub-batch size and the ring form of pass 1 come here; large batches take eq_pipeline below.
// n_words = ids in the sub-batch.  peek[3] = positions in d_offsets whose values are returned in eq->h_ctr[CTR_PEEK..]
static int eq_partitioned(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t first, uint32_t cnt,
                          uint64_t n_words, const uint64_t* peek) {
    hipStream_t st = eq->stream;
    int rc;
    eq->settled = false;
    // The table doubles when the classes seen so far would fill more than half of it (SFGPU_EQ_LOAD_DIV: 1/div).  A
    // region's LDS image overflows at 3/4 (kRegionLimit) -- 22 standard deviations above a half-full region of 4096
    // slots -- and an overflowing region only defers its reads to the generic kernel.  Half rather than a quarter
    // halves the number of regions: pass 2 is one block per region, two blocks per CU, and fewer, longer-lived
    // blocks amortise the region load / write-back and the per-block ramp (cfg3, 1.6 M classes: 2048 -> 1024
    // regions, class build 22.8 -> 21.5 ms; cfg2 unchanged).
    const uint64_t load_div = part_load_div();
    while (eq->n_classes + kMinHeadroom > eq->cap / load_div) {
        if ((rc = eq_grow(eq, eq->cap * 2))) return rc;
    }
    if ((eq->cap >> kRegionBits) > (uint64_t)kMaxRegions) {     // grew past what the partition passes handle
        for (uint32_t f2 = first; f2 < first + cnt; f2 += eq->sub_batch) {
            const uint32_t c2 = (first + cnt - f2 < eq->sub_batch) ? (first + cnt - f2) : eq->sub_batch;
            if ((rc = eq_generic(eq, d_ids, d_offsets, f2, c2, nullptr, nullptr))) return rc;
        }
        return SFGPU_OK;
    }
    const uint32_t n_regions = (uint32_t)(eq->cap >> kRegionBits);
    uint64_t cls_need = eq->n_classes + cnt + 1;
    SF_REQUIRE(cls_need < kArenaBit, SFGPU_ERR_RANGE, "more than 2^31 equivalence classes");
    if ((rc = eq->cls_hash.reserve(cls_need, st, true, eq->n_classes))) return rc;
    if ((rc = eq->cls_off.reserve(cls_need, st, true, eq->n_classes))) return rc;
    if ((rc = eq->cls_len.reserve(cls_need, st, true, eq->n_classes))) return rc;
    if ((rc = eq->cls_slot.reserve(cls_need, st, true, eq->n_classes))) return rc;
    // A table of more than kGroupRegions regions (16 M slots: more than ~8 M classes) is built in GROUPS of kGroupRegions regions:
    // one route + insert pass over the sub-batch per group, each skipping the reads whose label lives in another group.  The
    // cursors of a group fit the route pass's LDS next to a second block (2 x 4096 x 4 B), its bins stay at 2 M lines; the price
    // is reading and hashing the sub-batch once per group (round 4: before, such tables fell back to the generic kernel at
    // ~6 G reads/s).
    const uint32_t n_groups = (n_regions + kGroupRegions - 1) / kGroupRegions;
    const uint32_t grp_n = n_regions / n_groups;                  // (both powers of two)
    // every block of pass 1 owns one contiguous tile of reads and one bin per region.  The RING form of the pass (bins written
    // through LDS rings, eqclass_part.h) wants one block per CU and fits up to 1024 regions (a table of <= 4 M slots).  It is
    // OFF by default (SFGPU_EQ_RING=1 selects it): it writes whole 64-byte units -- no partial lines -- but one block per CU is
    // 4 wavefronts per SIMD and the pass is bound by instruction issue there: cfg3 9.4 ms per build against 9.7 under rocprofv3,
    // no difference in the bench step, and skewed / sorted streams and cfg2 are 10-80 % slower (profiles/r3_class_build_notes.md)
    const int ring_mode = []() { const char* e = SF_DEV_ENV("SFGPU_EQ_RING"); return e ? atoi(e) : 0; }();      // (read per sub-batch: tests switch it)
    static const uint32_t ring_blocks = []() { const char* e = SF_DEV_ENV("SFGPU_EQ_RING_BLOCKS"); long v = e ? atol(e) : 0; return (uint32_t)(v >= 1 && v <= 1024 ? v : 0); }();
    bool ring = ring_mode != 0 && n_regions >= 2 && n_regions <= kRingMaxRegions;
    // (a bin's share of the stream is computed over ALL regions -- a group's launch fills the bins of its regions only)
    PartGeom gm = part_geometry(cnt, n_words, n_regions, ring ? (ring_blocks ? ring_blocks : (uint32_t)device_cus()) : part_max_blocks());
    if (ring && gm.cap > kRingMaxCap) {       // (16-bit cursors: few regions and a huge sub-batch take the direct form)
        ring = false;
        gm = part_geometry(cnt, n_words, n_regions, part_max_blocks());
    }
    // the QUAD form of pass 1 (single-granule labels written four at a time through LDS mailboxes, eqclass_part.h): tables of up
    // to 1024 regions, bins the 16-bit cursors can address.  OFF by default (SFGPU_EQ_QUAD=1 selects it): it does write whole
    // 64-byte units, but its per-region ticket couples the wavefronts of a block (a label waits for every label that reserved
    // before it in its region, wherever that wavefront is) -- route 8.75 ms per build against the direct form's 8.0 on one box
    // (profiles/r4_class_build_notes.md).
    const int quad_mode = []() { const char* e = SF_DEV_ENV("SFGPU_EQ_QUAD"); return e ? atoi(e) : 0; }();         // (read per sub-batch: tests switch it)
    const bool quad = !ring && quad_mode != 0 && n_groups == 1 && n_regions >= 2 && n_regions <= kRingMaxRegions && gm.cap <= kRingMaxCap;
    // the SHARED form of pass 1 (round 6, eqclass_part.h): the blocks of an XCD share one bin per region and reserve a step's granules of a
    // region with one atomic.  SFGPU_EQ_SHARED=1 selects it (dev builds).
    const int shared_mode = []() { const char* e = SF_DEV_ENV("SFGPU_EQ_SHARED"); return e ? atoi(e) : 0; }();
    const bool shared = !ring && !quad && shared_mode != 0 && n_groups == 1 && n_regions >= 2 && n_regions <= (uint32_t)kPartBlock && gm.n_blocks >= 64u;
    const uint32_t n_blocks = gm.n_blocks, tile = gm.tile;
    // (shared: a bin is an XCD's share of the region -- n_blocks / 8 block bins and their slack in one; + 1/8 for XCDs that got more blocks)
    const uint32_t n_pass2_bins = shared ? kSharedBins : n_blocks;
    const uint64_t cap = shared ? (((gm.cap * n_blocks / kSharedBins) * 9 / 8 + 7) & ~7ull) : gm.cap, n_bins = (uint64_t)grp_n * n_pass2_bins;
    // positions inside the bins are 31-bit granule indices (bit 31 of a slot's rep marks arena entries)
    SF_REQUIRE(n_bins * cap < (1ull << 31), SFGPU_ERR_RANGE, "partition buffer would exceed 2^31 granules");
    if ((rc = eq->part_words.reserve(n_bins * cap * 4 + 8, st, false))) return rc;
    if ((rc = eq->part_hist.reserve(4 * n_bins + 1, st, false))) return rc;      // fill of every bin (front, back) + the ring form's cut marks (shared form: the XCDs' cursors)
    if ((rc = eq->part_long.reserve(cnt, st, false))) return rc;
    if ((rc = eq->deferred_a.reserve(2ull * cnt, st, false))) return rc;
    hipLaunchKernelGGL(k_sub_batch_begin, dim3(1), dim3(64), 0, st, eq->d_ctr, d_offsets, peek[0], peek[1], peek[2], (unsigned long long)eq->n_classes);   // CTR_NEW, CTR_DEFER, long-label counter
    SF_CHECK_LAUNCH();
    SF_HIP(hipEventRecord(eq->ev0, st));
    uint4* bins = reinterpret_cast<uint4*>(eq->part_words.p);
    const uint32_t hs = (ring || quad) ? kRingHotSlots : kHotSlots;
    bool rebuilt = false;
    if ((rc = eq_hot_refresh(eq, n_regions, hs, st, eq->hot_cur, eq->reads_seen, eq->n_classes != 0, &rebuilt))) return rc;
    unsigned long long* hot_h = eq->hot_bufs[eq->hot_cur].p;
    uint32_t* fill_f = eq->part_hist.p, *fill_b = fill_f + n_bins, *cutmarks = fill_b + n_bins;
    uint64_t saved_def = 0;                        // deferred pairs already copied out of the bins (groups)
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t grp_lo = g * grp_n;
        RouteArgs ra{d_ids, d_offsets + first, first, cnt, tile, n_regions - 1u, (uint32_t)cap, bins, fill_f, fill_b, cutmarks,
                     eq->d_ctr + 3, eq->part_long.p, hot_h, eq->arena.p, eq->table.p, eq->d_ctr + CTR_HOT, eq->mix_mode, grp_lo, grp_n, 0u, 0u};
        if (!ring && !quad && gm.tile_lo) { ra.tile = gm.tile_hi; ra.tile_lo = gm.tile_lo; ra.half = n_blocks / 2; }
#ifdef SFGPU_VARIANTS
        if (shared) {
            ra.gcur = cutmarks; ra.gcut = cutmarks + n_bins;
            hipLaunchKernelGGL(k_shared_begin, dim3(grid_for(n_bins)), dim3(kBlock), 0, st, ra.gcur, ra.gcut, (uint32_t)n_bins);
            const size_t route_lds = (size_t)2 * grp_n * 4 + (size_t)kPartWaves * (kStageWords / 4 + 4) * 16 + (size_t)kHotSlots * 12;
            hipLaunchKernelGGL(k_part_route<kFormShared>, dim3(n_blocks), dim3(kPartBlock), route_lds, st, ra);
            hipLaunchKernelGGL(k_shared_fill, dim3(grid_for(n_bins)), dim3(kBlock), 0, st, ra.gcur, ra.gcut, grp_n, fill_f, fill_b);
        } else if (ring) {
            const size_t route_lds = (size_t)n_regions * 128 + (size_t)n_regions * 8 + (size_t)kPartWaves * (kRingStageWords / 4 + 4) * 16 + (size_t)kRingHotSlots * 8 +
                                     (size_t)kPartWaves * kRingFlushList * 4;
            static const bool attr_ok = []() {
                return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_route<kFormRing>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
            }();
            (void)attr_ok;
            hipLaunchKernelGGL(k_part_route<kFormRing>, dim3(n_blocks), dim3(kPartBlock), route_lds, st, ra);
        } else if (quad) {
            const size_t route_lds = (size_t)n_regions * 48 + (size_t)n_regions * 8 + (size_t)kPartWaves * (kRingStageWords / 4 + 4) * 16 + (size_t)kRingHotSlots * 8;
            static const bool attr_ok = []() {
                return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_route<kFormQuad>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
            }();
            (void)attr_ok;
            hipLaunchKernelGGL(k_part_route<kFormQuad>, dim3(n_blocks), dim3(kPartBlock), route_lds, st, ra);
        } else
#endif
        {
            const size_t route_lds = (size_t)2 * grp_n * 4 + (size_t)kPartWaves * (kStageWords / 4 + 4) * 16 + (size_t)kHotSlots * 12;
            hipLaunchKernelGGL(k_part_route<kFormDirect>, dim3(n_blocks), dim3(kPartBlock), route_lds, st, ra);
        }
        SF_CHECK_LAUNCH();
        PartArgs pa{eq->table.p, bins, fill_f, fill_b, n_pass2_bins, (uint32_t)cap, eq->cls_hash.p, eq->cls_off.p, eq->cls_len.p, eq->cls_slot.p,
                    eq->arena.p, eq->d_ctr, eq->d_ctr + CTR_ARENA, eq->d_ctr + CTR_GCLS, eq->deferred_a.p, eq->mix_mode, grp_lo, 0u, eq->probe.p, shared ? 1u : 0u};
        static const bool route_only = SF_DEV_ENV("SFGPU_X_ROUTE_ONLY") != nullptr;     // (dev: time pass 1 alone -- its experiment variants leave no valid bins)
        if (!route_only) hipLaunchKernelGGL(k_part_insert, dim3(grp_n), dim3(kPartBlock), 0, st, pa);
        SF_CHECK_LAUNCH();
        if (g + 1 < n_groups) {
            // the next group reuses the bins: labels this group deferred (they are addressed by their position in the bins) leave now
            SF_HIP(hipMemcpyAsync(eq->h_ctr + CTR_DEFER, eq->d_ctr + CTR_DEFER, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            SF_HIP(hipStreamSynchronize(st));
            const uint64_t n_def_now = eq->h_ctr[CTR_DEFER];
            if (n_def_now > saved_def) {
                if ((rc = eq_deferred_save(eq, eq->part_words.p, eq->deferred_a.p + 2 * saved_def, n_def_now - saved_def))) return rc;
                saved_def = n_def_now;
            }
        }
    }
    SF_HIP(hipEventRecord(eq->ev1, st));
    SF_HIP(hipMemcpyAsync(eq->h_ctr, eq->d_ctr, CTR_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    { float ms = 0.f; if (hipEventElapsedTime(&ms, eq->ev0, eq->ev1) == hipSuccess) eq->stats.insert_ms += ms; }
    eq->stats.insert_launches++;
    eq->n_classes += eq->h_ctr[CTR_NEW];
    eq->arena_used = eq->h_ctr[CTR_ARENA];
    const uint64_t n_def = eq->h_ctr[CTR_DEFER], n_long = eq->h_ctr[3];
    eq->stats.hot_reads += eq->h_ctr[CTR_HOT];
    if (n_def > saved_def && (rc = eq_deferred_save(eq, eq->part_words.p, eq->deferred_a.p + 2 * saved_def, n_def - saved_def))) return rc;
    if ((rc = eq_part_fixups(eq, d_ids, d_offsets, eq->part_long.p, n_long))) return rc;
    eq->settled = true;                  // the stream has been waited for and arena_used is the device's cursor
    return SFGPU_OK;
}

constexpr uint32_t kScoutReads = 1u << 19;      // the first sub-batch of a builder: enough reads to see which classes are hot

// ---- the PIPELINED form of the partitioned path (round 4) ---------------------------------------------------------------
// The serial form runs route(k), insert(k), a host round trip, route(k + 1), ...: the two passes never overlap although they lean
// on different parts of a CU (route: scattered 16-byte stores, the memory path; insert: LDS probes and instruction issue), and
// the device idles through every round trip.  Here the sub-batches of ONE large batch are laid out ahead of the device:
//   * the offsets at every possible sub-batch boundary (a grid of kScoutReads reads) come back in one round trip per batch;
//   * two sets of bins / counters: route(k + 1) fills one set on the builder's stream while insert(k) drains the other on a
//     second stream (events order route(k) -> insert(k); the host has collected set k - 2 before it reuses it);
//   * insert numbers its new classes from a device-side counter (PartArgs::gcls) and adds its counts with atomics (the route
//     pass of the next sub-batch may be adding what it counted for hot classes), so neither needs what the host knows;
//   * the host reads sub-batch k - 1's counters (pinned copy behind insert(k - 1)) AFTER enqueueing sub-batch k: it is one
//     sub-batch behind the device, which always has the next route + insert queued;
//   * everything that needs a quiescent table -- growth, the generic kernel for deferred / spilled reads, a reallocation of the
//     arena or the class arrays -- DRAINS the pipeline first (both sets collected, both streams idle); the hot-class table is
//     rebuilt on the insert stream into its second copy, one sub-batch behind, except around the scout (drained: the bulk
//     must not start before the hot classes are known).
// Returns in *consumed how many reads it took; the caller's serial loop continues from there (0: nothing suited the pipeline).
__global__ void k_gather_offsets(const uint32_t* __restrict__ off, uint32_t n_reads, uint32_t g, uint32_t n_out, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    uint64_t pos = (uint64_t)i * g;
    if (pos > n_reads) pos = n_reads;                 // (the last entry is the batch's end)
    out[i] = off[pos];
}
__global__ void k_set_u64(unsigned long long* p, unsigned long long v) { *p = v; }
__global__ void k_zero_ctr(unsigned long long* ctr) { if (threadIdx.x < (unsigned)CTR_N) ctr[threadIdx.x] = 0ull; }

#ifdef SFGPU_VARIANTS      // (the pipelined forms of the partition passes: built, bit-exact, no faster -- profiles/r4_class_build_notes.md section 1, r6 section 4)
static int eq_pipeline(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t n_reads, uint32_t* consumed) {
    *consumed = 0;
    hipStream_t sr = eq->stream;
    int rc;
    if ((eq->cap >> kRegionBits) > (uint64_t)kGroupRegions) return SFGPU_OK;       // (tables built in groups take the serial form)
    if (!eq->ins_stream) {
        SF_HIP(stream_acquire(&eq->ins_stream));
        SF_HIP(hipEventCreateWithFlags(&eq->ev_fork, hipEventDisableTiming));
        SF_HIP(hipEventCreateWithFlags(&eq->ev_hot, hipEventDisableTiming));
        SF_HIP(hipEventCreateWithFlags(&eq->ev_join, hipEventDisableTiming));
        for (auto& S : eq->pset) {
            SF_HIP(pool_malloc(&S.d_ctr, CTR_N * sizeof(unsigned long long)));
            SF_HIP(pinned_malloc(&S.h_ctr, 2 * CTR_N * sizeof(unsigned long long)));
            SF_HIP(hipEventCreateWithFlags(&S.ev_route, hipEventDisableTiming));
            SF_HIP(hipEventCreateWithFlags(&S.ev_ins, hipEventDisableTiming));
        }
    }
    hipStream_t si = eq->pipe_mode == 2 ? sr : eq->ins_stream;       // (mode 2: everything in the builder's stream, in the order it is queued here)
    // ---- offsets at the grid of possible sub-batch boundaries: one round trip for the whole batch
    const uint32_t g = kScoutReads;
    const uint32_t n_grid = n_reads / g + 2;
    if ((rc = eq->grid_dev.reserve(n_grid, sr, false))) return rc;
    if (eq->grid_cap < n_grid) {
        if (eq->grid_host) pinned_free(eq->grid_host);
        eq->grid_host = nullptr; eq->grid_cap = 0;
        uint64_t c = 1024; while (c < n_grid) c *= 2;
        SF_HIP(pinned_malloc(&eq->grid_host, c * 4));
        eq->grid_cap = c;
    }
    hipLaunchKernelGGL(k_gather_offsets, dim3(grid_for(n_grid)), dim3(kBlock), 0, sr, d_offsets, n_reads, g, n_grid, eq->grid_dev.p);
    SF_CHECK_LAUNCH();
    SF_HIP(hipMemcpyAsync(eq->grid_host, eq->grid_dev.p, (size_t)n_grid * 4, hipMemcpyDeviceToHost, sr));
    SF_HIP(hipStreamSynchronize(sr));
    const uint32_t* grid = eq->grid_host;
    auto off_at = [&](uint64_t pos) -> uint32_t { return pos >= n_reads ? grid[n_grid - 1] : grid[pos / g]; };    // pos: a multiple of g, or the end
    const uint64_t batch_words = (uint64_t)(uint32_t)(off_at(n_reads) - off_at(0));                           // (modulo 2^32 like the offsets themselves)

    const uint64_t load_div = part_load_div();
    int n_fly = 0;                                   // sets in flight (launched, not collected): 0, 1 or 2
    int k = 0;                                       // sub-batches launched so far: sub-batch k uses set k & 1
    uint64_t fly_reads = 0, fly_words = 0;           // what the sets in flight hold
    double rate = 1.0;                               // new classes per read of the sub-batch collected last
    bool timing = false;
    int hot_next = -1;                               // copy of the hot table that becomes current with the next sub-batch (-1: none pending)
    auto open_timing = [&]() -> int { if (!timing) { SF_HIP(hipEventRecord(eq->ev0, sr)); timing = true; } return SFGPU_OK; };

    // collect the oldest set in flight: wait for its insert pass, take its counters
    auto collect = [&](sfgpu_eq::PartSet& S) -> int {
        SF_HIP(hipEventSynchronize(S.ev_ins));
        const unsigned long long* h = S.h_ctr;
        const uint64_t classes_before = eq->n_classes;
        eq->n_classes = h[CTR_N + CTR_GCLS];
        eq->arena_used = h[CTR_N + CTR_ARENA];
        S.n_def = h[CTR_DEFER]; S.n_long = h[CTR_TMP];
        S.fix = S.n_def != 0 || S.n_long != 0;
        eq->stats.hot_reads += h[CTR_HOT];
        eq->stats.insert_launches++;
        rate = (double)(eq->n_classes - classes_before + 1) / (double)S.cnt;
        fly_reads -= S.cnt; fly_words -= S.n_words;
        eq->reads_seen += S.cnt;
        S.in_flight = false; --n_fly;
        return SFGPU_OK;
    };
    // drain: both sets collected, both streams idle, deferred / spilled reads replayed, the device's class counter in step
    auto drain = [&]() -> int {
        int r;
        for (int j = 0; j < 2; ++j) {                // oldest first
            sfgpu_eq::PartSet& S = eq->pset[(k + j) & 1];
            if (S.in_flight && (r = collect(S))) return r;
        }
        SF_HIP(hipStreamSynchronize(si));
        if (timing) {
            SF_HIP(hipEventRecord(eq->ev1, sr));
            SF_HIP(hipStreamSynchronize(sr));
            float ms = 0.f; if (hipEventElapsedTime(&ms, eq->ev0, eq->ev1) == hipSuccess) eq->stats.insert_ms += ms;
            timing = false;
        } else SF_HIP(hipStreamSynchronize(sr));
        // (the deferred labels of BOTH sets first, then the fix-ups: eq_part_fixups doubles the table when it has deferred labels to
        //  replay -- set by set it doubled twice per drain and began its stuck-cluster detection anew)
        bool fixed = false;
        for (int j = 0; j < 2; ++j) {
            sfgpu_eq::PartSet& S = eq->pset[(k + j) & 1];
            if (S.fix && (r = eq_deferred_save(eq, S.words.p, S.deferred.p, S.n_def))) return r;
        }
        for (int j = 0; j < 2; ++j) {
            sfgpu_eq::PartSet& S = eq->pset[(k + j) & 1];
            if (!S.fix) continue;
            S.fix = false; fixed = true;
            if ((r = eq_part_fixups(eq, d_ids, d_offsets, S.longl.p, S.n_long))) return r;
        }
        if (fixed) {                                  // the generic kernel committed classes the device-side counter has not seen
            hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, sr, eq->d_ctr + CTR_GCLS, (unsigned long long)eq->n_classes);
            SF_CHECK_LAUNCH();
        }
        eq->stats.pipeline_drains++;
        return SFGPU_OK;
    };

    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, sr, eq->d_ctr + CTR_GCLS, (unsigned long long)eq->n_classes);
    SF_CHECK_LAUNCH();
    SF_HIP(hipEventRecord(eq->ev_fork, sr));
    SF_HIP(hipStreamWaitEvent(si, eq->ev_fork, 0));
    // the arena and the class arrays hold what the sub-batches in flight can add in the worst case (every read a new class); they
    // are sized once, for two sub-batches of the largest size, so that no reallocation (= drain) falls into the steady state
    {
        const uint64_t span = std::min<uint64_t>(n_reads, 2 * kMaxSubBatch);
        const uint64_t words_span = std::min<uint64_t>(batch_words, (uint64_t)((double)batch_words / (double)n_reads * 1.25 * (double)span) + 1024);
        const uint64_t need = eq->arena_used + words_span + 8 * span + 4;
        SF_REQUIRE((need >> 2) < kArenaBit, SFGPU_ERR_RANGE, "sfgpu_eq_add_batch: label arena would exceed 2^33 words");
        if ((rc = eq->arena.reserve(need, sr, true, eq->arena_used))) return rc;
        const uint64_t cls_need = eq->n_classes + span + 1;
        SF_REQUIRE(cls_need < kArenaBit, SFGPU_ERR_RANGE, "more than 2^31 equivalence classes");
        if ((rc = eq->cls_hash.reserve(cls_need, sr, true, eq->n_classes)) || (rc = eq->cls_off.reserve(cls_need, sr, true, eq->n_classes)) ||
            (rc = eq->cls_len.reserve(cls_need, sr, true, eq->n_classes)) || (rc = eq->cls_slot.reserve(cls_need, sr, true, eq->n_classes))) return rc;
    }

    uint32_t first = 0;
    uint32_t step = eq->part_sub_batch;
    const uint32_t usual_step = step;
    bool scout = eq->reads_seen == 0 && n_reads > (1u << 22) && step > kScoutReads;
    bool bail = false;
    while (first < n_reads && !bail) {
        uint32_t cnt = scout ? kScoutReads : ((n_reads - first < step) ? (n_reads - first) : step);
        uint64_t n_words = (uint32_t)(off_at((uint64_t)first + cnt) - off_at(first));
        while (n_words >= (1ull << 31) && cnt > (1u << 20)) {       // too many ids for 31-bit positions in the bins: halve
            cnt = (cnt / 2 + g - 1) / g * g; step = cnt;
            n_words = (uint32_t)(off_at((uint64_t)first + cnt) - off_at(first));
        }
        if (n_words >= (1ull << 31) || n_words < 64) { bail = true; break; }            // (the serial loop knows what to do with those)
        // ---- table growth, from the classes the host knows of (one sub-batch behind; a region that overflows meanwhile only
        //      defers its labels); capacity of the arena and the class arrays for the worst case of what is in flight + this
        if (eq->n_classes + kMinHeadroom > eq->cap / load_div) {
            if ((rc = drain())) return rc;
            while (eq->n_classes + kMinHeadroom > eq->cap / load_div) if ((rc = eq_grow(eq, eq->cap * 2))) return rc;
        }
        {
            uint64_t need = eq->arena_used + fly_words + 8 * fly_reads + n_words + 8ull * cnt + 4;
            uint64_t cls_need = eq->n_classes + fly_reads + cnt + 1;
            if (need > eq->arena.cap || cls_need > eq->cls_hash.cap) {
                if ((rc = drain())) return rc;
                need = eq->arena_used + n_words + 8ull * cnt + 4; cls_need = eq->n_classes + cnt + 1;
                SF_REQUIRE((need >> 2) < kArenaBit, SFGPU_ERR_RANGE, "sfgpu_eq_add_batch: label arena would exceed 2^33 words");
                SF_REQUIRE(cls_need < kArenaBit, SFGPU_ERR_RANGE, "more than 2^31 equivalence classes");
                if ((rc = eq->arena.reserve(need, sr, true, eq->arena_used))) return rc;
                if ((rc = eq->cls_hash.reserve(cls_need, sr, true, eq->n_classes)) || (rc = eq->cls_off.reserve(cls_need, sr, true, eq->n_classes)) ||
                    (rc = eq->cls_len.reserve(cls_need, sr, true, eq->n_classes)) || (rc = eq->cls_slot.reserve(cls_need, sr, true, eq->n_classes))) return rc;
            }
        }
        if ((eq->cap >> kRegionBits) > (uint64_t)kGroupRegions) { bail = true; break; }
        const uint32_t n_regions = (uint32_t)(eq->cap >> kRegionBits);
        const PartGeom gm = part_geometry(cnt, n_words, n_regions, part_max_blocks());
        if (gm.n_bins * gm.cap >= (1ull << 31)) { bail = true; break; }
        sfgpu_eq::PartSet& S = eq->pset[k & 1];           // free: the host collected its previous sub-batch (k - 2) an iteration ago
        if (S.in_flight && (rc = collect(S))) return rc;   // (cannot be: kept for safety)
        if (S.fix) { if ((rc = drain())) return rc; }      // its bins still hold deferred labels: replay them before they are overwritten
        // (reserve() may wait for the route stream when a buffer grows: first step only)
        if ((rc = S.words.reserve(gm.n_bins * gm.cap * 4 + 8, sr, false)) || (rc = S.hist.reserve(4 * gm.n_bins + 1, sr, false)) ||
            (rc = S.longl.reserve(cnt, sr, false)) || (rc = S.deferred.reserve(2ull * cnt, sr, false))) return rc;
        // ---- hot classes.  (A) exactly, for THIS sub-batch, drained: the first table, a grown table, and the scout's boundary (the
        //      bulk must not start before the hot classes are known).  (B) otherwise a rebuild that is due is made for the NEXT
        //      sub-batch, on the insert stream AHEAD of insert(k) -- i.e. right behind insert(k - 1), next to route(k) -- into the
        //      second copy (route(k) and perhaps route(k - 1) are reading the first); route(k + 1) waits for its event, which by
        //      then fired long ago.  So in the steady state route(k + 1) knows the table as of insert(k - 1): one sub-batch staler
        //      than the serial form, which only means that a class that has just become hot spills for one more sub-batch.
        const uint64_t reads_tab = eq->reads_seen + fly_reads;             // reads the table holds once the queued insert passes are done
        {
            const bool have = eq->n_classes != 0 || k > 0;
            const bool relaid = !eq->hot_bufs[0].p || eq->hot_cap != eq->cap || eq->hot_slots != kHotSlots;
            const bool small_due = have && reads_tab && reads_tab < (1ull << 22) && reads_tab >= 4 * eq->hot_reads;
            bool rebuilt = false;
            if (relaid || small_due) {
                if (n_fly && (rc = drain())) return rc;
                hot_next = -1;
                if ((rc = eq_hot_refresh(eq, n_regions, kHotSlots, sr, eq->hot_cur, eq->reads_seen, eq->n_classes != 0, &rebuilt))) return rc;
            } else if (hot_next >= 0) {
                SF_HIP(hipStreamWaitEvent(sr, eq->ev_hot, 0));
                eq->hot_cur = hot_next; hot_next = -1;
            }
        }
        // ---- route(k) on the builder's stream
        if ((rc = open_timing())) return rc;
        hipLaunchKernelGGL(k_zero_ctr, dim3(1), dim3(64), 0, sr, S.d_ctr);
        SF_CHECK_LAUNCH();
        uint4* bins = reinterpret_cast<uint4*>(S.words.p);
        uint32_t* fill_f = S.hist.p, *fill_b = fill_f + gm.n_bins, *cutmarks = fill_b + gm.n_bins;
        RouteArgs ra{d_ids, d_offsets + first, first, cnt, gm.tile, n_regions - 1u, (uint32_t)gm.cap, bins, fill_f, fill_b, cutmarks,
                     S.d_ctr + CTR_TMP, S.longl.p, eq->hot_bufs[eq->hot_cur].p, eq->arena.p, eq->table.p, S.d_ctr + CTR_HOT, eq->mix_mode, 0u, n_regions, 0u, 0u};
        const size_t route_lds = (size_t)2 * n_regions * 4 + (size_t)kPartWaves * (kStageWords / 4 + 4) * 16 + (size_t)kHotSlots * 12;
        hipLaunchKernelGGL(k_part_route<kFormDirect>, dim3(gm.n_blocks), dim3(kPartBlock), route_lds, sr, ra);
        SF_CHECK_LAUNCH();
        SF_HIP(hipEventRecord(S.ev_route, sr));
        // ---- (B) the hot table of the NEXT sub-batch, if due: behind insert(k - 1), ahead of insert(k)
        if (reads_tab >= (1ull << 22) && reads_tab >= 4 * eq->hot_reads && eq->hot_cap == eq->cap) {
            bool rebuilt = false;
            const int into = eq->hot_cur ^ 1;
            if ((rc = eq_hot_refresh(eq, n_regions, kHotSlots, si, into, reads_tab, true, &rebuilt))) return rc;
            if (rebuilt) { SF_HIP(hipEventRecord(eq->ev_hot, si)); hot_next = into; }
        }
        // ---- insert(k) on the second stream, behind route(k) (and, in stream order, behind insert(k - 1))
        SF_HIP(hipStreamWaitEvent(si, S.ev_route, 0));
        PartArgs pa{eq->table.p, bins, fill_f, fill_b, gm.n_blocks, (uint32_t)gm.cap, eq->cls_hash.p, eq->cls_off.p, eq->cls_len.p, eq->cls_slot.p,
                    eq->arena.p, S.d_ctr, eq->d_ctr + CTR_ARENA, eq->d_ctr + CTR_GCLS, S.deferred.p, eq->mix_mode, 0u, 1u, eq->probe.p};
        hipLaunchKernelGGL(k_part_insert, dim3(n_regions), dim3(kPartBlock), 0, si, pa);
        SF_CHECK_LAUNCH();
        SF_HIP(hipMemcpyAsync(S.h_ctr, S.d_ctr, CTR_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, si));
        SF_HIP(hipMemcpyAsync(S.h_ctr + CTR_N, eq->d_ctr, CTR_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, si));
        SF_HIP(hipEventRecord(S.ev_ins, si));
        S.first = first; S.cnt = cnt; S.n_words = n_words; S.n_blocks = gm.n_blocks; S.cap = gm.cap; S.in_flight = true;
        ++n_fly; fly_reads += cnt; fly_words += n_words;
        first += cnt; ++k;
        // ---- the host now looks at the sub-batch BEFORE the one it just queued
        sfgpu_eq::PartSet& P = eq->pset[k & 1];            // (k was advanced: this is set (k - 2) & 1 = the older one in flight)
        if (P.in_flight && (rc = collect(P))) return rc;
        if (scout) {                                        // the scout is extra: drained, so that the bulk starts with its hot classes
            scout = false; step = usual_step;
            if ((rc = drain())) return rc;
            continue;
        }
        // sub-batches grow (x4, x2) while, at the rate new classes appeared in the last collected one, the table stays under half full
        for (uint32_t mult = 4; mult >= 2; mult /= 2) {
            const uint64_t next = (uint64_t)step * mult;
            if (next <= kMaxSubBatch && (double)eq->n_classes + rate * (double)(fly_reads + next) <= (double)(eq->cap / 2)) { step = (uint32_t)next; break; }
        }
    }
    // the end of the batch: everything collected, deferred reads replayed, the builder's stream behind both passes
    {
        if ((rc = drain())) return rc;
        eq->stats.pipeline_drains--;                        // (the final one is not an interruption)
        if (hot_next >= 0) { eq->hot_cur = hot_next; hot_next = -1; }      // (rebuilt for a sub-batch that did not come: complete -- both streams are idle)
    }
    *consumed = first;
    return SFGPU_OK;
}
#endif

// caller holds eq->mu
static int eq_add_locked(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t n_reads,
                         const uint64_t* d_weights = nullptr) {
    SF_REQUIRE(n_reads < kArenaBit, SFGPU_ERR_RANGE, "sfgpu_eq_add_batch: a batch holds < 2^31 reads");
    SF_REQUIRE(!eq->finished, SFGPU_ERR_STATE, "sfgpu_eq_add_batch: builder already finished (call start)");
    if (n_reads == 0) return SFGPU_OK;
    hipStream_t st = eq->stream;
    // big unweighted batches take the radix-partitioned path, everything else the generic one
    const bool part = eq->use_part && !d_weights && n_reads >= (1u << 16);
    uint32_t step = part ? eq->part_sub_batch : eq->sub_batch;
    // offsets the host has seen: (position, value).  A partitioned sub-batch brings the possible ends of the next one back
    // with its counters (k_sub_batch_begin), so the loop below needs a round trip of its own only when it has to guess again
    uint64_t known_pos[8]; uint32_t known_val[8]; int n_known = 0;
    auto remember = [&](uint64_t pos, uint32_t val) { known_pos[n_known & 7] = pos; known_val[n_known & 7] = val; ++n_known; };
    auto recall = [&](uint64_t pos, uint32_t* val) -> bool {
        for (int i = 0; i < (n_known < 8 ? n_known : 8); ++i) if (known_pos[i] == pos) { *val = known_val[i]; return true; }
        return false;
    };
    const bool adaptive = part && getenv("SFGPU_EQ_SUBBATCH") == nullptr;
    // the very first sub-batch of a builder is small: it shows which classes are hot (eq_partitioned) at the price of one launch
    const uint32_t usual_step = step;
    bool scout = adaptive && eq->reads_seen == 0 && n_reads > (1u << 22) && step > kScoutReads;
    if (scout) step = kScoutReads;
    // (read into the pinned counter block -- slots past the counters -- so that the three copies are queued and cost one wait:
    //  a copy into pageable memory is a round trip of its own)
    uint32_t* ends = reinterpret_cast<uint32_t*>(eq->h_ctr + CTR_N);
    const uint64_t first_end = (n_reads < step) ? n_reads : step;
    SF_HIP(hipMemcpyAsync(&ends[0], d_offsets, 4, hipMemcpyDeviceToHost, st));
    SF_HIP(hipMemcpyAsync(&ends[1], d_offsets + n_reads, 4, hipMemcpyDeviceToHost, st));
    SF_HIP(hipMemcpyAsync(&ends[2], d_offsets + first_end, 4, hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    remember(0, ends[0]); remember(n_reads, ends[1]); remember(first_end, ends[2]);
    SF_REQUIRE(ends[1] >= ends[0], SFGPU_ERR_INVALID, "sfgpu_eq_add_batch: offsets not ascending");
    uint64_t batch_ids = (uint64_t)ends[1] - ends[0];
    int rc;
    // arena room for the reads about to be inserted -- worst case every read opens a class: its ids + the
    // entry header and padding (<= 4 words) + the probe granule in front of the entry (4 words).  Reserved per sub-batch on the partitioned path (a 400 M-read
    // batch would otherwise pin 13 GB for ~30 MB of labels), once per batch on the generic one.
    auto reserve_arena = [&](uint64_t words, uint64_t reads) -> int {
        const uint64_t need = eq->arena_used + words + 8 * reads + 4;
        SF_REQUIRE((need >> 2) < kArenaBit, SFGPU_ERR_RANGE, "sfgpu_eq_add_batch: label arena would exceed 2^33 words");
        return eq->arena.reserve(need, st, true, eq->arena_used);
    };

    if (!part && (rc = reserve_arena(batch_ids, n_reads))) return rc;
    uint32_t first0 = 0;
    const bool ring_wanted = []() { const char* e = SF_DEV_ENV("SFGPU_EQ_RING"); return e && atoi(e) != 0; }();
#ifdef SFGPU_VARIANTS
    if (part && adaptive && eq->pipe_mode != 0 && !ring_wanted && n_reads >= (1u << 22)) {
        // large batches: the sub-batches are queued ahead of the host (mode 1: route(k + 1) next to insert(k))
        if ((rc = eq_pipeline(eq, d_ids, d_offsets, n_reads, &first0))) return rc;
        if (first0) { scout = false; step = usual_step; }      // (whatever is left takes the serial loop at the usual size)
    }
#else
    (void)ring_wanted;
#endif
    // Sub-batches bound the partition buffer and let the table grow between them.  Each one shows how fast
    // new classes appear (the rate only falls as the table fills); the next sub-batch is made up to four
    // times larger (at most 2^26 reads) as long as, at that rate, the table would stay under half full:
    // fewer launches and longer region segments (400 M reads: 24 sub-batches -> 8).
    bool settled = first0 == n_reads && first0 != 0;          // (the pipeline ends drained)
    for (uint32_t first = first0; first < n_reads; ) {
        uint32_t cnt = (n_reads - first < step) ? (n_reads - first) : step;
        const uint64_t classes_before = eq->n_classes;
        bool done = false;
        if (part && (eq->cap >> kRegionBits) <= (uint64_t)kMaxRegions) {
            uint32_t se[2];
            if (!recall(first, &se[0]) || !recall((uint64_t)first + cnt, &se[1])) {
                SF_HIP(hipMemcpyAsync(&se[0], d_offsets + first, 4, hipMemcpyDeviceToHost, st));
                SF_HIP(hipMemcpyAsync(&se[1], d_offsets + first + cnt, 4, hipMemcpyDeviceToHost, st));
                SF_HIP(hipStreamSynchronize(st));
            }
            uint64_t n_words = (uint64_t)se[1] - se[0];
            if (n_words >= (1ull << 31) && cnt > (1u << 20)) { step = cnt / 2; continue; }     // too many ids for 31-bit offsets: halve
            if ((rc = reserve_arena(n_words, cnt))) return rc;
            if (n_words < (1ull << 31) && n_words >= 64) {       // (a sub-batch of next to no ids is not worth a partition)
                // the next sub-batch starts at `nf` and is step, 2 step or 4 step reads long (see below)
                const uint64_t nf = (uint64_t)first + cnt, left = n_reads - nf;
                uint64_t peek[3];
                for (int i = 0; i < 3; ++i) { const uint64_t len = (uint64_t)(scout ? usual_step : step) << i; peek[i] = nf + (len < left ? len : left); }
                if ((rc = eq_partitioned(eq, d_ids, d_offsets, first, cnt, n_words, peek))) return rc;
                remember(nf, se[1]);
                for (int i = 0; i < 3; ++i) remember(peek[i], (uint32_t)eq->h_ctr[CTR_PEEK + i]);
                done = true;
            }
        } else if (part) {
            // table already beyond the partition kernels: this sub-batch goes to the generic kernel
            uint32_t se[2];
            SF_HIP(hipMemcpyAsync(&se[0], d_offsets + first, 4, hipMemcpyDeviceToHost, st));
            SF_HIP(hipMemcpyAsync(&se[1], d_offsets + first + cnt, 4, hipMemcpyDeviceToHost, st));
            SF_HIP(hipStreamSynchronize(st));
            if ((rc = reserve_arena((uint64_t)(se[1] - se[0]), cnt))) return rc;
        }
        if (!done) {
            for (uint32_t f2 = first; f2 < first + cnt; f2 += eq->sub_batch) {
                uint32_t c2 = (first + cnt - f2 < eq->sub_batch) ? (first + cnt - f2) : eq->sub_batch;
                if ((rc = eq_generic(eq, d_ids, d_offsets, f2, c2, nullptr, d_weights))) return rc;
            }
            if (part) {     // the next sub-batch's reservation starts from the real cursor
                SF_HIP(hipMemcpyAsync(eq->h_ctr + CTR_ARENA, eq->d_ctr + CTR_ARENA, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                SF_HIP(hipStreamSynchronize(st));
                eq->arena_used = eq->h_ctr[CTR_ARENA];
            }
        }
        first += cnt;
        eq->reads_seen += cnt;
        settled = done && eq->settled;   // (a partitioned sub-batch returns with the stream waited for and the arena cursor read)
        if (scout) { scout = false; step = usual_step; continue; }       // the scout sub-batch is extra: the usual sizes follow
        if (adaptive && done) {
            const double rate = (double)(eq->n_classes - classes_before + 1) / (double)cnt;
            for (uint32_t mult = 4; mult >= 2; mult /= 2) {
                const uint64_t next = (uint64_t)step * mult;
                if (next <= kMaxSubBatch && (double)eq->n_classes + rate * (double)next <= (double)(eq->cap / 2)) { step = (uint32_t)next; break; }
            }
        }
    }
    // the caller may reuse its buffers on return: wait for the last commit (a batch that ended in a partitioned sub-batch has)
    if (!settled) {
        SF_HIP(hipMemcpyAsync(eq->h_ctr + CTR_ARENA, eq->d_ctr + CTR_ARENA, sizeof(unsigned long long),
                              hipMemcpyDeviceToHost, st));
        SF_HIP(hipStreamSynchronize(st));
        eq->arena_used = eq->h_ctr[CTR_ARENA];
    }
    return SFGPU_OK;
}

// offsets of a small device batch, moved behind what the accumulation buffer already holds (modulo 2^32 like everything else)
__global__ void k_rebase_offsets(const uint32_t* __restrict__ src, uint32_t n, uint32_t shift, uint32_t* __restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) dst[i] = src[i] + shift;
}

constexpr uint32_t kDevAccMin = 1u << 23;        // device batches below this are gathered (a 4 M-read batch built on its own costs 0.3 ms)
constexpr uint32_t kDevAccReads = 1u << 24;      // ... up to this many reads
constexpr uint64_t kDevAccIds = 1ull << 27;      // ... or this many ids

// caller holds eq->mu: build the gathered device batches
static int eq_flush_dacc_locked(sfgpu_eq* eq) {
    if (eq->dacc_n_reads == 0) return SFGPU_OK;
    const uint32_t n = eq->dacc_n_reads;
    eq->dacc_n_reads = 0; eq->dacc_n_ids = 0;
    return eq_add_locked(eq, eq->dacc_ids.p, eq->dacc_off.p, n);
}

// caller holds eq->mu: copy a small device batch behind the gathered ones (the caller may reuse its buffers on return)
static int eq_dacc_append_locked(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t n_reads) {
    hipStream_t st = eq->stream;
    uint32_t ends[2];
    SF_HIP(hipMemcpyAsync(&ends[0], d_offsets, 4, hipMemcpyDeviceToHost, st));
    SF_HIP(hipMemcpyAsync(&ends[1], d_offsets + n_reads, 4, hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    SF_REQUIRE(ends[1] >= ends[0], SFGPU_ERR_INVALID, "sfgpu_eq_add_batch: offsets not ascending");
    const uint64_t n_ids = (uint64_t)ends[1] - ends[0];
    int rc;
    if (n_ids > kDevAccIds / 2) {                       // (few reads, very long labels: not worth gathering)
        if ((rc = eq_flush_dacc_locked(eq))) return rc;
        return eq_add_locked(eq, d_ids, d_offsets, n_reads);
    }
    if (eq->dacc_n_reads + (uint64_t)n_reads > kDevAccReads || eq->dacc_n_ids + n_ids > kDevAccIds)
        if ((rc = eq_flush_dacc_locked(eq))) return rc;
    // (the buffers grow with what is gathered, doubling: a 10 000-read job does not pin half a gigabyte)
    if ((rc = eq->dacc_ids.reserve(std::max<uint64_t>(eq->dacc_n_ids + n_ids + 8, 1u << 20), st, true, eq->dacc_n_ids)) ||
        (rc = eq->dacc_off.reserve(std::max<uint64_t>((uint64_t)eq->dacc_n_reads + n_reads + 2, 1u << 18), st, true, (uint64_t)eq->dacc_n_reads + 1))) return rc;
    if (n_ids) SF_HIP(hipMemcpyAsync(eq->dacc_ids.p + eq->dacc_n_ids, d_ids + ends[0], n_ids * 4, hipMemcpyDeviceToDevice, st));
    const uint32_t shift = (uint32_t)eq->dacc_n_ids - ends[0];
    hipLaunchKernelGGL(k_rebase_offsets, dim3((n_reads + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, d_offsets, n_reads, shift,
                       eq->dacc_off.p + eq->dacc_n_reads);
    SF_CHECK_LAUNCH();
    SF_HIP(hipStreamSynchronize(st));
    eq->dacc_n_reads += n_reads; eq->dacc_n_ids += n_ids;
    return SFGPU_OK;
}

// caller holds eq->mu: build the accumulated host batch
static int eq_flush_acc_locked(sfgpu_eq* eq) {
    if (eq->acc_n_reads == 0) return SFGPU_OK;
    while (eq->acc_writers.load(std::memory_order_acquire) != 0) std::this_thread::yield();   // ranges reserved earlier are filled outside the lock
    const uint32_t n = eq->acc_n_reads; const uint64_t n_ids = eq->acc_n_ids;
    eq->acc_off[n] = (uint32_t)n_ids;
    int rc;
    if ((rc = eq->stage_ids.reserve(n_ids + 1, eq->stream, false))) return rc;
    if ((rc = eq->stage_off.reserve((uint64_t)n + 1, eq->stream, false))) return rc;
    if (n_ids) SF_HIP(hipMemcpyAsync(eq->stage_ids.p, eq->acc_ids, n_ids * 4, hipMemcpyHostToDevice, eq->stream));
    SF_HIP(hipMemcpyAsync(eq->stage_off.p, eq->acc_off, ((uint64_t)n + 1) * 4, hipMemcpyHostToDevice, eq->stream));
    eq->acc_n_reads = 0; eq->acc_n_ids = 0;
    return eq_add_locked(eq, eq->stage_ids.p, eq->stage_off.p, n);     // returns after the stream has drained
}

int sfgpu_eq_add_batch_host(sfgpu_eq* eq, const uint32_t* h_ids, const uint32_t* h_offsets, uint32_t n_reads) {
    SF_REQUIRE(eq && h_offsets, SFGPU_ERR_INVALID, "sfgpu_eq_add_batch_host: null pointer");
    if (n_reads == 0) return SFGPU_OK;
    SF_REQUIRE(h_offsets[n_reads] >= h_offsets[0], SFGPU_ERR_INVALID, "sfgpu_eq_add_batch_host: offsets not ascending");
    const uint32_t base = h_offsets[0];
    const uint64_t n_ids = (uint64_t)h_offsets[n_reads] - base;
    SF_REQUIRE(n_ids == 0 || h_ids, SFGPU_ERR_INVALID, "sfgpu_eq_add_batch_host: null ids");
    // mapping threads call this concurrently: the lock covers only the reservation of a range of the
    // accumulation buffer (and a flush when it is full); the copy itself runs outside it
    uint32_t* dst_ids; uint32_t* dst_off; uint32_t shift;
    {
        std::lock_guard<std::mutex> lk(eq->mu);
        SF_REQUIRE(!eq->finished, SFGPU_ERR_STATE, "sfgpu_eq_add_batch: builder already finished (call start)");
        int rc;
        if (n_reads >= kAccReads / 2 || n_ids >= kAccIds / 2) {
            // Already a large batch: it streams through two device staging buffers in chunks of kHostChunkReads reads --
            // the copy of chunk k + 1 (its own stream) runs while chunk k is built, so a batch that sits in pinned host
            // memory costs its PCIe transfer plus the build of the LAST chunk (SURVEY 8d's timed region: "packed hit lists
            // resident in host-pinned memory -> final device CSR").  Pageable memory works too (the runtime stages it).
            if (!eq->copy_stream) {
                SF_HIP(stream_acquire(&eq->copy_stream));
                SF_HIP(hipEventCreateWithFlags(&eq->ev_copy[0], hipEventDisableTiming));
                SF_HIP(hipEventCreateWithFlags(&eq->ev_copy[1], hipEventDisableTiming));
            }
            const uint32_t chunk_reads = []() { const char* e = getenv("SFGPU_EQ_HOST_CHUNK"); long v = e ? atol(e) : 0; return (uint32_t)(v >= 1024 ? v : kHostChunkReads); }();
            auto chunk_end = [&](uint32_t r0) -> uint32_t {                       // reads [r0, r1): <= chunk_reads reads, <= kHostChunkIds ids
                uint32_t r1 = (n_reads - r0 > chunk_reads) ? r0 + chunk_reads : n_reads;
                while (r1 > r0 + 1 && (uint64_t)h_offsets[r1] - h_offsets[r0] > kHostChunkIds) r1 = r0 + (r1 - r0) / 2;
                return r1;
            };
            auto enqueue_copy = [&](uint32_t r0, uint32_t r1, int slot) -> int {
                const uint64_t c_ids = (uint64_t)h_offsets[r1] - h_offsets[r0];
                const uint32_t c = r1 - r0;
                int rc2;
                if ((rc2 = eq->stage2_ids[slot].reserve(c_ids + 1, eq->copy_stream, false))) return rc2;
                if ((rc2 = eq->stage2_off[slot].reserve((uint64_t)c + 1, eq->copy_stream, false))) return rc2;
                if (c_ids) SF_HIP(hipMemcpyAsync(eq->stage2_ids[slot].p, h_ids + h_offsets[r0], c_ids * 4, hipMemcpyHostToDevice, eq->copy_stream));
                SF_HIP(hipMemcpyAsync(eq->stage2_off[slot].p, h_offsets + r0, ((uint64_t)c + 1) * 4, hipMemcpyHostToDevice, eq->copy_stream));
                if (h_offsets[r0]) {     // offsets relative to the staged ids
                    hipLaunchKernelGGL(k_rebase, dim3(grid_for((uint64_t)c + 1)), dim3(kBlock), 0, eq->copy_stream, eq->stage2_off[slot].p,
                                       (uint64_t)c + 1, h_offsets[r0]);
                    SF_CHECK_LAUNCH();
                }
                SF_HIP(hipEventRecord(eq->ev_copy[slot], eq->copy_stream));
                return SFGPU_OK;
            };
            uint32_t r0 = 0, r1 = chunk_end(0);
            int slot = 0;
            if ((rc = enqueue_copy(r0, r1, slot))) return rc;
            while (r0 < n_reads) {
                const uint32_t nr0 = r1, nr1 = (nr0 < n_reads) ? chunk_end(nr0) : nr0;
                // the other buffer is free: the build that used it has returned (eq_add_locked drains the stream)
                if (nr0 < n_reads && (rc = enqueue_copy(nr0, nr1, slot ^ 1))) return rc;
                SF_HIP(hipStreamWaitEvent(eq->stream, eq->ev_copy[slot], 0));
                if ((rc = eq_add_locked(eq, eq->stage2_ids[slot].p, eq->stage2_off[slot].p, r1 - r0))) return rc;
                r0 = nr0; r1 = nr1; slot ^= 1;
            }
            return SFGPU_OK;
        }
        if (!eq->acc_ids) {
            SF_HIP(pinned_malloc(&eq->acc_ids, kAccIds * 4));
            SF_HIP(pinned_malloc(&eq->acc_off, ((uint64_t)kAccReads + 1) * 4));
        }
        if (eq->acc_n_reads + n_reads > kAccReads || eq->acc_n_ids + n_ids > kAccIds)
            if ((rc = eq_flush_acc_locked(eq))) return rc;
        dst_ids = eq->acc_ids + eq->acc_n_ids;
        dst_off = eq->acc_off + eq->acc_n_reads;
        shift = (uint32_t)eq->acc_n_ids - base;                          // modulo 2^32: o[i] = acc_n_ids + (h_off[i] - base)
        eq->acc_n_reads += n_reads; eq->acc_n_ids += n_ids;
        eq->acc_writers.fetch_add(1, std::memory_order_acq_rel);
    }
    if (n_ids) memcpy(dst_ids, h_ids + base, n_ids * 4);
    for (uint32_t i = 0; i < n_reads; ++i) dst_off[i] = h_offsets[i] + shift;
    eq->acc_writers.fetch_sub(1, std::memory_order_release);
    return SFGPU_OK;
}

int sfgpu_eq_add_batch_device(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t n_reads) {
    SF_REQUIRE(eq && d_offsets, SFGPU_ERR_INVALID, "sfgpu_eq_add_batch: null pointer");
    std::lock_guard<std::mutex> lk(eq->mu);
    SF_REQUIRE(!eq->finished, SFGPU_ERR_STATE, "sfgpu_eq_add_batch: builder already finished (call start)");
    if (n_reads == 0) return SFGPU_OK;
    if (eq->use_part && n_reads < kDevAccMin && getenv("SFGPU_EQ_SUBBATCH") == nullptr) return eq_dacc_append_locked(eq, d_ids, d_offsets, n_reads);
    int rc = eq_flush_dacc_locked(eq);
    return rc ? rc : eq_add_locked(eq, d_ids, d_offsets, n_reads);
}

int sfgpu_eq_add_weighted_device(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets,
                                 const uint64_t* d_counts, uint32_t n_groups) {
    SF_REQUIRE(eq && d_offsets && d_counts, SFGPU_ERR_INVALID, "sfgpu_eq_add_weighted: null pointer");
    std::lock_guard<std::mutex> lk(eq->mu);
    int rc = eq_flush_dacc_locked(eq);                      // small device batches gathered earlier are built first (errors surface here, in order)
    return rc ? rc : eq_add_locked(eq, d_ids, d_offsets, n_groups, d_counts);
}

int sfgpu_eq_get_stats(sfgpu_eq* eq, sfgpu_eq_stats* out) {
    SF_REQUIRE(eq && out, SFGPU_ERR_INVALID, "sfgpu_eq_get_stats: null pointer");
    std::lock_guard<std::mutex> lk(eq->mu);
    // small batches are gathered before they are built (host: acc, device: dacc): build them so that the statistics -- and the
    // class count a caller may be watching -- cover every read handed over so far
    int rc;
    if (!eq->finished) { if ((rc = eq_flush_acc_locked(eq))) return rc; if ((rc = eq_flush_dacc_locked(eq))) return rc; }
    *out = eq->stats;
    out->table_slots = eq->cap;
    return SFGPU_OK;
}

int sfgpu_eq_finish(sfgpu_eq* eq, uint64_t* n_classes, uint64_t* nnz, uint64_t* total_reads) {
    SF_REQUIRE(eq, SFGPU_ERR_INVALID, "sfgpu_eq_finish: null handle");
    std::lock_guard<std::mutex> lk(eq->mu);
    hipStream_t st = eq->stream;
    int rc;
#ifdef SFGPU_X_EQ_STAMP
    {   // dev: how the blocks of the LAST route / insert launch spread (us)
        (void)hipStreamSynchronize(st);
        static unsigned long long h[2][2][4096];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(sfgpu::g_eq_stamp), sizeof(h)) == hipSuccess) {
            for (int k = 0; k < 2; ++k) {
                unsigned long long t0 = ~0ull, t1 = 0; double sum = 0, mx = 0; int n = 0; std::vector<double> d;
                for (int b = 0; b < 4096; ++b) if (h[k][0][b] && h[k][1][b] > h[k][0][b]) { t0 = std::min(t0, h[k][0][b]); t1 = std::max(t1, h[k][1][b]); }
                for (int b = 0; b < 4096; ++b) if (h[k][0][b] && h[k][1][b] > h[k][0][b]) { const double x = (double)(h[k][1][b] - h[k][0][b]) * 0.01; d.push_back(x); sum += x; mx = std::max(mx, x); ++n; }
                if (!n) continue;
                std::sort(d.begin(), d.end());
                double late = 0; for (int b = 0; b < 4096; ++b) if (h[k][0][b] && h[k][1][b] > h[k][0][b]) late = std::max(late, (double)(h[k][0][b] - t0) * 0.01);
                fprintf(stderr, "eq stamps %s: %d blocks, span %.1f us, block mean %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f, last start +%.1f\n", k ? "insert" : "route", n,
                        (double)(t1 - t0) * 0.01, sum / n, d[n / 10], d[n / 2], d[n * 9 / 10], mx, late);
                // by block index mod 8 (the XCD a block lands on), by index / 256, and first round (start within 5 us) vs later
                double m8[8] = {0}, c8[8] = {0}, q[16] = {0}, cq[16] = {0}, r1 = 0, n1 = 0, r2 = 0, n2 = 0;
                for (int b = 0; b < 4096; ++b) if (h[k][0][b] && h[k][1][b] > h[k][0][b]) {
                    const double x = (double)(h[k][1][b] - h[k][0][b]) * 0.01;
                    m8[b & 7] += x; c8[b & 7] += 1; q[b >> 8] += x; cq[b >> 8] += 1;
                    if ((double)(h[k][0][b] - t0) * 0.01 < 5.0) { r1 += x; n1 += 1; } else { r2 += x; n2 += 1; }
                }
                fprintf(stderr, "   by index mod 8:");
                for (int i = 0; i < 8; ++i) fprintf(stderr, " %.1f", c8[i] ? m8[i] / c8[i] : 0.0);
                fprintf(stderr, " | by index / 256:");
                for (int i = 0; i < 16; ++i) if (cq[i]) fprintf(stderr, " %.1f", q[i] / cq[i]);
                fprintf(stderr, " | first round %.1f (%d blocks), later %.1f (%d)\n", n1 ? r1 / n1 : 0.0, (int)n1, n2 ? r2 / n2 : 0.0, (int)n2);
            }
        }
    }
#endif
    if ((rc = eq_flush_acc_locked(eq))) return rc;          // reads still waiting in the host accumulation buffer
    if ((rc = eq_flush_dacc_locked(eq))) return rc;         // ... and in the device one
    uint64_t n = eq->n_classes;
    if ((rc = eq->order.reserve(n + 1, st, false))) return rc;
    if ((rc = eq->rowptr64.reserve(n + 2, st, false))) return rc;
    eq->nnz = 0; eq->total_reads = 0;
    if (n) {
        DevBuf<uint64_t> keys_in, keys_out; DevBuf<uint32_t> vals_in, lens;
        if ((rc = keys_in.reserve(n, st, false)) || (rc = keys_out.reserve(n, st, false)) ||
            (rc = vals_in.reserve(n, st, false)) || (rc = lens.reserve(n + 1, st, false))) return rc;
        SF_HIP(hipMemsetAsync(eq->d_ctr + 3, 0, 2 * sizeof(unsigned long long), st));           // (max first id, total reads: slots 3 and 4 = CTR_TMP + 1)
        static_assert(CTR_N >= 5, "finish() uses counter slots 3 and 4");
        hipLaunchKernelGGL(k_sort_keys, dim3(grid_for(n) < 1024 ? grid_for(n) : 1024), dim3(kBlock), 0, st, n, eq->cls_hash.p, eq->cls_off.p,
                           eq->arena.p, keys_in.p, vals_in.p, eq->d_ctr + 3, eq->table.p, eq->cls_slot.p, eq->d_ctr + 4);
        SF_CHECK_LAUNCH();
        SF_HIP(hipMemcpyAsync(eq->h_ctr + 3, eq->d_ctr + 3, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        SF_HIP(hipStreamSynchronize(st));                       // (~15 us; each radix pass it saves costs ~35)
        eq->total_reads = eq->h_ctr[4];
        int key_bits = kSortHashBits + 1;
        while (key_bits < 64 && (eq->h_ctr[3] >> (key_bits - kSortHashBits)) != 0) ++key_bits;
        // (no host wait inside the sort and the scan: everything they touch lives until the synchronisation below)
        if ((rc = sort_pairs_u64_u32(keys_in.p, keys_out.p, vals_in.p, eq->order.p, n, st, key_bits, false))) return rc;
        hipLaunchKernelGGL(k_tie_fix, dim3(grid_for(n)), dim3(kBlock), 0, st, n, keys_out.p, eq->order.p, eq->cls_hash.p,
                           eq->cls_off.p, eq->cls_len.p, eq->arena.p);
        SF_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_sorted_lens, dim3(grid_for(n + 1)), dim3(kBlock), 0, st, n, eq->order.p, eq->cls_len.p, lens.p);
        SF_CHECK_LAUNCH();
        if ((rc = exclusive_scan_u32(lens.p, eq->rowptr64.p, n, st, false))) return rc;
        SF_HIP(hipMemcpyAsync(eq->h_ctr + CTR_NNZ, eq->rowptr64.p + n, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        SF_HIP(hipStreamSynchronize(st));
        eq->nnz = eq->h_ctr[CTR_NNZ];
    }
    eq->finished = true;
    if (n_classes) *n_classes = n;
    if (nnz) *nnz = eq->nnz;
    if (total_reads) *total_reads = eq->total_reads;
    // the two lines the reference logs in finish() (EquivalenceClassBuilder.hpp:75-78)
    log_msg(0, "Computed %llu rich equivalence classes for further processing", (unsigned long long)n);
    log_msg(0, "Counted %llu total reads in the equivalence classes ", (unsigned long long)eq->total_reads);
    return SFGPU_OK;
}

int sfgpu_eq_export_device(sfgpu_eq* eq, uint32_t* d_rowptr, uint32_t* d_ids, uint64_t* d_counts, uint64_t* d_hashes) {
    SF_REQUIRE(eq && d_rowptr, SFGPU_ERR_INVALID, "sfgpu_eq_export: null pointer");
    std::lock_guard<std::mutex> lk(eq->mu);
    SF_REQUIRE(eq->finished, SFGPU_ERR_STATE, "sfgpu_eq_export: call finish() first");
    SF_REQUIRE(eq->nnz < (1ull << 32), SFGPU_ERR_RANGE, "sfgpu_eq_export: nnz does not fit uint32 rowptr");
    uint64_t n = eq->n_classes;
    if (n == 0) { SF_HIP(hipMemsetAsync(d_rowptr, 0, 4, eq->stream)); return SFGPU_OK; }
    SF_REQUIRE(d_ids && d_counts, SFGPU_ERR_INVALID, "sfgpu_eq_export: null pointer");
    hipLaunchKernelGGL(k_export_classes, dim3(grid_for(n + 1)), dim3(kBlock), 0, eq->stream, n, eq->order.p, eq->rowptr64.p,
                       eq->table.p, eq->cls_slot.p, eq->cls_hash.p, d_rowptr, d_counts, d_hashes);
    SF_CHECK_LAUNCH();
    if (eq->nnz) {
        const uint64_t waves = (eq->nnz + kExportPerWave - 1) / kExportPerWave;
        hipLaunchKernelGGL(k_export_ids, dim3(grid_for(waves * kWave)), dim3(kBlock), 0, eq->stream, n, eq->nnz, eq->order.p, eq->rowptr64.p,
                           eq->cls_off.p, eq->arena.p, d_ids);
        SF_CHECK_LAUNCH();
    }
    return SFGPU_OK;
}

int sfgpu_eq_export_host(sfgpu_eq* eq, uint32_t* h_rowptr, uint32_t* h_ids, uint64_t* h_counts, uint64_t* h_hashes) {
    SF_REQUIRE(eq && h_rowptr, SFGPU_ERR_INVALID, "sfgpu_eq_export_host: null pointer");
    uint64_t n, nnz;
    { std::lock_guard<std::mutex> lk(eq->mu); n = eq->n_classes; nnz = eq->nnz;
      SF_REQUIRE(eq->finished, SFGPU_ERR_STATE, "sfgpu_eq_export_host: call finish() first"); }
    DevBuf<uint32_t> rp, ids; DevBuf<uint64_t> cnt, hs;
    hipStream_t st = eq->stream;
    int rc;
    if ((rc = rp.reserve(n + 1, st, false)) || (rc = ids.reserve(nnz + 1, st, false)) ||
        (rc = cnt.reserve(n + 1, st, false)) || (rc = hs.reserve(n + 1, st, false))) return rc;
    if ((rc = sfgpu_eq_export_device(eq, rp.p, ids.p, cnt.p, h_hashes ? hs.p : nullptr))) return rc;
    SF_HIP(hipMemcpyAsync(h_rowptr, rp.p, (n + 1) * 4, hipMemcpyDeviceToHost, st));
    if (n) {
        if (nnz) SF_HIP(hipMemcpyAsync(h_ids, ids.p, nnz * 4, hipMemcpyDeviceToHost, st));
        SF_HIP(hipMemcpyAsync(h_counts, cnt.p, n * 8, hipMemcpyDeviceToHost, st));
        if (h_hashes) SF_HIP(hipMemcpyAsync(h_hashes, hs.p, n * 8, hipMemcpyDeviceToHost, st));
    }
    SF_HIP(hipStreamSynchronize(st));
    return SFGPU_OK;
}

}  // extern "C"
