// mapper.hip -- a quasi-mapping front end on the device: reads in, sfgpu_hit records out.
//
// Stands where the reference calls RapMap: the SACollector / hit collection of processReadsQuasi
// (src/SailfishQuantify.cpp:141-142, 192-213 paired end, :487-488, 526-528 single end) produces, per read, the
// QuasiAlignment list that the hit-filtering loop (sfgpu_filter_hits) consumes.  RapMap is a third-party library that
// is fetched at build time (scripts/fetchRapMap.sh:20, COMBINE-lab/RapMap @ sf-v0.10.1) and is not in the reference
// tree: its suffix-array search is NOT what is implemented here, and parity with it is unpinned.  This is an
// exact-seed mapper with a contract of its own (restated on the CPU for the tests, which check it record for record
// and against the simulator's truth in the reference's bundled sample_data):
//
//   index : every k-mer (k <= 31, 2 bits per base) of every transcript made of A/C/G/T only, with (transcript, position);
//           built on the device: one pass packs the k-mers, a stable radix sort (rocPRIM) orders them -- occurrences of
//           a k-mer stay in (transcript, position) order -- and a table over the top bits of the k-mer bounds every
//           lookup to a handful of binary-search steps.
//   read  : two seeds, at offsets 0 and len - k, on the forward strand (fwd = 1) and as reverse complement (fwd = 0), in
//           the order fwd-seed0, fwd-seed1, rc-seed0, rc-seed1; at most max_occ occurrences per lookup; the first occurrence
//           seen for a (transcript, strand) fixes the read's position there (p - seed offset); hits sorted by
//           (transcript, strand).
//   pair  : every (left hit, right hit) on one transcript with opposite strands is a PAIRED_END_PAIRED record, fragment
//           length = max end - min start; without one, the left hits (PAIRED_END_LEFT) then the right hits
//           (PAIRED_END_RIGHT) are kept as orphans.  Single-end reads give SINGLE_END records.
// Kernels: lane per read (mate); variable-length outputs are produced count -> scan -> fill.
#include <vector>

#include "common.h"
#include "primitives.h"

namespace sfgpu {

constexpr int kMapBlock = 256;
static inline unsigned mpgrid(uint64_t n) { return (unsigned)((n + kMapBlock - 1) / kMapBlock); }

__device__ __forceinline__ uint32_t base_code(unsigned char b) {
    // A C G T (either case) -> 0 1 2 3, anything else -> 4
    const unsigned char u = b & 0xDFu;
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
}

// ---- index build -------------------------------------------------------------------------------------------------
// one lane per base position of the concatenated transcripts: the k-mer starting there (if it lies inside one transcript
// and holds A/C/G/T only); invalid positions get the key ~0 and sort to the end
__global__ void __launch_bounds__(kMapBlock)
k_index_kmers(const char* __restrict__ seq, const uint64_t* __restrict__ seq_off, const uint32_t* __restrict__ ref_len, uint64_t M, uint32_t k,
              const uint64_t* __restrict__ kmer_off /* [M + 1]: k-mer slots before transcript t */, uint64_t* keys, uint32_t* vals) {
    const uint32_t t = blockIdx.y + blockIdx.z * gridDim.y;
    if (t >= M) return;
    const uint32_t len = ref_len[t];
    if (len < k) return;
    const uint32_t n = len - k + 1;
    const char* s = seq + seq_off[t];
    const uint64_t o = kmer_off[t];
    for (uint32_t p = blockIdx.x * kMapBlock + threadIdx.x; p < n; p += gridDim.x * kMapBlock) {
        uint64_t key = 0; bool ok = true;
        for (uint32_t i = 0; i < k; ++i) { const uint32_t c = base_code((unsigned char)s[p + i]); ok = ok && c < 4u; key = (key << 2) | (c & 3u); }
        keys[o + p] = ok ? key : ~0ull;
        vals[o + p] = (uint32_t)(o + p);
    }
}
__global__ void k_index_lens(uint64_t M, const uint32_t* __restrict__ ref_len, uint32_t k, uint32_t* n_kmers) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) n_kmers[t] = ref_len[t] >= k ? ref_len[t] - k + 1 : 0u; else if (t == M) n_kmers[t] = 0u;
}
// occurrence slot (global k-mer slot) -> (transcript, position): binary search in kmer_off, once per occurrence at build time
__global__ void k_index_locate(uint64_t n, uint64_t M, const uint32_t* __restrict__ slot, const uint64_t* __restrict__ kmer_off,
                               uint32_t* tid, uint32_t* tpos) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = slot[i];
    uint64_t lo = 0, hi = M;                       // last t with kmer_off[t] <= s
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (kmer_off[mid] <= s) lo = mid; else hi = mid; }
    tid[i] = (uint32_t)lo; tpos[i] = (uint32_t)(s - kmer_off[lo]);
}
// bucket b (top `bits` bits of a 2k-bit key) -> first sorted index with key >= b << shift
__global__ void k_index_buckets(uint64_t n_valid, const uint64_t* __restrict__ keys, uint32_t shift, uint64_t n_buckets, uint32_t* start) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_buckets) return;
    if (b == n_buckets) { start[b] = (uint32_t)n_valid; return; }
    const uint64_t target = b << shift;
    uint64_t lo = 0, hi = n_valid;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (keys[mid] >= target) hi = mid; else lo = mid + 1; }
    start[b] = (uint32_t)lo;
}
__global__ void k_count_valid(uint64_t n, const uint64_t* __restrict__ keys, unsigned long long* out) {
    unsigned long long v = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v += keys[i] != ~0ull;
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && v) atomicAdd(out, v);
}

struct IndexView {
    const uint64_t* keys; const uint32_t* tid; const uint32_t* tpos; const uint32_t* bucket; uint64_t n; uint32_t k, shift, max_occ;
    uint32_t n_seeds;                      // seeds per strand (2: offsets 0 and len - k; more: spread evenly between them)
    // scan mode (seed_len != 0): the transcripts' text (the index's own copy) for the extension of a match past the seed
    uint32_t seed_len; const char* tseq; const uint64_t* tseq_off; const uint32_t* tlen;
};
constexpr uint32_t kScanGroups = 8;       // maximal matches kept per mate (both strands together)
constexpr uint32_t kMaxSeeds = 8;
// offset of seed j of S in a read of `len` bases (S = 2: 0 and len - k)
__device__ __host__ __forceinline__ uint32_t seed_offset(uint32_t j, uint32_t S, uint32_t len, uint32_t k) {
    return S <= 1 ? 0u : (uint32_t)(((uint64_t)j * (len - k)) / (S - 1));
}
// occurrences of `key`: [lo, lo + cnt)
__device__ __forceinline__ void index_lookup(const IndexView& x, uint64_t key, uint32_t& lo_out, uint32_t& cnt_out) {
    const uint64_t b = key >> x.shift;
    uint32_t lo = x.bucket[b], hi = x.bucket[b + 1];
    const uint32_t end = hi;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (x.keys[mid] >= key) hi = mid; else lo = mid + 1; }
    uint32_t e = lo;
    while (e < end && e - lo < x.max_occ && x.keys[e] == key) ++e;      // runs are short; max_occ bounds the walk
    lo_out = lo; cnt_out = e - lo;
}

// ---- mapping -----------------------------------------------------------------------------------------------------
// seed k-mer j of the read on one strand: fwd -> the window at offset o_j; rc -> the read's reverse complement at offset o_j,
// i.e. the complement of r[len - 1 - o_j .. len - o_j - k], reversed.  valid false if a base is not A/C/G/T.
__device__ __forceinline__ uint64_t seed_key(const char* r, uint32_t len, uint32_t k, uint32_t o, bool rc, bool& valid) {
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    uint64_t key = 0; bool ok = true;
    for (uint32_t i = 0; i < k; ++i) {
        const uint32_t c = base_code((unsigned char)(rc ? r[len - 1 - o - i] : r[o + i]));
        ok = ok && c < 4u;
        key = (key << 2) | ((rc ? 3u - c : c) & 3u);
    }
    valid = ok;
    return key & mask;
}

// S = 2 (the default), all four keys in one pass over the read: [fwd seed0, fwd seed1, rc seed0, rc seed1]; valid[i] false if a base is not A/C/G/T
__device__ __forceinline__ void seed_keys(const char* r, uint32_t len, uint32_t k, uint64_t (&key)[4], bool (&valid)[4]) {
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t o1 = len - k;
    uint64_t f0 = 0, f1 = 0, r0 = 0, r1 = 0; bool vf0 = true, vf1 = true, vr0 = true, vr1 = true;
    for (uint32_t i = 0; i < k; ++i) {
        const uint32_t a = base_code((unsigned char)r[i]), b = base_code((unsigned char)r[o1 + i]);
        vf0 = vf0 && a < 4u; vf1 = vf1 && b < 4u;
        f0 = (f0 << 2) | (a & 3u); f1 = (f1 << 2) | (b & 3u);
        // reverse complement of the read: its seed at offset 0 is the complement of r[len-1 .. len-k] (the read's LAST k
        // bases, i.e. the window at o1, reversed); its seed at offset len - k is that of r[k-1 .. 0]
        const uint32_t c = base_code((unsigned char)r[len - 1 - i]), d = base_code((unsigned char)r[k - 1 - i]);
        vr0 = vr0 && c < 4u; vr1 = vr1 && d < 4u;
        r0 = (r0 << 2) | ((3u - c) & 3u); r1 = (r1 << 2) | ((3u - d) & 3u);
    }
    key[0] = f0 & mask; key[1] = f1 & mask; key[2] = r0 & mask; key[3] = r1 & mask;
    valid[0] = vf0; valid[1] = vf1; valid[2] = vr0; valid[3] = vr1;
}

// pass A: the 2 S lookups of every mate (fwd seeds 0 .. S-1, then rc seeds 0 .. S-1); ranges[2 S m + i] = lo | cnt << 32;
// cand_cnt[m] = sum of the counts.  A seed whose offset equals the previous seed's (a read barely longer than k) is skipped.
__global__ void __launch_bounds__(kMapBlock)
k_map_lookup(IndexView x, const char* __restrict__ seq, const uint64_t* __restrict__ off, uint64_t n_mates, uint64_t* ranges, uint32_t* cand_cnt) {
    const uint64_t m = (uint64_t)blockIdx.x * kMapBlock + threadIdx.x;
    if (m > n_mates) return;
    if (m == n_mates) { cand_cnt[m] = 0; return; }
    const uint64_t b = off[m]; const uint32_t len = (uint32_t)(off[m + 1] - b);
    const uint32_t S = x.n_seeds;
    uint32_t total = 0;
    if (S == 2u) {                                          // the default: one fused pass computes the four keys
        uint64_t rg4[4] = {0, 0, 0, 0};
        if (len >= x.k) {
            uint64_t key[4]; bool valid[4];
            seed_keys(seq + b, len, x.k, key, valid);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t lo = 0, cnt = 0;
                if (valid[i] && !(i & 1 && len == x.k)) index_lookup(x, key[i], lo, cnt);     // (len == k: seed 1 is seed 0 again)
                rg4[i] = (uint64_t)lo | ((uint64_t)cnt << 32);
                total += cnt;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ranges[4 * m + i] = rg4[i];
        cand_cnt[m] = total;
        return;
    }
    for (uint32_t i = 0; i < 2 * S; ++i) {
        uint64_t rg = 0;
        if (len >= x.k) {
            const uint32_t j = i < S ? i : i - S;
            const uint32_t o = seed_offset(j, S, len, x.k);
            if (j == 0 || o != seed_offset(j - 1, S, len, x.k)) {
                bool valid; const uint64_t key = seed_key(seq + b, len, x.k, o, i >= S, valid);
                uint32_t lo = 0, cnt = 0;
                if (valid) index_lookup(x, key, lo, cnt);
                rg = (uint64_t)lo | ((uint64_t)cnt << 32);
                total += cnt;
            }
        }
        ranges[2ull * S * m + i] = rg;
    }
    cand_cnt[m] = total;
}

// candidate = tid << 32 | fwd << 31 | (pos + 2^30)   (sorting the words sorts by transcript, then strand)
__device__ __forceinline__ uint64_t cand_pack(uint32_t t, uint32_t fwd, int32_t pos) { return ((uint64_t)t << 32) | ((uint64_t)fwd << 31) | (uint32_t)(pos + (1 << 30)); }
__device__ __forceinline__ uint32_t cand_tid(uint64_t c) { return (uint32_t)(c >> 32); }
__device__ __forceinline__ uint32_t cand_fwd(uint64_t c) { return (uint32_t)(c >> 31) & 1u; }
__device__ __forceinline__ int32_t cand_pos(uint64_t c) { return (int32_t)((uint32_t)c & 0x7FFFFFFFu) - (1 << 30); }

// pass B: expand the ranges in the contract's order, keep the first occurrence per (transcript, strand) -- with S > 2 seeds
// also WHICH seeds hit it, and then only the (transcript, strand) pairs that the most seeds agree on --, sort by (transcript,
// strand).  Lists are short (a few entries); the dedupe and the insertion sort work in place in global memory.
__global__ void __launch_bounds__(kMapBlock)
k_map_hits(IndexView x, const uint64_t* __restrict__ off, uint64_t n_mates, const uint64_t* __restrict__ ranges,
           const uint64_t* __restrict__ cand_off, uint64_t* cand, uint8_t* votes, uint32_t* n_hits) {
    const uint64_t m = (uint64_t)blockIdx.x * kMapBlock + threadIdx.x;
    if (m >= n_mates) return;
    const uint32_t len = (uint32_t)(off[m + 1] - off[m]);
    const uint32_t S = x.n_seeds;
    uint64_t* out = cand + cand_off[m];
    uint8_t* vt = votes + cand_off[m];
    uint32_t n = 0;
    for (uint32_t i = 0; i < 2 * S; ++i) {
        const uint64_t rg = ranges[2ull * S * m + i];
        const uint32_t lo = (uint32_t)rg, cnt = (uint32_t)(rg >> 32);
        const uint32_t fwd = i < S ? 1u : 0u, j = i < S ? i : i - S;
        const int32_t o = cnt ? (int32_t)seed_offset(j, S, len, x.k) : 0;
        for (uint32_t q0 = 0; q0 < cnt; ++q0) {
            const uint32_t t = x.tid[lo + q0];
            uint32_t at = n;
            for (uint32_t q = 0; q < n && at == n; ++q) if (cand_tid(out[q]) == t && cand_fwd(out[q]) == fwd) at = q;
            if (at == n) { out[n] = cand_pack(t, fwd, (int32_t)x.tpos[lo + q0] - o); vt[n] = (uint8_t)(1u << j); ++n; }
            else vt[at] |= (uint8_t)(1u << j);
        }
    }
    if (S > 2 && n > 1) {                                   // the pairs the most seeds agree on
        uint32_t best = 0;
        for (uint32_t q = 0; q < n; ++q) { const uint32_t v = (uint32_t)__popc((unsigned)vt[q]); best = v > best ? v : best; }
        uint32_t w = 0;
        for (uint32_t q = 0; q < n; ++q) if ((uint32_t)__popc((unsigned)vt[q]) == best) out[w++] = out[q];
        n = w;
    }
    for (uint32_t a = 1; a < n; ++a) {                      // (transcript, strand) ascending: the key is the word's top 33 bits
        const uint64_t v = out[a]; uint32_t b = a;
        while (b > 0 && (out[b - 1] >> 31) > (v >> 31)) { out[b] = out[b - 1]; --b; }
        out[b] = v;
    }
    n_hits[m] = n;
}

// ---- scan mode: maximal-match extension (RapMap-style; parity with RapMap itself is unpinned -- it is not in the reference tree) ----
// The sorted k-mer table doubles as a suffix array of depth k: a seed of s <= k bases is a PREFIX range of it.  A mate is walked
// on both strands in lockstep: window at i -> prefix range; none, or more than max_occ -> i += 1; else every occurrence is extended
// base by base against the transcript's text; L = the longest extension, the occurrences that reach it form a GROUP (i, L);
// i += L - s + 1; a match that covers the whole read ends both walks.  Reads with substitutions map as long as s error-free bases remain somewhere an indexed k-mer starts in (not within the last k - s bases of a transcript or k - 1 bases upstream of an N) (the fixed 31-mer end seeds of the first
// contract lose a 50-base read to a single substitution in its middle).  The contract is restated on the CPU for the tests.
// first sorted index whose key is >= key (key < 4^k; key == 4^k: the end)
__device__ __forceinline__ uint32_t index_lower_bound(const IndexView& x, uint64_t key) {
    if (x.k < 32u && (key >> (2u * x.k)) != 0ull) return (uint32_t)x.n;
    const uint64_t b = key >> x.shift;
    uint32_t lo = x.bucket[b], hi = x.bucket[b + 1];
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (x.keys[mid] >= key) hi = mid; else lo = mid + 1; }
    return lo;
}
// base j of the mate on a strand: fwd -> r[j]; rc -> complement of r[n - 1 - j]; 4 = not A/C/G/T
__device__ __forceinline__ uint32_t strand_code(const char* r, uint32_t n, bool rc, uint32_t j) {
    const uint32_t c = base_code((unsigned char)(rc ? r[n - 1u - j] : r[j]));
    return c > 3u ? 4u : (rc ? 3u - c : c);
}
// how far q[i ...] and transcript t from p on agree, given that the first s bases do
__device__ __forceinline__ uint32_t scan_extend(const IndexView& x, const char* r, uint32_t n, bool rc, uint32_t i, uint32_t t, uint32_t p) {
    const char* ts = x.tseq + x.tseq_off[t];
    const uint32_t tl = x.tlen[t];
    uint32_t e = x.seed_len;
    while (i + e < n && p + e < tl) {
        const uint32_t cq = strand_code(r, n, rc, i + e);
        if (cq > 3u || base_code((unsigned char)ts[p + e]) != cq) break;
        ++e;
    }
    return e;
}
// group words: w0 = first sorted index | occurrences << 32;  w1 = i (24 bits) | L << 24 (24 bits) | fwd << 48 | 1 << 49 (present):
// mates of up to 2^24 - 1 bases (a longer mate is left unmapped: see kMaxScanMate; round 3 packed 16 bits each and a mate of
// >= 65536 bases spilled into the strand and present bits)
// pass A: walk the mate; groups[2 G m + 2 g ..], cand_cnt[m] = occurrences that reach L, summed over the groups.
// The two strands are walked in LOCKSTEP -- one lookup on the forward strand, one on the reverse complement, and so on -- and
// a match that covers the whole read ends both walks: an error-free read costs one or two lookups whichever strand it came
// from (walking the forward strand to its end first cost a reverse-strand read ~80 missed lookups: 188 ms instead of 35 ms per
// 10 M pairs of 2 x 100 bases).
__global__ void __launch_bounds__(kMapBlock)
k_scan_lookup(IndexView x, const char* __restrict__ seq, const uint64_t* __restrict__ off, uint64_t n_mates, uint64_t* groups, uint32_t* cand_cnt) {
    const uint64_t m = (uint64_t)blockIdx.x * kMapBlock + threadIdx.x;
    if (m > n_mates) return;
    if (m == n_mates) { cand_cnt[m] = 0; return; }
    const uint64_t b = off[m]; const uint32_t n = (uint32_t)(off[m + 1] - b);
    const char* r = seq + b;
    const uint32_t s = x.seed_len;
    uint64_t* out = groups + 2ull * kScanGroups * m;
    uint32_t ng = 0, total = 0;
    const uint64_t smask = (s == 32u) ? ~0ull : ((1ull << (2u * s)) - 1ull);
    const uint32_t sh = 2u * (x.k - s);
    // per strand: i = start of the window, key / have = the rolling window (have valid bases collected), live = still walking
    uint32_t wi[2] = {0u, 0u}, whave[2] = {0u, 0u};
    uint64_t wkey[2] = {0ull, 0ull};
    constexpr uint32_t kMaxScanMate = (1u << 24) - 1u;       // i and L travel in 24 bits each
    bool live[2] = {n >= s && n <= kMaxScanMate, n >= s && n <= kMaxScanMate};
    bool whole = false;
    while ((live[0] || live[1]) && !whole && ng < kScanGroups) {
#pragma unroll
        for (int strand = 0; strand < 2; ++strand) {
            if (!live[strand] || whole || ng >= kScanGroups) continue;
            const bool rc = strand == 1;
            uint32_t i = wi[strand], have = whave[strand];
            uint64_t key = wkey[strand];
            // bring the next window without a non-ACGT base into key (skipping such windows is not a step)
            while (have < s && i + s <= n) {
                const uint32_t c = strand_code(r, n, rc, i + have);
                if (c > 3u) { i += have + 1u; have = 0; key = 0; continue; }
                key = ((key << 2) | c) & smask; ++have;
            }
            if (have < s) { live[strand] = false; continue; }
            // ONE step: look the window up
            const uint32_t lo = index_lower_bound(x, key << sh);
            uint32_t hi = lo;
            while (hi < (uint32_t)x.n && hi - lo <= x.max_occ && (x.keys[hi] >> sh) == key) ++hi;      // (runs are short; max_occ bounds the walk)
            const uint32_t cnt = hi - lo;
            if (cnt == 0u || cnt > x.max_occ) { ++i; --have; }                        // a miss: slide by one (the oldest base leaves through smask)
            else {
                uint32_t L = 0, reach = 0;
                for (uint32_t q = lo; q < hi; ++q) {
                    const uint32_t e = scan_extend(x, r, n, rc, i, x.tid[q], x.tpos[q]);
                    if (e > L) { L = e; reach = 1; } else if (e == L) ++reach;
                }
                out[2u * ng] = (uint64_t)lo | ((uint64_t)cnt << 32);
                out[2u * ng + 1u] = (uint64_t)i | ((uint64_t)L << 24) | ((uint64_t)(rc ? 0u : 1u) << 48) | (1ull << 49);
                ++ng; total += reach;
                if (L == n) whole = true;                                               // the whole read matched: both walks end
                i += L - s + 1u; have = 0; key = 0;
            }
            if (i + s > n) live[strand] = false;
            wi[strand] = i; whave[strand] = have; wkey[strand] = key;
        }
    }
    for (uint32_t g = ng; g < kScanGroups; ++g) { out[2u * g] = 0; out[2u * g + 1u] = 0; }
    cand_cnt[m] = total;
}
// pass B: the occurrences that reach their group's L, in group order -> first position per (transcript, strand), one vote per
// group; the pairs with the most votes; sorted by (transcript, strand)
__global__ void __launch_bounds__(kMapBlock)
k_scan_hits(IndexView x, const char* __restrict__ seq1, const uint64_t* __restrict__ off1, const char* __restrict__ seq2, const uint64_t* __restrict__ off2,
            int paired, uint64_t n_mates, const uint64_t* __restrict__ groups, const uint64_t* __restrict__ cand_off, uint64_t* cand, uint8_t* votes, uint32_t* n_hits) {
    const uint64_t m = (uint64_t)blockIdx.x * kMapBlock + threadIdx.x;
    if (m >= n_mates) return;
    const uint64_t rd = paired ? (m >> 1) : m;
    const bool second = paired && (m & 1ull);
    const uint64_t* off = second ? off2 : off1;
    const char* r = (second ? seq2 : seq1) + off[rd];
    const uint32_t n = (uint32_t)(off[rd + 1] - off[rd]);
    uint64_t* out = cand + cand_off[m];
    uint8_t* vt = votes + cand_off[m];
    const uint64_t* gw = groups + 2ull * kScanGroups * m;
    uint32_t nc = 0;
    for (uint32_t g = 0; g < kScanGroups; ++g) {
        const uint64_t w0 = gw[2u * g], w1 = gw[2u * g + 1u];
        if (!(w1 >> 49)) break;
        const uint32_t lo = (uint32_t)w0, cnt = (uint32_t)(w0 >> 32);
        const uint32_t i = (uint32_t)(w1 & 0xFFFFFFu), L = (uint32_t)((w1 >> 24) & 0xFFFFFFu), fwd = (uint32_t)(w1 >> 48) & 1u;
        for (uint32_t q = lo; q < lo + cnt; ++q) {
            const uint32_t t = x.tid[q], p = x.tpos[q];
            if (scan_extend(x, r, n, fwd == 0u, i, t, p) != L) continue;
            uint32_t at = nc;
            for (uint32_t c = 0; c < nc && at == nc; ++c) if (cand_tid(out[c]) == t && cand_fwd(out[c]) == fwd) at = c;
            if (at == nc) { out[nc] = cand_pack(t, fwd, (int32_t)p - (int32_t)i); vt[nc] = (uint8_t)(1u << g); ++nc; }
            else vt[at] |= (uint8_t)(1u << g);
        }
    }
    if (nc > 1) {                                            // the pairs the most groups agree on
        uint32_t best = 0;
        for (uint32_t c = 0; c < nc; ++c) { const uint32_t v = (uint32_t)__popc((unsigned)vt[c]); best = v > best ? v : best; }
        uint32_t w = 0;
        for (uint32_t c = 0; c < nc; ++c) if ((uint32_t)__popc((unsigned)vt[c]) == best) out[w++] = out[c];
        nc = w;
    }
    for (uint32_t a = 1; a < nc; ++a) {
        const uint64_t v = out[a]; uint32_t bpos = a;
        while (bpos > 0 && (out[bpos - 1] >> 31) > (v >> 31)) { out[bpos] = out[bpos - 1]; --bpos; }
        out[bpos] = v;
    }
    n_hits[m] = nc;
}
// where the transcripts' text ends (max over t of seq_off[t] + ref_len[t]): the index copies that much
__global__ void k_seq_extent(uint64_t M, const uint64_t* __restrict__ seq_off, const uint32_t* __restrict__ ref_len, unsigned long long* out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) atomicMax(out, (unsigned long long)(seq_off[t] + ref_len[t]));
}

// pass C / D: records per read, then the records.  Paired: mates 2r (left) and 2r + 1 (right).
template <bool FILL>
__global__ void __launch_bounds__(kMapBlock)
k_map_records(uint64_t n_reads, int paired, const uint64_t* __restrict__ off, const uint64_t* __restrict__ cand_off, const uint64_t* __restrict__ cand,
              const uint32_t* __restrict__ n_hits, uint32_t* rec_cnt, const uint64_t* __restrict__ rec_off, sfgpu_hit* hits) {
    const uint64_t r = (uint64_t)blockIdx.x * kMapBlock + threadIdx.x;
    if (r > n_reads) return;
    if (r == n_reads) { if (!FILL) rec_cnt[r] = 0; return; }
    const uint64_t ml = paired ? 2 * r : r;
    const uint64_t* L = cand + cand_off[ml]; const uint32_t nl = n_hits[ml];
    const uint16_t len1 = (uint16_t)(off[ml + 1] - off[ml]);
    sfgpu_hit* out = FILL ? hits + rec_off[r] : nullptr;
    uint32_t n = 0;
    if (!paired) {
        if (FILL) for (uint32_t i = 0; i < nl; ++i) out[i] = sfgpu_hit{cand_tid(L[i]), cand_pos(L[i]), 0, 0u, len1, 0, (uint8_t)cand_fwd(L[i]), 0, 0, 0};
        n = nl;
    } else {
        const uint64_t* R = cand + cand_off[ml + 1]; const uint32_t nr = n_hits[ml + 1];
        const uint16_t len2 = (uint16_t)(off[ml + 2] - off[ml + 1]);
        // every (left, right) on one transcript with opposite strands, left-major in sorted order
        for (uint32_t i = 0; i < nl; ++i) {
            const uint32_t t = cand_tid(L[i]), f = cand_fwd(L[i]);
            for (uint32_t j = 0; j < nr; ++j) {
                if (cand_tid(R[j]) != t || cand_fwd(R[j]) == f) continue;
                if (FILL) {
                    const int32_t p = cand_pos(L[i]), p2 = cand_pos(R[j]);
                    const int32_t e1 = p + (int32_t)len1, e2 = p2 + (int32_t)len2;
                    const int32_t frag = (e1 > e2 ? e1 : e2) - (p < p2 ? p : p2);
                    out[n] = sfgpu_hit{t, p, p2, (uint32_t)frag, len1, len2, (uint8_t)f, (uint8_t)(1u - f), 3, 0};
                }
                ++n;
            }
        }
        if (n == 0) {                                       // orphans: the left run, then the right run
            if (FILL) {
                for (uint32_t i = 0; i < nl; ++i) out[i] = sfgpu_hit{cand_tid(L[i]), cand_pos(L[i]), 0, 0u, len1, len2, (uint8_t)cand_fwd(L[i]), 0, 1, 0};
                for (uint32_t j = 0; j < nr; ++j) out[nl + j] = sfgpu_hit{cand_tid(R[j]), cand_pos(R[j]), 0, 0u, len2, len1, (uint8_t)cand_fwd(R[j]), 0, 2, 0};
            }
            n = nl + nr;
        }
    }
    if (!FILL) rec_cnt[r] = n;
}
// lengths of the interleaved mates (left, right, left, ...) + the scan's zero sentinel
__global__ void k_mate_lens(uint64_t n_reads, const uint64_t* __restrict__ o1, const uint64_t* __restrict__ o2, uint32_t* lens) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) { lens[2 * r] = (uint32_t)(o1[r + 1] - o1[r]); lens[2 * r + 1] = (uint32_t)(o2[r + 1] - o2[r]); }
    else if (r == n_reads) lens[2 * r] = 0;
}
// side s of a pair: ranges / counts of read r -> mate 2 r + s
__global__ void k_interleave(uint64_t n_reads, int side, uint32_t per_mate, const uint64_t* __restrict__ s_ranges, const uint32_t* __restrict__ s_cnt,
                             uint64_t* ranges, uint32_t* cand_cnt) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t m = 2 * r + side;
    for (uint32_t i = 0; i < per_mate; ++i) ranges[(uint64_t)per_mate * m + i] = s_ranges[(uint64_t)per_mate * r + i];
    cand_cnt[m] = s_cnt[r];
}
__global__ void k_narrow_off(uint64_t n, const uint64_t* __restrict__ in, uint32_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[i];
}

}  // namespace sfgpu

using namespace sfgpu;

// waits for the stream when it goes out of scope
struct StreamSyncOnExit { hipStream_t s; ~StreamSyncOnExit() { (void)hipStreamSynchronize(s); } };

struct sfgpu_index {
    uint32_t k = 31, shift = 0, max_occ = 1000, n_seeds = 2;
    uint32_t seed_len = 0;                 // != 0: scan mode with seeds of this many bases (the default: min(19, k)); 0: end seeds
    uint64_t n_valid = 0, n_slots = 0, M = 0, n_buckets = 0;
    DevBuf<uint64_t> keys; DevBuf<uint32_t> tid, tpos, bucket;
    DevBuf<char> tseq; DevBuf<uint64_t> tseq_off; DevBuf<uint32_t> tlen;      // the transcripts' text (scan mode extends matches on it)
};
constexpr uint32_t kDefaultSeedLen = 19;

extern "C" {

int sfgpu_index_build(sfgpu_index** out, const char* d_seq, const uint64_t* d_seq_off, const uint32_t* d_ref_len, uint64_t M, uint32_t k,
                      uint32_t max_occ, sfgpu_stream stream) {
    SF_REQUIRE(out && d_seq && d_seq_off && d_ref_len && M > 0, SFGPU_ERR_INVALID, "sfgpu_index_build: null pointer");
    SF_REQUIRE(k >= 8 && k <= 31, SFGPU_ERR_INVALID, "sfgpu_index_build: 8 <= k <= 31");
    SF_REQUIRE(M < (1ull << 31), SFGPU_ERR_RANGE, "sfgpu_index_build: transcript ids must fit 31 bits");
    hipStream_t st = as_stream(stream);
    sfgpu_index* x = new sfgpu_index();
    x->k = k; x->M = M; x->max_occ = max_occ ? max_occ : 1000;
    int rc;
#define IDX_TRY(expr) do { if ((rc = (expr))) { delete x; return rc; } } while (0)
    DevBuf<uint32_t> n_kmers, vals_in, vals; DevBuf<uint64_t> kmer_off, keys_in; DevBuf<unsigned long long> ctr;
    StreamSyncOnExit sync_first{st};         // (destroyed before the scratch buffers above: see sfgpu_map_reads)
    IDX_TRY(n_kmers.reserve(M + 1, st, false)); IDX_TRY(kmer_off.reserve(M + 2, st, false)); IDX_TRY(ctr.reserve(1, st, false));
    hipLaunchKernelGGL(k_index_lens, dim3(mpgrid(M + 1)), dim3(kMapBlock), 0, st, M, d_ref_len, k, n_kmers.p);
    IDX_TRY(exclusive_scan_u32(n_kmers.p, kmer_off.p, M, st));
    uint64_t n = 0;
    if (hipMemcpyAsync(&n, kmer_off.p + M, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { delete x; set_error("sfgpu_index_build: copy failed"); return SFGPU_ERR_HIP; }
    if (n >= (1ull << 32)) { delete x; set_error("sfgpu_index_build: more than 2^32 k-mer positions"); return SFGPU_ERR_RANGE; }
    x->n_slots = n;
    const uint64_t nn = n ? n : 1;
    IDX_TRY(keys_in.reserve(nn, st, false)); IDX_TRY(vals_in.reserve(nn, st, false)); IDX_TRY(vals.reserve(nn, st, false));
    IDX_TRY(x->keys.reserve(nn + 1, st, false)); IDX_TRY(x->tid.reserve(nn, st, false)); IDX_TRY(x->tpos.reserve(nn, st, false));
    if (n) {
        const unsigned gy = M < 32768 ? (unsigned)M : 32768u, gz = (unsigned)((M + gy - 1) / gy);
        hipLaunchKernelGGL(k_index_kmers, dim3(4, gy, gz), dim3(kMapBlock), 0, st, d_seq, d_seq_off, d_ref_len, M, k, kmer_off.p, keys_in.p, vals_in.p);
        if (hipGetLastError() != hipSuccess) { delete x; set_error("sfgpu_index_build: launch failed"); return SFGPU_ERR_HIP; }
        // stable: occurrences of a k-mer stay in (transcript, position) order; invalid k-mers (key ~0) go last
        IDX_TRY(sort_pairs_u64_u32(keys_in.p, x->keys.p, vals_in.p, vals.p, n, st, 64));
        hipLaunchKernelGGL(k_index_locate, dim3(mpgrid(n)), dim3(kMapBlock), 0, st, n, M, vals.p, kmer_off.p, x->tid.p, x->tpos.p);
        (void)hipMemsetAsync(ctr.p, 0, 8, st);
        hipLaunchKernelGGL(k_count_valid, dim3(1024), dim3(kMapBlock), 0, st, n, x->keys.p, ctr.p);
        unsigned long long nv = 0;
        if (hipMemcpyAsync(&nv, ctr.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { delete x; set_error("sfgpu_index_build: copy failed"); return SFGPU_ERR_HIP; }
        x->n_valid = nv;
    }
    // the index's own copy of the text: the scan mode extends matches past the seed on it
    {
        (void)hipMemsetAsync(ctr.p, 0, 8, st);
        hipLaunchKernelGGL(k_seq_extent, dim3(mpgrid(M)), dim3(kMapBlock), 0, st, M, d_seq_off, d_ref_len, ctr.p);
        unsigned long long ext = 0;
        if (hipMemcpyAsync(&ext, ctr.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { delete x; set_error("sfgpu_index_build: copy failed"); return SFGPU_ERR_HIP; }
        IDX_TRY(x->tseq.reserve(ext + 1, st, false)); IDX_TRY(x->tseq_off.reserve(M, st, false)); IDX_TRY(x->tlen.reserve(M, st, false));
        if (hipMemcpyAsync(x->tseq.p, d_seq, ext, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(x->tseq_off.p, d_seq_off, M * 8, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(x->tlen.p, d_ref_len, M * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { delete x; set_error("sfgpu_index_build: copy failed"); return SFGPU_ERR_HIP; }
        x->seed_len = k < kDefaultSeedLen ? k : kDefaultSeedLen;
    }
    // bucket table over the top bits: ~8 k-mers per bucket, between 2^8 and 2^26 buckets
    uint32_t bits = 8; while (bits < 26 && bits < 2 * k && (1ull << bits) * 8 < x->n_valid) ++bits;
    x->shift = 2 * k - bits; x->n_buckets = 1ull << bits;
    IDX_TRY(x->bucket.reserve(x->n_buckets + 1, st, false));
    hipLaunchKernelGGL(k_index_buckets, dim3(mpgrid(x->n_buckets + 1)), dim3(kMapBlock), 0, st, x->n_valid, x->keys.p, x->shift, x->n_buckets, x->bucket.p);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { delete x; set_error("sfgpu_index_build: bucket table failed"); return SFGPU_ERR_HIP; }
#undef IDX_TRY
    log_msg(0, "index: %llu transcripts, %llu k-mer positions (%llu of A/C/G/T only), k = %u, %llu buckets", (unsigned long long)M,
            (unsigned long long)n, (unsigned long long)x->n_valid, k, (unsigned long long)x->n_buckets);
    *out = x;
    return SFGPU_OK;
}

int sfgpu_index_destroy(sfgpu_index* x) { delete x; return SFGPU_OK; }

int sfgpu_index_set_seeds(sfgpu_index* x, uint32_t seeds_per_strand) {
    SF_REQUIRE(x, SFGPU_ERR_INVALID, "sfgpu_index_set_seeds: null handle");
    SF_REQUIRE(seeds_per_strand >= 2 && seeds_per_strand <= kMaxSeeds, SFGPU_ERR_INVALID, "sfgpu_index_set_seeds: 2 .. 8 seeds per strand");
    x->n_seeds = seeds_per_strand;
    return SFGPU_OK;
}

int sfgpu_index_set_scan(sfgpu_index* x, uint32_t seed_len) {
    SF_REQUIRE(x, SFGPU_ERR_INVALID, "sfgpu_index_set_scan: null handle");
    SF_REQUIRE(seed_len == 0 || (seed_len >= 8 && seed_len <= x->k), SFGPU_ERR_INVALID, "sfgpu_index_set_scan: seed length 0 (end seeds) or 8 .. k");
    x->seed_len = seed_len;
    return SFGPU_OK;
}

int sfgpu_index_info(const sfgpu_index* x, uint32_t* k, uint64_t* n_positions, uint64_t* n_kmers) {
    SF_REQUIRE(x, SFGPU_ERR_INVALID, "sfgpu_index_info: null handle");
    if (k) *k = x->k;
    if (n_positions) *n_positions = x->n_slots;
    if (n_kmers) *n_kmers = x->n_valid;
    return SFGPU_OK;
}

int sfgpu_map_reads(const sfgpu_index* x, const char* d_seq1, const uint64_t* d_off1, const char* d_seq2, const uint64_t* d_off2,
                    uint32_t n_reads, sfgpu_hit* d_hits, uint64_t hit_capacity, uint32_t* d_hit_offsets, uint64_t* n_hits_out,
                    sfgpu_stream stream) {
    SF_REQUIRE(x && d_hit_offsets && n_hits_out, SFGPU_ERR_INVALID, "sfgpu_map_reads: null pointer");
    hipStream_t st = as_stream(stream);
    *n_hits_out = 0;
    if (n_reads == 0) { SF_HIP(hipMemsetAsync(d_hit_offsets, 0, 4, st)); SF_HIP(hipStreamSynchronize(st)); return SFGPU_OK; }
    SF_REQUIRE(d_seq1 && d_off1 && (!d_seq2 || d_off2), SFGPU_ERR_INVALID, "sfgpu_map_reads: null reads");
    const int paired = d_seq2 != nullptr;
    const uint64_t n_mates = paired ? 2ull * n_reads : n_reads;
    IndexView v{x->keys.p, x->tid.p, x->tpos.p, x->bucket.p, x->n_valid, x->k, x->shift, x->max_occ, x->n_seeds,
                x->seed_len, x->tseq.p, x->tseq_off.p, x->tlen.p};
    const bool scan = x->seed_len != 0;
    const uint64_t per_mate = scan ? 2ull * kScanGroups : 2ull * x->n_seeds;       // words per mate: group descriptors / lookups
    int rc;
    // Mates are numbered 2 r (left) and 2 r + 1 (right).  The two files are looked up side by side (each with its own
    // sequence buffer and offsets) and interleaved; from then on only the mates' LENGTHS are needed, as offsets `moff`.
    DevBuf<uint64_t> ranges, cand_off, cand, rec_off, moff, s_ranges; DevBuf<uint32_t> cand_cnt, n_hits, rec_cnt, s_cnt, lens; DevBuf<uint8_t> votes;
    StreamSyncOnExit sync_first{st};         // (declared after the buffers: destroyed before them -- an early error return must not hand
                                             //  scratch back to the pool while a kernel enqueued above may still be writing it)
    if ((rc = ranges.reserve(per_mate * n_mates, st, false)) || (rc = cand_cnt.reserve(n_mates + 1, st, false)) || (rc = cand_off.reserve(n_mates + 2, st, false)) ||
        (rc = n_hits.reserve(n_mates, st, false)) || (rc = rec_cnt.reserve((uint64_t)n_reads + 1, st, false)) ||
        (rc = rec_off.reserve((uint64_t)n_reads + 2, st, false)) || (rc = moff.reserve(n_mates + 2, st, false))) return rc;
    if (!paired) {
        if (scan) hipLaunchKernelGGL(k_scan_lookup, dim3(mpgrid((uint64_t)n_reads + 1)), dim3(kMapBlock), 0, st, v, d_seq1, d_off1, (uint64_t)n_reads, ranges.p, cand_cnt.p);
        else hipLaunchKernelGGL(k_map_lookup, dim3(mpgrid((uint64_t)n_reads + 1)), dim3(kMapBlock), 0, st, v, d_seq1, d_off1, (uint64_t)n_reads, ranges.p, cand_cnt.p);
        SF_CHECK_LAUNCH();
        SF_HIP(hipMemcpyAsync(moff.p, d_off1, ((uint64_t)n_reads + 1) * 8, hipMemcpyDeviceToDevice, st));
    } else {
        if ((rc = s_ranges.reserve(per_mate * n_reads, st, false)) || (rc = s_cnt.reserve((uint64_t)n_reads + 1, st, false)) || (rc = lens.reserve(n_mates + 1, st, false))) return rc;
        for (int side = 0; side < 2; ++side) {
            if (scan) hipLaunchKernelGGL(k_scan_lookup, dim3(mpgrid((uint64_t)n_reads + 1)), dim3(kMapBlock), 0, st, v, side ? d_seq2 : d_seq1, side ? d_off2 : d_off1,
                                         (uint64_t)n_reads, s_ranges.p, s_cnt.p);
            else hipLaunchKernelGGL(k_map_lookup, dim3(mpgrid((uint64_t)n_reads + 1)), dim3(kMapBlock), 0, st, v, side ? d_seq2 : d_seq1, side ? d_off2 : d_off1,
                               (uint64_t)n_reads, s_ranges.p, s_cnt.p);
            hipLaunchKernelGGL(k_interleave, dim3(mpgrid(n_reads)), dim3(kMapBlock), 0, st, (uint64_t)n_reads, side, (uint32_t)per_mate, s_ranges.p, s_cnt.p, ranges.p, cand_cnt.p);
            SF_CHECK_LAUNCH();
        }
        SF_HIP(hipMemsetAsync(cand_cnt.p + n_mates, 0, 4, st));
        hipLaunchKernelGGL(k_mate_lens, dim3(mpgrid((uint64_t)n_reads + 1)), dim3(kMapBlock), 0, st, (uint64_t)n_reads, d_off1, d_off2, lens.p);
        SF_CHECK_LAUNCH();
        if ((rc = exclusive_scan_u32(lens.p, moff.p, n_mates, st))) return rc;
    }
    if ((rc = exclusive_scan_u32(cand_cnt.p, cand_off.p, n_mates, st))) return rc;
    uint64_t n_cand = 0;
    SF_HIP(hipMemcpyAsync(&n_cand, cand_off.p + n_mates, 8, hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    if ((rc = cand.reserve(n_cand + 1, st, false)) || (rc = votes.reserve(n_cand + 1, st, false))) return rc;
    if (scan) hipLaunchKernelGGL(k_scan_hits, dim3(mpgrid(n_mates)), dim3(kMapBlock), 0, st, v, d_seq1, d_off1, d_seq2, d_off2, paired, n_mates, ranges.p, cand_off.p,
                                 cand.p, votes.p, n_hits.p);
    else hipLaunchKernelGGL(k_map_hits, dim3(mpgrid(n_mates)), dim3(kMapBlock), 0, st, v, moff.p, n_mates, ranges.p, cand_off.p, cand.p, votes.p, n_hits.p);
    SF_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_map_records<false>, dim3(mpgrid((uint64_t)n_reads + 1)), dim3(kMapBlock), 0, st, (uint64_t)n_reads, paired, moff.p, cand_off.p, cand.p,
                       n_hits.p, rec_cnt.p, (const uint64_t*)nullptr, (sfgpu_hit*)nullptr);
    SF_CHECK_LAUNCH();
    if ((rc = exclusive_scan_u32(rec_cnt.p, rec_off.p, n_reads, st))) return rc;
    uint64_t n_rec = 0;
    SF_HIP(hipMemcpyAsync(&n_rec, rec_off.p + n_reads, 8, hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    *n_hits_out = n_rec;
    SF_REQUIRE(n_rec < (1ull << 32), SFGPU_ERR_RANGE, "sfgpu_map_reads: more than 2^32 hit records in one batch");
    hipLaunchKernelGGL(k_narrow_off, dim3(mpgrid((uint64_t)n_reads + 1)), dim3(kMapBlock), 0, st, (uint64_t)n_reads + 1, rec_off.p, d_hit_offsets);
    SF_CHECK_LAUNCH();
    if (n_rec > hit_capacity || (n_rec && !d_hits)) {        // the caller sizes d_hits from *n_hits_out and calls again
        SF_HIP(hipStreamSynchronize(st));
        set_error("sfgpu_map_reads: %llu hit records, capacity %llu", (unsigned long long)n_rec, (unsigned long long)hit_capacity);
        return SFGPU_ERR_RANGE;
    }
    if (n_rec) {
        hipLaunchKernelGGL(k_map_records<true>, dim3(mpgrid((uint64_t)n_reads + 1)), dim3(kMapBlock), 0, st, (uint64_t)n_reads, paired, moff.p, cand_off.p, cand.p,
                           n_hits.p, (uint32_t*)nullptr, rec_off.p, d_hits);
        SF_CHECK_LAUNCH();
    }
    SF_HIP(hipStreamSynchronize(st));
    return SFGPU_OK;
}

}  // extern "C"
