// filter.hip -- per-read hit filtering on the device (sfgpu_filter_hits).
//
// Replaces the per-read bodies of processReadsQuasi, src/SailfishQuantify.cpp:215-417 (paired end) and
// :530-626 (single end), and sailfish::utils::compatibleHit / hitType, src/SailfishUtils.cpp:157-289.
//
// Shape: one lane per read.  What a read contributes is decided by two facts about its hit list -- how many
// hits are compatible with the expected library type, and how many hits there are -- because the reference's
// running "haveCompat" logic (:324-341) amounts to: the compatible hits in order if there is any, else every
// hit in order (nothing when enforceLibCompat).  So pass 1 counts and parks the kept ids at the read's input
// offset, an exclusive scan places the reads' output lists, pass 2 moves the ids there (4 bytes per id instead
// of a second read of the 24-byte records).  Orphan hit lists (left mate's run, then the
// right mate's) are visited in transcript order with a two-finger merge (:231-246) instead of being
// rearranged.  The fragment-length sample budget (:419-434) is "the first N qualifying reads": a scan of
// the qualifying flags gives every read its rank, so the histogram equals a single mapping thread's.
#include "common.h"
#include "primitives.h"

namespace sfgpu {

constexpr int kFilterBlock = 256;
enum : uint8_t { MS_SINGLE = 0, MS_LEFT = 1, MS_RIGHT = 2, MS_PAIRED = 3 };
enum : uint8_t { OR_SAME = 0, OR_AWAY = 1, OR_TOWARD = 2, OR_NONE = 3 };
enum : uint8_t { ST_SA = 0, ST_AS = 1, ST_S = 2, ST_A = 3, ST_U = 4 };

// compatibleHit(expected, start, isForward, ms) -- single-end reads and orphans (src/SailfishUtils.cpp:157-207)
__device__ __forceinline__ bool compat_single(sfgpu_libfmt e, bool fwd, uint8_t ms) {
    const uint8_t es = e.strandedness;
    const bool same = e.orientation == OR_SAME;
    switch (ms) {
        case MS_SINGLE: return fwd ? (es == ST_U || es == ST_S) : (es == ST_U || es == ST_A);
        case MS_LEFT:
            if (same) return es == ST_U || (es == ST_S && fwd) || (es == ST_A && !fwd);
            return fwd ? (es == ST_U || es == ST_S) : (es == ST_U || es == ST_A);
        case MS_RIGHT:
            if (same) return es == ST_U || (es == ST_S && fwd) || (es == ST_A && !fwd);
            return fwd ? (es == ST_U || es == ST_A) : (es == ST_U || es == ST_S);
        default: return false;      // "SHOULD NOT GET HERE"
    }
}

// hitType (:232-281) followed by compatibleHit(expected, observed) (:210-229) for a proper pair
__device__ __forceinline__ bool compat_pair(sfgpu_libfmt e, const sfgpu_hit& h, bool dovetail) {
    const bool f1 = h.fwd != 0, f2 = h.mate_fwd != 0;
    // :344-345: uint32 arithmetic, then passed as int32
    const int32_t s1 = (int32_t)(f1 ? (uint32_t)h.pos : (uint32_t)h.pos + (uint32_t)h.read_len);
    const int32_t s2 = (int32_t)(f2 ? (uint32_t)h.mate_pos : (uint32_t)h.mate_pos + (uint32_t)h.mate_len);
    uint8_t oo, os;
    if (f1 != f2) {
        if (f1) { const int32_t stretch = dovetail ? (int32_t)h.mate_len : 0; oo = (s1 <= s2 + stretch) ? OR_TOWARD : OR_AWAY; os = ST_SA; }
        else    { const int32_t stretch = dovetail ? (int32_t)h.read_len : 0; oo = (s2 <= s1 + stretch) ? OR_TOWARD : OR_AWAY; os = ST_AS; }
    } else { oo = OR_SAME; os = f1 ? ST_S : ST_A; }
    if (e.orientation != oo) return false;
    return e.strandedness == ST_U || e.strandedness == os;
}

struct ReadView {
    const sfgpu_hit* h; uint32_t n_raw, n;     // n: after the maxReadOccs / orphan cuts
    bool paired_hits;                          // jointHits.front().mateStatus == PAIRED_END_PAIRED
    uint32_t n_left;                           // orphans of a paired library: length of the left mate's run
};

// `hits` holds the records from global hit number `first` on: the global array (first = 0) or the block's LDS copy
__device__ __forceinline__ ReadView view_read(const sfgpu_hit* hits, uint32_t first, const uint32_t* off, uint32_t r,
                                              const sfgpu_filter_opts& o) {
    ReadView v;
    const uint32_t b = off[r];
    v.h = hits + (b - first); v.n_raw = off[r + 1] - b; v.n = v.n_raw; v.paired_hits = false; v.n_left = 0;
    if (v.n > o.max_read_occs) v.n = 0;                                   // :217 / :532
    if (v.n && o.paired_library) {
        v.paired_hits = v.h[0].mate_status == MS_PAIRED;                  // :221
        if (o.discard_orphans && !v.paired_hits) v.n = 0;                 // :226
        if (v.n && !v.paired_hits) while (v.n_left < v.n && v.h[v.n_left].mate_status == MS_LEFT) ++v.n_left;   // :233-237
    }
    return v;
}

// visit the read's hits in the order the reference's loop sees them; f(hit, compatible)
template <typename F>
__device__ __forceinline__ void for_each_hit(const ReadView& v, const sfgpu_filter_opts& o, F f) {
    const bool dovetail = o.can_dovetail != 0, ignore = o.ignore_compat != 0;
    if (o.paired_library && v.paired_hits) {
        for (uint32_t i = 0; i < v.n; ++i) f(v.h[i], ignore || compat_pair(o.expected, v.h[i], dovetail));
    } else if (o.paired_library) {
        uint32_t i = 0, j = v.n_left;                                     // std::inplace_merge by transcriptID (:241-245): stable
        while (i < v.n_left || j < v.n) {
            const bool take_left = (j >= v.n) || (i < v.n_left && !(v.h[j].tid < v.h[i].tid));
            const sfgpu_hit& h = take_left ? v.h[i++] : v.h[j++];
            f(h, ignore || compat_single(o.expected, h.fwd != 0, h.mate_status));
        }
    } else {
        for (uint32_t i = 0; i < v.n; ++i) f(v.h[i], ignore || compat_single(o.expected, v.h[i].fwd != 0, v.h[i].mate_status));
    }
}

// "forward" for the strand counters: :313-320 for orphans, h.fwd otherwise (:328, :353, :595)
__device__ __forceinline__ bool counts_as_fwd(const sfgpu_hit& h, bool orphan_of_pair) {
    if (!orphan_of_pair) return h.fwd != 0;
    return (h.mate_status == MS_LEFT && h.fwd) || (h.mate_status == MS_RIGHT && !h.fwd);
}

struct FilterCounters { unsigned long long mapped, total_hits, upper, fwd, rc; };
constexpr int kCtrCopies = 64;      // block totals go to copy blockIdx % 64: same-address atomics serialise (~6 ns each)

// A block's reads own one contiguous range of hit records: it is copied to LDS with coalesced 8-byte loads and
// the lanes then walk their hits there.  A lane walking 24-byte records in global memory makes every load
// instruction touch 64 different cache lines (the same address-rate limit as in the class builder's passes).
constexpr uint32_t kStageHits = 1536;                        // 36 KB: four 256-thread blocks per CU
__device__ __forceinline__ const sfgpu_hit* stage_block_hits(const sfgpu_hit* __restrict__ hits, const uint32_t* __restrict__ off,
                                                             uint32_t n_reads, sfgpu_hit* lds, uint32_t& first) {
    first = 0;
    const uint32_t r0 = blockIdx.x * kFilterBlock;
    const uint32_t r1 = (r0 + kFilterBlock < n_reads) ? r0 + kFilterBlock : n_reads;
    if (r0 >= n_reads) return hits;
    const uint32_t h_lo = off[r0], h_hi = off[r1];
    if (h_hi - h_lo > kStageHits) return hits;               // uniform: a block with unusually long hit lists reads global memory
    const uint2* src = reinterpret_cast<const uint2*>(hits + h_lo);          // records are 24 bytes: 8-byte aligned
    uint2* dst = reinterpret_cast<uint2*>(lds);
    const uint32_t n8 = (h_hi - h_lo) * 3;
    for (uint32_t i = threadIdx.x; i < n8; i += kFilterBlock) dst[i] = src[i];
    __syncthreads();
    first = h_lo;
    return lds;
}

__global__ void __launch_bounds__(kFilterBlock)
k_filter_count(const sfgpu_hit* __restrict__ hits, const uint32_t* __restrict__ off, uint32_t n_reads, sfgpu_filter_opts o,
               uint32_t* __restrict__ out_len, uint32_t* __restrict__ kept, uint32_t* __restrict__ fl_flag,
               uint32_t* __restrict__ fl_len, FilterCounters* ctr) {
    const uint32_t r = blockIdx.x * kFilterBlock + threadIdx.x;
    unsigned long long mapped = 0, total = 0, upper = 0, nf = 0, nr = 0;
    __shared__ __attribute__((aligned(8))) sfgpu_hit lds_hits[kStageHits];
    uint32_t first_hit;
    const sfgpu_hit* my_hits = stage_block_hits(hits, off, n_reads, lds_hits, first_hit);
    if (r < n_reads) {
        const ReadView v = view_read(my_hits, first_hit, off, r, o);
        uint32_t n_compat = 0, fw_c = 0, fw_a = 0, n_seen = 0;
        const bool orphan = o.paired_library && !v.paired_hits;
        // The kept ids go to a scratch list at the read's INPUT offset (an upper bound of its output size):
        // compatible hits are packed at the front as they are found, and while none has been found every hit
        // is appended too -- the reference's txpIDsCompat / txpIDsAll (:324-341) sharing one buffer, since only
        // one of the two survives.  Pass 2 then only moves 4-byte ids instead of re-reading 24-byte records.
        uint32_t* mine = kept + off[r];
        for_each_hit(v, o, [&](const sfgpu_hit& h, bool compat) {
            const bool fw = counts_as_fwd(h, orphan);
            if (compat) { mine[n_compat++] = h.tid; fw_c += fw; }
            else if (n_compat == 0 && !o.enforce_compat) mine[n_seen] = h.tid;
            ++n_seen; fw_a += fw;
        });
        const uint32_t len = n_compat ? n_compat : (o.enforce_compat ? 0u : v.n);
        out_len[r] = len;
        upper = v.n_raw > 0; total = v.n; mapped = len > 0;
        if (len) { const uint32_t f = n_compat ? fw_c : fw_a; nf = f; nr = len - f; }
        // :419-434: a unique, properly paired, mapped fragment shorter than maxFragLen is a length sample
        uint32_t flag = 0;
        if (fl_flag) {
            if (o.paired_library && v.n == 1 && v.paired_hits && len > 0 && v.h[0].frag_len < o.max_frag_len) { flag = 1; fl_len[r] = v.h[0].frag_len; }
            fl_flag[r] = flag;
        }
    } else if (r == n_reads) { out_len[r] = 0; if (fl_flag) fl_flag[r] = 0; }     // scan sentinels
    // block totals -> five atomics per block
    __shared__ unsigned long long red[5][kFilterBlock / kWave];
    unsigned long long v5[5] = {mapped, total, upper, nf, nr};
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        unsigned long long x = v5[q];
        for (int s = kWave / 2; s > 0; s >>= 1) x += __shfl_down(x, s, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) red[q][threadIdx.x / kWave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        unsigned long long t = 0;
        for (int w = 0; w < kFilterBlock / kWave; ++w) t += red[threadIdx.x][w];
        if (t) atomicAdd(reinterpret_cast<unsigned long long*>(ctr + (blockIdx.x % kCtrCopies)) + threadIdx.x, t);
    }
}

// pass 2: move every read's kept ids from its input offset to its output offset; fragment-length samples
__global__ void __launch_bounds__(kFilterBlock)
k_filter_compact(const uint32_t* __restrict__ off, uint32_t n_reads, const uint32_t* __restrict__ kept,
                 const uint64_t* __restrict__ out_off64, uint32_t* __restrict__ ids_out, uint32_t* __restrict__ off_out,
                 const uint32_t* __restrict__ fl_flag, const uint32_t* __restrict__ fl_len, const uint64_t* __restrict__ fl_rank,
                 uint64_t fl_budget, uint32_t* fl_counts) {
    const uint32_t r = blockIdx.x * kFilterBlock + threadIdx.x;
    if (r > n_reads) return;
    const uint64_t o0 = out_off64[r];
    off_out[r] = (uint32_t)o0;
    if (r == n_reads) return;
    const uint32_t len = (uint32_t)(out_off64[r + 1] - o0);
    const uint32_t* src = kept + off[r];
    uint32_t* dst = ids_out + o0;
    for (uint32_t k = 0; k < len; ++k) dst[k] = src[k];
    if (fl_flag && fl_flag[r] && fl_rank[r] < fl_budget) atomicAdd(&fl_counts[fl_len[r]], 1u);
}

// ---- bias / GC samples of the same loop (sfgpu_sample_bias) ---------------------------------------------
struct SamplerDev {
    const char* seq; const uint64_t* seq_off; const uint32_t* ref_len; const uint32_t* gc_prefix;
    uint32_t* observed_gc; int want_seq; uint32_t gc_step;
};

// Transcript::gcCountInterp_ (include/Transcript.hpp:133-162) on the per-base table: see bias.hip (gc_count_interp)
__device__ __forceinline__ double sampled_gc_count(const uint32_t* __restrict__ G, uint32_t L, uint32_t step, uint32_t p) {
    if (p == L - 1) return (double)G[L - 1];
    const double frac_p = (double)p / (double)step;
    const uint32_t samp = (uint32_t)floor(frac_p);
    const double lambda = (frac_p - (double)samp) / ((double)(samp + 1) - (double)samp);
    const uint32_t nxt = (samp + 1) * step < L - 1 ? (samp + 1) * step : L - 1;
    return lambda * (double)G[samp * step] + (1.0 - lambda) * (double)G[nxt];
}

// indexForKmer (include/UtilityFunctions.hpp:93-148): false when a byte is not ACGTU (the reference then gets
// 0xFFFFFFFF and ReadKmerDist::update reports no success)
__device__ __forceinline__ bool kmer_index6(const char* p, bool revcomp, uint32_t& idx) {
    uint32_t f = 0, r = 0; bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const unsigned c = (unsigned char)p[j] & 0xDFu;
        uint32_t code = 0;
        if (c == 'A') code = 0; else if (c == 'C') code = 1; else if (c == 'G') code = 2; else if (c == 'T' || c == 'U') code = 3; else ok = false;
        f = (f << 2) | code;
        r |= (3u - code) << (2 * j);
    }
    idx = revcomp ? r : f;
    return ok;
}

constexpr int kGcBins = 101;

// One lane per read, hits staged as in k_filter_count.  The read-start context is only flagged here (the budget
// is "the first N successes in read order": ranks come from a scan, k_apply_bias_samples adds the survivors);
// GC samples have no budget and go to a block-private histogram in LDS.
__global__ void __launch_bounds__(kFilterBlock)
k_sample_bias(const sfgpu_hit* __restrict__ hits, const uint32_t* __restrict__ off, uint32_t n_reads, sfgpu_filter_opts o,
              SamplerDev s, uint32_t* __restrict__ flag, uint32_t* __restrict__ kmer, unsigned long long* n_gc) {
    const uint32_t r = blockIdx.x * kFilterBlock + threadIdx.x;
    __shared__ __attribute__((aligned(8))) sfgpu_hit lds_hits[kStageHits];
    __shared__ uint32_t gc_hist[kGcBins];
    if (threadIdx.x < kGcBins) gc_hist[threadIdx.x] = 0;                   // stage_block_hits synchronises (or returns uniformly)
    __syncthreads();
    uint32_t first_hit;
    const sfgpu_hit* my_hits = stage_block_hits(hits, off, n_reads, lds_hits, first_hit);
    if (r < n_reads) {
        const ReadView v = view_read(my_hits, first_hit, off, r, o);
        bool need = s.want_seq != 0;                                         // needBiasSample (:255 / :545)
        uint32_t got = 0, my_idx = 0;
        for_each_hit(v, o, [&](const sfgpu_hit& h, bool) {
            const uint32_t L = s.ref_len[h.tid];
            if (need) {                                                      // :270-287 / :559-581
                const int32_t start_pos = h.fwd ? h.pos : h.pos + (int32_t)h.read_len;
                if (start_pos > 0 && (uint32_t)start_pos < L) {
                    const char* txp = s.seq + s.seq_off[h.tid];
                    // ReadKmerDist::update: 2 bases before a forward read's start / 4 before a reverse read's, K = 6
                    const int32_t back = h.fwd ? 2 : 4;
                    if (start_pos >= back && (uint32_t)(start_pos - back + 6) < L) {
                        uint32_t idx;
                        if (kmer_index6(txp + (start_pos - back), h.fwd != 0, idx)) { got = 1; my_idx = idx; need = false; }
                    }
                }
            }
            if (s.observed_gc && o.paired_library && h.mate_status == MS_PAIRED) {   // :375-389
                const int32_t start = h.pos < h.mate_pos ? h.pos : h.mate_pos;
                const int32_t stop = (int32_t)((uint32_t)start + h.frag_len);
                if (start > 0 && (uint32_t)stop < L) {
                    const uint32_t* G = s.gc_prefix + s.seq_off[h.tid];
                    long g;
                    if (s.gc_step <= 1) {
                        const uint32_t d = G[stop] - G[start];
                        g = lrint((100.0 * (double)d) / (double)(stop - start + 1));           // Transcript::gcFrac
                    } else {
                        const double cs = sampled_gc_count(G, L, s.gc_step, (uint32_t)start), ce = sampled_gc_count(G, L, s.gc_step, (uint32_t)stop);
                        g = lrint((100.0 * (ce - cs)) / (double)(stop - start + 1));
                        g = g < 0 ? 0 : (g > 100 ? 100 : g);                                   // the reference indexes out of range here
                    }
                    atomicAdd(&gc_hist[g], 1u);
                }
            }
        });
        if (flag) { flag[r] = got; kmer[r] = my_idx; }
    } else if (r == n_reads && flag) flag[r] = 0;
    __syncthreads();
    if (threadIdx.x < 2 * kWave && s.observed_gc) {                          // 101 bins: the first two wavefronts
        const uint32_t c = threadIdx.x < kGcBins ? gc_hist[threadIdx.x] : 0u;
        if (c) atomicAdd(&s.observed_gc[threadIdx.x], c);
        unsigned long long t = c;
        for (int o = kWave / 2; o > 0; o >>= 1) t += __shfl_down(t, o, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0 && t) atomicAdd(n_gc, t);
    }
}

__global__ void __launch_bounds__(kFilterBlock)
k_apply_bias_samples(uint32_t n_reads, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ kmer,
                     const uint64_t* __restrict__ rank, uint64_t budget, uint32_t* read_bias) {
    const uint32_t r = blockIdx.x * kFilterBlock + threadIdx.x;
    if (r < n_reads && flag[r] && rank[r] < budget) atomicAdd(&read_bias[kmer[r]], 1u);
}

}  // namespace sfgpu

using namespace sfgpu;

extern "C" int sfgpu_sample_bias(const sfgpu_hit* d_hits, const uint32_t* d_hit_offsets, uint32_t n_reads,
                                 const sfgpu_filter_opts* opts, sfgpu_bias_sampler* sp, sfgpu_stream stream) {
    SF_REQUIRE(d_hit_offsets && opts && sp, SFGPU_ERR_INVALID, "sfgpu_sample_bias: null pointer");
    SF_REQUIRE(n_reads < 0x7FFFFFFFu, SFGPU_ERR_RANGE, "sfgpu_sample_bias: a batch holds < 2^31 reads");
    const bool want_seq = sp->d_read_bias && sp->remaining_bias_samples && *sp->remaining_bias_samples > 0;
    const bool want_gc = sp->d_observed_gc != nullptr && opts->paired_library;
    if (n_reads == 0 || (!want_seq && !want_gc)) return SFGPU_OK;
    SF_REQUIRE(sp->d_seq && sp->d_seq_off && sp->d_ref_len, SFGPU_ERR_INVALID, "sfgpu_sample_bias: null pointer");   // d_hits may be NULL when no read has a hit
    SF_REQUIRE(!want_gc || sp->d_gc_prefix, SFGPU_ERR_INVALID, "sfgpu_sample_bias: GC sampling needs d_gc_prefix (sfgpu_gc_prefix)");
    hipStream_t st = as_stream(stream);
    const size_t n1 = (size_t)n_reads + 1;
    uint32_t *d_flag = nullptr, *d_kmer = nullptr; uint64_t* d_rank = nullptr; unsigned long long* d_ngc = nullptr;
    int rc = SFGPU_OK;
    hipError_t e = pool_malloc(&d_ngc, 8);
    if (e == hipSuccess && want_seq) e = pool_malloc(&d_flag, n1 * 4);
    if (e == hipSuccess && want_seq) e = pool_malloc(&d_kmer, n1 * 4);
    if (e == hipSuccess && want_seq) e = pool_malloc(&d_rank, (n1 + 1) * 8);
    if (e == hipSuccess) e = hipMemsetAsync(d_ngc, 0, 8, st);
    const unsigned grid = (unsigned)((n1 + kFilterBlock - 1) / kFilterBlock);
    SamplerDev sd{sp->d_seq, sp->d_seq_off, sp->d_ref_len, sp->d_gc_prefix, want_gc ? sp->d_observed_gc : nullptr, want_seq ? 1 : 0, sp->gc_size_samp};
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sample_bias, dim3(grid), dim3(kFilterBlock), 0, st, d_hits, d_hit_offsets, n_reads, *opts, sd, d_flag, d_kmer, d_ngc);
        e = hipGetLastError();
    }
    uint64_t h_succ = 0; unsigned long long h_ngc = 0;
    if (e == hipSuccess && want_seq) rc = exclusive_scan_u32(d_flag, d_rank, n_reads, st);
    if (e == hipSuccess && rc == SFGPU_OK) {
        if (want_seq) e = hipMemcpyAsync(&h_succ, d_rank + n_reads, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&h_ngc, d_ngc, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    const uint64_t budget = want_seq ? (uint64_t)*sp->remaining_bias_samples : 0;
    if (e == hipSuccess && rc == SFGPU_OK && want_seq) {
        hipLaunchKernelGGL(k_apply_bias_samples, dim3(grid), dim3(kFilterBlock), 0, st, n_reads, d_flag, d_kmer, d_rank, budget, sp->d_read_bias);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    for (void* p : {(void*)d_flag, (void*)d_kmer, (void*)d_rank, (void*)d_ngc}) if (p) pool_free(p);
    SF_HIP(e);
    if (rc) return rc;
    if (want_seq) {
        const uint64_t taken = h_succ < budget ? h_succ : budget;
        *sp->remaining_bias_samples -= (int64_t)taken;
        sp->n_bias_sampled += taken;
    }
    sp->n_gc_sampled += h_ngc;
    return SFGPU_OK;
}

extern "C" int sfgpu_filter_hits(const sfgpu_hit* d_hits, const uint32_t* d_hit_offsets, uint32_t n_reads,
                                 const sfgpu_filter_opts* opts, uint32_t* d_ids_out, uint32_t* d_offsets_out,
                                 uint32_t* d_fl_counts, int64_t* remaining_fl_ops, sfgpu_filter_stats* stats,
                                 sfgpu_stream stream) {
    SF_REQUIRE(d_hit_offsets && opts && d_offsets_out, SFGPU_ERR_INVALID, "sfgpu_filter_hits: null pointer");
    SF_REQUIRE(n_reads < 0x7FFFFFFFu, SFGPU_ERR_RANGE, "sfgpu_filter_hits: a batch holds < 2^31 reads");
    hipStream_t st = as_stream(stream);
    if (n_reads == 0) { SF_HIP(hipMemsetAsync(d_offsets_out, 0, 4, st)); SF_HIP(hipStreamSynchronize(st)); return SFGPU_OK; }

    const bool want_fl = d_fl_counts && remaining_fl_ops && *remaining_fl_ops > 0 && opts->paired_library && opts->max_frag_len > 0;
    uint32_t *d_len = nullptr, *d_flag = nullptr, *d_kept = nullptr, *d_fllen = nullptr;
    uint64_t *d_off64 = nullptr, *d_rank = nullptr; FilterCounters* d_ctr = nullptr;
    uint32_t n_hits = 0;
    SF_HIP(hipMemcpyAsync(&n_hits, d_hit_offsets + n_reads, 4, hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
    SF_REQUIRE(n_hits == 0 || (d_hits && d_ids_out), SFGPU_ERR_INVALID, "sfgpu_filter_hits: null pointer");
    const size_t n1 = (size_t)n_reads + 1;
    int rc = SFGPU_OK;
    hipError_t e = pool_malloc(&d_len, n1 * 4);
    if (e == hipSuccess) e = pool_malloc(&d_kept, ((size_t)n_hits + 1) * 4);
    if (e == hipSuccess) e = pool_malloc(&d_off64, (n1 + 1) * 8);
    if (e == hipSuccess) e = pool_malloc(&d_ctr, kCtrCopies * sizeof(FilterCounters));
    if (e == hipSuccess && want_fl) e = pool_malloc(&d_flag, n1 * 4);
    if (e == hipSuccess && want_fl) e = pool_malloc(&d_fllen, n1 * 4);
    if (e == hipSuccess && want_fl) e = pool_malloc(&d_rank, (n1 + 1) * 8);
    if (e == hipSuccess) e = hipMemsetAsync(d_ctr, 0, kCtrCopies * sizeof(FilterCounters), st);
    const unsigned grid = (unsigned)((n1 + kFilterBlock - 1) / kFilterBlock);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_filter_count, dim3(grid), dim3(kFilterBlock), 0, st, d_hits, d_hit_offsets, n_reads, *opts, d_len, d_kept, d_flag, d_fllen, d_ctr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) rc = exclusive_scan_u32(d_len, d_off64, n_reads, st);
    if (e == hipSuccess && rc == SFGPU_OK && want_fl) rc = exclusive_scan_u32(d_flag, d_rank, n_reads, st);
    uint64_t h_tot[2] = {0, 0};
    FilterCounters h_ctr{};
    FilterCounters h_copies[kCtrCopies];
    if (e == hipSuccess && rc == SFGPU_OK) {
        e = hipMemcpyAsync(&h_tot[0], d_off64 + n_reads, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && want_fl) e = hipMemcpyAsync(&h_tot[1], d_rank + n_reads, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_copies, d_ctr, sizeof(h_copies), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) for (const FilterCounters& c : h_copies) {
            h_ctr.mapped += c.mapped; h_ctr.total_hits += c.total_hits; h_ctr.upper += c.upper; h_ctr.fwd += c.fwd; h_ctr.rc += c.rc;
        }
    }
    if (e == hipSuccess && rc == SFGPU_OK && h_tot[0] >= (1ull << 32)) { set_error("sfgpu_filter_hits: the batch's output exceeds 2^32 ids"); rc = SFGPU_ERR_RANGE; }
    const uint64_t budget = want_fl ? (uint64_t)*remaining_fl_ops : 0;
    if (e == hipSuccess && rc == SFGPU_OK) {
        hipLaunchKernelGGL(k_filter_compact, dim3(grid), dim3(kFilterBlock), 0, st, d_hit_offsets, n_reads, d_kept, d_off64,
                           d_ids_out, d_offsets_out, d_flag, d_fllen, d_rank, budget, d_fl_counts);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    for (void* p : {(void*)d_len, (void*)d_off64, (void*)d_ctr, (void*)d_flag, (void*)d_rank, (void*)d_kept, (void*)d_fllen}) if (p) pool_free(p);
    SF_HIP(e);
    if (rc) return rc;
    const uint64_t sampled = want_fl ? (h_tot[1] < budget ? h_tot[1] : budget) : 0;
    if (want_fl) *remaining_fl_ops -= (int64_t)sampled;
    if (stats) {
        stats->n_observed += n_reads; stats->n_mapped += h_ctr.mapped; stats->total_hits += h_ctr.total_hits;
        stats->upper_bound_hits += h_ctr.upper; stats->n_fwd += h_ctr.fwd; stats->n_rc += h_ctr.rc; stats->fl_sampled += sampled;
    }
    return SFGPU_OK;
}
