// xxh64_device.h -- XXH64 (seed 0) of a uint32 label, as gfx950 device code.
//
// Arithmetic contract: src/xxhash.c:231-235 (primes), :346-455 (core) of the reference, as
// called by TranscriptGroup's constructor (src/TranscriptGroup.cpp:9-12: bytes = the uint32
// ids, little endian, length 4*n).  The input is always a whole number of 4-byte words, so
// the 1-byte tail loop (:441-446) cannot run and is not present here.
//
// The label is consumed as 32-bit words through a caller-supplied accessor so the same
// routine hashes labels that sit in global memory, LDS or registers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sfgpu {

__device__ __forceinline__ uint64_t xx_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

constexpr uint64_t XP1 = 11400714785074694791ULL;
constexpr uint64_t XP2 = 14029467366897019727ULL;
constexpr uint64_t XP3 = 1609587929392839161ULL;
constexpr uint64_t XP4 = 9650029242287828579ULL;
constexpr uint64_t XP5 = 2870177450012600261ULL;

__device__ __forceinline__ uint64_t xx_round(uint64_t acc, uint64_t in) {
    acc += in * XP2; acc = xx_rotl(acc, 31); return acc * XP1;
}
__device__ __forceinline__ uint64_t xx_merge(uint64_t h, uint64_t v) {
    v = xx_round(0, v); h ^= v; return h * XP1 + XP4;
}

// word(i) -> i-th uint32 of the label, n = number of words.
template <typename WordFn>
__device__ __forceinline__ uint64_t xxh64_words(WordFn word, uint32_t n) {
    uint32_t i = 0;
    uint64_t h;
    auto w64 = [&](uint32_t k) -> uint64_t { return (uint64_t)word(k) | ((uint64_t)word(k + 1) << 32); };
    if (n >= 8) {  // len >= 32 bytes: four-lane stripe loop (:361-415)
        uint64_t v1 = XP1 + XP2, v2 = XP2, v3 = 0, v4 = 0 - XP1;
        do {
            v1 = xx_round(v1, w64(i)); v2 = xx_round(v2, w64(i + 2));
            v3 = xx_round(v3, w64(i + 4)); v4 = xx_round(v4, w64(i + 6));
            i += 8;
        } while (i + 8 <= n);
        h = xx_rotl(v1, 1) + xx_rotl(v2, 7) + xx_rotl(v3, 12) + xx_rotl(v4, 18);
        h = xx_merge(h, v1); h = xx_merge(h, v2); h = xx_merge(h, v3); h = xx_merge(h, v4);
    } else {
        h = XP5;  // seed(0) + PRIME64_5 (:416-419)
    }
    h += (uint64_t)n * 4;  // :421
    while (i + 2 <= n) {   // 8-byte steps (:423-432)
        uint64_t k = xx_round(0, w64(i));
        h ^= k; h = xx_rotl(h, 27) * XP1 + XP4; i += 2;
    }
    if (i < n) {           // one 4-byte step (:434-439)
        h ^= (uint64_t)word(i) * XP1;
        h = xx_rotl(h, 23) * XP2 + XP3;
    }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;  // avalanche (:448-452)
    return h;
}


// ---- short labels from registers -------------------------------------------------------------
// Most labels have <= 8 ids.  Fetching their words with 8 independent (predicated) loads and hashing
// from registers removes the load -> multiply -> load dependency chain of the generic loop, which is
// what bounds a lane-per-label kernel (one memory round trip per 8 bytes otherwise).
constexpr int kHead = 8;

template <typename WordFn>
__device__ __forceinline__ void label_head(WordFn word, uint32_t n, uint32_t (&w)[kHead]) {
#pragma unroll
    for (int i = 0; i < kHead; ++i) w[i] = ((uint32_t)i < n) ? word(i) : 0u;
}

// XXH64 of a label of n <= 8 words held in w[] (same arithmetic as xxh64_words)
__device__ __forceinline__ uint64_t xxh64_head(const uint32_t (&w)[kHead], uint32_t n) {
    auto w64 = [&](int k) -> uint64_t { return (uint64_t)w[k] | ((uint64_t)w[k + 1] << 32); };
    uint64_t h;
    if (n == 8) {   // exactly one 32-byte stripe
        uint64_t v1 = xx_round(XP1 + XP2, w64(0)), v2 = xx_round(XP2, w64(2));
        uint64_t v3 = xx_round(0, w64(4)), v4 = xx_round(0 - XP1, w64(6));
        h = xx_rotl(v1, 1) + xx_rotl(v2, 7) + xx_rotl(v3, 12) + xx_rotl(v4, 18);
        h = xx_merge(h, v1); h = xx_merge(h, v2); h = xx_merge(h, v3); h = xx_merge(h, v4);
        h += 32;
    } else {
        h = XP5 + (uint64_t)n * 4;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if ((uint32_t)(2 * j + 2) <= n) { uint64_t k = xx_round(0, w64(2 * j)); h ^= k; h = xx_rotl(h, 27) * XP1 + XP4; }
        if (n & 1) {
            uint32_t last = (n == 1) ? w[0] : (n == 3) ? w[2] : (n == 5) ? w[4] : w[6];
            h ^= (uint64_t)last * XP1; h = xx_rotl(h, 23) * XP2 + XP3;
        }
    }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

// ---- bucket hash -----------------------------------------------------------------------------
// Where a label goes in the device table (region, slot, tag) is decided by a 64-bit mix built from
// two 32-bit lanes (a: multiply-rotate chain, one multiply per id; b: add-rotate chain fed by a), NOT by XXH64:
// a 64-bit integer multiply costs ~8 VALU multiplies on CDNA4 and hashing every read with XXH64
// measured ~0.5 ms per 16.7 M labels per pass.  Class identity never depends on this value (it is
// decided by a full label compare); XXH64 -- TranscriptGroup::hash -- is computed once per CLASS
// when the class is committed, and is what the export and the canonical order use.
constexpr uint32_t MP1 = 2654435761u, MP2 = 2246822519u, MP3 = 3266489917u;   // PRIME32_1..3 (src/xxhash.c:225-227)
__device__ __forceinline__ uint32_t mix_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t mix_fin(uint32_t h) {
    h ^= h >> 15; h *= MP2; h ^= h >> 13; h *= MP3; h ^= h >> 16; return h;
}
// one id into the two lanes: a single 32-bit multiply (quarter rate on CDNA4), the rest shifts and adds
__device__ __forceinline__ void mix_round(uint32_t& a, uint32_t& b, uint32_t w) {
    a = mix_rotl((a ^ w) * MP1, 13);
    b = mix_rotl(b, 11) + (w ^ a);
    b += b << 2;
}
// the head: 8 rounds over the first 8 ids, zero padded -- the length is in the seeds, so zero padding cannot alias a shorter
// label -- which keeps the code branch-free
__device__ __forceinline__ void label_mix_head(const uint32_t (&w)[kHead], uint32_t n, uint32_t& a, uint32_t& b) {
    a = MP1 + n; b = 0x27D4EB2Fu ^ (n * MP3);
#pragma unroll
    for (int i = 0; i < kHead; ++i) mix_round(a, b, w[i]);
}
// Labels of more than 8 ids: by default (mode 0) the bucket hash does NOT walk the tail.  It takes the length, the first 8 ids
// and three ids OF THE TAIL -- the last one, the one in the middle of the tail and the one a quarter into it (three more rounds,
// no loop): a label is an ordered id list, so labels that agree in all of these and still differ are rare, and when they do
// they only share a slot neighbourhood -- class identity is decided by the full label compare, never by this value.  (Round 3:
// walking the tail cost the route pass half of its vector instructions -- 13 % of the labels have one, so some lane of nearly
// every wavefront walked while the others waited: 9.6 instead of 6.4 ms per build in the ring form; hashing the tail granules by
// lanes of their own cost as much in bookkeeping: profiles/r3_class_build_notes.md.  Round 4: the middle sample was id[n / 2],
// which for n <= 15 lies in the head and added nothing.)
// Thousands of DISTINCT long labels that agree in every sampled field would share one home slot at every table size, fill their
// region and be deferred for ever.  The builder notices that (a grown table that defers as many reads as before) and switches
// to mode 1 for the rest of its life: the tail is walked, one round per id, as before round 3 -- the table is rehashed with it.
constexpr uint32_t kMixSampled = 0u, kMixFull = 1u;
__device__ __forceinline__ uint32_t mix_tail_mid(uint32_t n) { return (uint32_t)kHead + ((n - (uint32_t)kHead) >> 1); }
__device__ __forceinline__ uint32_t mix_tail_quarter(uint32_t n) { return (uint32_t)kHead + ((n - (uint32_t)kHead) >> 2); }
__device__ __forceinline__ void label_mix_far(uint32_t& a, uint32_t& b, uint32_t last, uint32_t middle, uint32_t quarter) {
    mix_round(a, b, last); mix_round(a, b, middle); mix_round(a, b, quarter);
}
__device__ __forceinline__ uint64_t label_mix_final(uint32_t a, uint32_t b) {
    return ((uint64_t)mix_fin(a ^ b) << 32) | mix_fin(b + (a >> 3));
}
// the tail of a label of n > 8 ids into the two lanes, by the builder's mode
template <typename WordFn>
__device__ __forceinline__ void label_mix_tail(uint32_t& a, uint32_t& b, WordFn word, uint32_t n, uint32_t mode) {
    if (mode == kMixSampled) label_mix_far(a, b, word(n - 1u), word(mix_tail_mid(n)), word(mix_tail_quarter(n)));
    else for (uint32_t k = kHead; k < n; ++k) mix_round(a, b, word(k));
}
// bucket hash of any label (w[] receives the zero-padded head)
template <typename WordFn>
__device__ __forceinline__ uint64_t label_mix64(WordFn word, uint32_t n, uint32_t (&w)[kHead], uint32_t mode) {
    label_head(word, n, w);
    uint32_t a, b;
    label_mix_head(w, n, a, b);
    if (n > (uint32_t)kHead) label_mix_tail(a, b, word, n, mode);
    return label_mix_final(a, b);
}

// hash of any label: registers for n <= 8, the generic loop otherwise
template <typename WordFn>
__device__ __forceinline__ uint64_t xxh64_label(WordFn word, uint32_t n, uint32_t (&w)[kHead]) {
    label_head(word, n, w);
    return (n <= (uint32_t)kHead) ? xxh64_head(w, n) : xxh64_words(word, n);
}

}  // namespace sfgpu
