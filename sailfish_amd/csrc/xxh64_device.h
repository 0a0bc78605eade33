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

}  // namespace sfgpu
