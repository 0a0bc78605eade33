// colour.h -- host-side first-fit colouring of classes that may share transcripts (used by the Gibbs plan, gibbs.hip).
// Plain C++: the CPU test suite compiles it on its own (tests/colour_harness.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace sfgpu {

// First-fit colouring of the wide classes, in class order: colour(c) = the smallest colour that no earlier class sharing a
// transcript with c has.  Classes of one colour share no transcript, so a colour is one launch of independent blocks, and
// two classes that do share one are in different launches: visiting the colours in order is a sequential scan of the wide
// classes in SOME order -- all a systematic scan needs (the reference's own order is its hash table's).
// Per transcript: a 64-bit mask of the colours 0..63 its classes hold, overflow
// words for the colours beyond (allocated for the few transcripts that need them), and the lowest colour that may still be
// free -- a transcript shared by thousands of classes hands out its colours in O(1) each.
struct ColourState {
    std::vector<uint64_t> small;                 // colours 0..63 per transcript
    std::vector<uint32_t> hint;                  // all colours < hint[t] are taken at t
    std::vector<int32_t> ovf_at;                 // index into ovf, or -1
    std::vector<std::vector<uint64_t>> ovf;      // colours 64.. per transcript that needs them
    explicit ColourState(uint64_t M) : small(M, 0), hint(M, 0), ovf_at(M, -1) {}
    uint64_t word(uint32_t t, uint32_t w) const {
        if (w == 0) return small[t];
        const int32_t o = ovf_at[t];
        if (o < 0 || w - 1 >= ovf[(size_t)o].size()) return 0;
        return ovf[(size_t)o][w - 1];
    }
    void set(uint32_t t, uint32_t colour) {
        const uint32_t w = colour >> 6; const uint64_t bit = 1ull << (colour & 63);
        if (w == 0) small[t] |= bit;
        else {
            if (ovf_at[t] < 0) { ovf_at[t] = (int32_t)ovf.size(); ovf.emplace_back(); }
            auto& v = ovf[(size_t)ovf_at[t]];
            if (v.size() < w) v.resize(w, 0);
            v[w - 1] |= bit;
        }
        uint32_t h = hint[t];
        while ((word(t, h >> 6) >> (h & 63)) & 1ull) ++h;
        hint[t] = h;
    }
};
inline uint32_t colour_wide_classes(const std::vector<uint32_t>& wl, const std::vector<uint32_t>& rowptr, const std::vector<uint32_t>& ids,
                                    uint64_t M, std::vector<uint32_t>& colour_of) {
    ColourState cs(M);
    colour_of.resize(wl.size());
    uint32_t n_colours = 0;
    for (size_t i = 0; i < wl.size(); ++i) {
        const uint32_t b = rowptr[wl[i]], e = rowptr[wl[i] + 1];
        uint32_t start = 0;
        for (uint32_t j = b; j < e; ++j) start = std::max(start, cs.hint[ids[j]]);
        uint32_t colour = start;
        for (uint32_t w = start >> 6;; ++w) {
            uint64_t used = 0;
            for (uint32_t j = b; j < e; ++j) used |= cs.word(ids[j], w);
            if (w == (start >> 6)) used |= (1ull << (start & 63)) - 1ull;          // colours below `start` are taken somewhere
            if (~used) { colour = (w << 6) + (uint32_t)__builtin_ctzll(~used); break; }
        }
        colour_of[i] = colour;
        for (uint32_t j = b; j < e; ++j) cs.set(ids[j], colour);
        if (colour + 1 > n_colours) n_colours = colour + 1;
    }
    return n_colours;
}

// Connected components of the wide classes (two classes are connected when they share a transcript, directly or through
// other wide classes): comp_of[i] = component of wl[i], numbered in order of first appearance.  Classes of different
// components commute, so components can be visited concurrently whatever happens inside them.
inline uint32_t components_of_wide_classes(const std::vector<uint32_t>& wl, const std::vector<uint32_t>& rowptr, const std::vector<uint32_t>& ids,
                                           uint64_t M, std::vector<uint32_t>& comp_of) {
    std::vector<uint32_t> parent(M);
    for (uint64_t t = 0; t < M; ++t) parent[t] = (uint32_t)t;
    auto find = [&](uint32_t t) { while (parent[t] != t) { parent[t] = parent[parent[t]]; t = parent[t]; } return t; };
    for (size_t i = 0; i < wl.size(); ++i) {
        const uint32_t b = rowptr[wl[i]], e = rowptr[wl[i] + 1];
        if (e == b) continue;
        const uint32_t r0 = find(ids[b]);
        for (uint32_t j = b + 1; j < e; ++j) { const uint32_t r = find(ids[j]); if (r != r0) parent[r] = r0; }
    }
    std::vector<uint32_t> label(M, 0xFFFFFFFFu);
    comp_of.resize(wl.size());
    uint32_t n = 0;
    for (size_t i = 0; i < wl.size(); ++i) {
        const uint32_t b = rowptr[wl[i]], e = rowptr[wl[i] + 1];
        if (e == b) { comp_of[i] = n++; continue; }
        const uint32_t r = find(ids[b]);
        if (label[r] == 0xFFFFFFFFu) label[r] = n++;
        comp_of[i] = label[r];
    }
    return n;
}

}  // namespace sfgpu
