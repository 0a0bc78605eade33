// ref_glue.cpp -- TEST INFRASTRUCTURE.  A thin extern "C" driver around units of the REFERENCE that compile
// here unmodified with nothing but the standard library (g++ -std=c++11 -I/root/reference/include):
//   include/MultinomialSampler.hpp   (std only)
//   include/LibraryFormat.hpp + src/LibraryFormat.cpp
//   include/cuckoohash_map.hh (+ cuckoohash_config.hh, cuckoohash_util.hh: vendored libcuckoo, std only)
//   src/xxhash.c
// The reference sources are compiled FROM WHERE THEY LIE (oracle/Makefile, target `ref`); nothing of them is copied
// into this repository and no stand-in header is involved.  The output, oracle/_ref/libsailfish_ref.so, is used by
// tests/ to pin the repo's own restatement (oracle/sf_oracle.c) and the HIP path -- never by the product.
//
// What is NOT reachable this way: TranscriptGroup / EquivalenceClassBuilder / CollapsedEMOptimizer need Boost and
// TBB headers, which this image lacks.  ref_eq_build below therefore drives the reference's own hash table and the
// reference's own XXH64 with addGroup's upsert (include/EquivalenceClassBuilder.hpp:90-108: "present -> count++,
// absent -> insert with count 1") on a key type written here: an ordered id list, equality = vector equality, hash =
// XXH64 of the ids' bytes with seed 0 -- what TranscriptGroup is (include/TranscriptGroup.hpp:9-49,
// src/TranscriptGroup.cpp:9-12, 53-55).
#include <cstdint>
#include <cstring>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "LibraryFormat.hpp"
#include "MultinomialSampler.hpp"
#include "cuckoohash_map.hh"
extern "C" {
#include "xxhash.h"
}

namespace {
struct Label {
    std::vector<uint32_t> txps;
    size_t hash;
    bool operator==(const Label& o) const { return txps == o.txps; }
};
struct LabelHasher {
    size_t operator()(const Label& l) const { return l.hash; }
};
}  // namespace

extern "C" {

// ---- LibraryFormat ---------------------------------------------------------------------------
uint8_t ref_format_id(int type, int orientation, int strandedness) {
    LibraryFormat f(static_cast<ReadType>(type), static_cast<ReadOrientation>(orientation), static_cast<ReadStrandedness>(strandedness));
    return f.formatID();
}
void ref_format_from_id(uint8_t id, int* out3) {
    LibraryFormat f = LibraryFormat::formatFromID(id);
    out3[0] = static_cast<int>(f.type); out3[1] = static_cast<int>(f.orientation); out3[2] = static_cast<int>(f.strandedness);
}
int ref_format_check(int type, int orientation, int strandedness) {
    LibraryFormat f(static_cast<ReadType>(type), static_cast<ReadOrientation>(orientation), static_cast<ReadStrandedness>(strandedness));
    return f.check() ? 1 : 0;
}
int ref_format_max_id() { return LibraryFormat::maxLibTypeID(); }
int ref_format_str(int type, int orientation, int strandedness, char* buf, int cap) {
    LibraryFormat f(static_cast<ReadType>(type), static_cast<ReadOrientation>(orientation), static_cast<ReadStrandedness>(strandedness));
    std::ostringstream os; os << f;
    std::string s = os.str();
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// ---- MultinomialSampler ------------------------------------------------------------------------
// one call of the reference's operator(): n draws over k categories (seeded from std::random_device, as in the
// reference: the draws differ from run to run -- parity is distributional)
void ref_multinomial(uint32_t n, uint32_t k, const double* probs, uint64_t* counts) {
    std::random_device rd;
    MultinomialSampler ms(rd);
    std::vector<double> p(probs, probs + k);
    std::vector<uint64_t> c(k, 0);
    ms(c.begin(), n, k, p.begin());
    for (uint32_t i = 0; i < k; ++i) counts[i] = c[i];
}

// ---- the class table: libcuckoo upsert + XXH64, as addGroup uses them -----------------------------
// reads r = ids[off[r] .. off[r+1]) (empty lists skipped, like the call sites' guard); n_threads > 1 splits the reads
// over std::threads that upsert concurrently, as the mapping threads do.  Results in table order (lock_table
// iteration, finish() :64-80): out_len / out_cnt / out_hash per class, labels concatenated in out_ids.
// Returns the number of classes, or -1 if the capacities are too small.
long ref_eq_build(const uint32_t* ids, const uint64_t* off, uint64_t n_reads, int n_threads,
                  uint32_t* out_ids, uint64_t cap_ids, uint32_t* out_len, uint64_t* out_cnt, uint64_t* out_hash, uint64_t cap_classes) {
    cuckoohash_map<Label, uint64_t, LabelHasher> table;
    table.reserve(1000000);                                                 // :57
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; ++r) {
            if (off[r + 1] == off[r]) continue;
            Label l;
            l.txps.assign(ids + off[r], ids + off[r + 1]);
            l.hash = XXH64(static_cast<const void*>(l.txps.data()), l.txps.size() * sizeof(uint32_t), 0);   // TranscriptGroup.cpp:9-12
            auto upfn = [](uint64_t& x) { x++; };                             // :94-96
            table.upsert(l, upfn, (uint64_t)1);                                // :105-106
        }
    };
    if (n_threads <= 1) work(0, n_reads);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, n_reads * t / n_threads, n_reads * (t + 1) / n_threads);
        for (auto& t : th) t.join();
    }
    uint64_t nc = 0, nid = 0;
    auto lt = table.lock_table();
    for (auto& kv : lt) {
        if (nc >= cap_classes || nid + kv.first.txps.size() > cap_ids) return -1;
        out_len[nc] = (uint32_t)kv.first.txps.size(); out_cnt[nc] = kv.second; out_hash[nc] = kv.first.hash;
        memcpy(out_ids + nid, kv.first.txps.data(), kv.first.txps.size() * sizeof(uint32_t));
        nid += kv.first.txps.size(); ++nc;
    }
    return (long)nc;
}

}  // extern "C"
