// sfgpu_sailfish.hpp -- the C++ host side above the C ABI: drop-in counterparts of the reference classes that sit on
// the quantification hot path, with the reference's names, argument meaning and error behaviour, so that host code
// written against the reference (src/SailfishQuantify.cpp) compiles against these instead.
//
//   sailfish::gpu::TranscriptGroup / TGValue      include/TranscriptGroup.hpp:9-35, EquivalenceClassBuilder.hpp:5-38
//   sailfish::gpu::EquivalenceClassBuilder        include/EquivalenceClassBuilder.hpp:40-119
//   sailfish::gpu::Transcript                     include/Transcript.hpp:14-99, 204-206 (the members the path touches)
//   sailfish::gpu::SailfishOpts                   include/SailfishOpts.hpp:9-41 (the members the path reads)
//   sailfish::gpu::ReadExperiment                 include/ReadExperiment.hpp:65-99, 236-257
//   sailfish::gpu::CollapsedEMOptimizer           include/CollapsedEMOptimizer.hpp:20-35, src/CollapsedEMOptimizer.cpp:557-893
//   sailfish::gpu::CollapsedGibbsSampler          include/CollapsedGibbsSampler.hpp:22-32, src/CollapsedGibbsSampler.cpp:187-291
//
// Header only; needs sfgpu.h, the HIP runtime API (hipMalloc / hipMemcpy for the caller-owned buffers the ABI takes)
// and C++14.  No Boost, TBB, spdlog or Eigen: the logger is a std::function<void(int level, const std::string&)>.
// Compiled and run by tests/test_abi.py (tests/cpp_host_test.cpp).
#ifndef SFGPU_SAILFISH_HPP
#define SFGPU_SAILFISH_HPP

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "sfgpu.h"

namespace sailfish {
namespace gpu {

using Logger = std::function<void(int, const std::string&)>;     // level 0 info, 1 warn, 2 error (spdlog's jointLog)

inline void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
inline void check(int rc, const char* what) {
    if (rc != SFGPU_OK) throw std::runtime_error(std::string(what) + ": " + sfgpu_last_error());
}

// caller-owned device buffer (the ABI never allocates what it hands back)
template <typename T>
class DeviceBuf {
  public:
    DeviceBuf() = default;
    explicit DeviceBuf(size_t n) { resize(n); }
    explicit DeviceBuf(const std::vector<T>& h) { resize(h.size()); upload(h); }
    DeviceBuf(const DeviceBuf&) = delete;
    DeviceBuf& operator=(const DeviceBuf&) = delete;
    ~DeviceBuf() { if (p_) (void)hipFree(p_); }
    void resize(size_t n) {
        if (p_) { (void)hipFree(p_); p_ = nullptr; }
        n_ = n;
        check_hip(hipMalloc(reinterpret_cast<void**>(&p_), (n ? n : 1) * sizeof(T)), "hipMalloc");
    }
    void upload(const std::vector<T>& h) { if (n_) check_hip(hipMemcpy(p_, h.data(), n_ * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy"); }
    std::vector<T> download() const {
        std::vector<T> h(n_);
        if (n_) check_hip(hipMemcpy(h.data(), p_, n_ * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy");
        return h;
    }
    T* get() const { return p_; }
    size_t size() const { return n_; }
  private:
    T* p_ = nullptr; size_t n_ = 0;
};

// ---- include/TranscriptGroup.hpp:9-35 --------------------------------------------------------------------
class TranscriptGroup {
  public:
    TranscriptGroup() = default;
    explicit TranscriptGroup(std::vector<uint32_t> txpsIn) : txps(std::move(txpsIn)) {}
    TranscriptGroup(std::vector<uint32_t> txpsIn, size_t hashIn) : txps(std::move(txpsIn)), hash(hashIn) {}
    std::vector<uint32_t> txps;
    size_t hash = 0;            // XXH64 of the id bytes, seed 0 (src/TranscriptGroup.cpp:9-12): filled by eqVec()
    double totalMass = 0.0;
    mutable bool valid = true;
};
inline bool operator==(const TranscriptGroup& a, const TranscriptGroup& b) { return a.txps == b.txps; }

// EquivalenceClassBuilder.hpp:5-38.  The weights are 1.0 at every addGroup call site and optimize() overwrites them;
// eqVec() returns them as optimize() would first set them only on request (they are not stored on the device).
struct TGValue {
    TGValue() = default;
    TGValue(std::vector<double> w, uint64_t c) : weights(std::move(w)), count(c) {}
    mutable std::vector<double> weights;
    uint64_t count = 0;
};

// ---- include/EquivalenceClassBuilder.hpp:40-119 ----------------------------------------------------------
class EquivalenceClassBuilder {
  public:
    explicit EquivalenceClassBuilder(Logger loggerIn = nullptr) : logger_(std::move(loggerIn)) {
        check(sfgpu_eq_create(&h_, 1000000 /* countMap_.reserve(1000000), :44 */, nullptr), "sfgpu_eq_create");
    }
    ~EquivalenceClassBuilder() { if (h_) sfgpu_eq_destroy(h_); }
    EquivalenceClassBuilder(const EquivalenceClassBuilder&) = delete;
    EquivalenceClassBuilder& operator=(const EquivalenceClassBuilder&) = delete;

    void start() { check(sfgpu_eq_start(h_), "sfgpu_eq_start"); active_ = true; vec_.clear(); }        // :62

    // :90-108, called concurrently by the mapping threads; a thread's reads travel in batches of at most 1000
    // (the reference's parser job size, src/SailfishQuantify.cpp:73).  `weights` are all 1.0 and are dropped.
    inline void addGroup(TranscriptGroup&& g, std::vector<double>& /*weights*/) {
        Batch& b = my_batch();
        b.ids.insert(b.ids.end(), g.txps.begin(), g.txps.end());
        b.offsets.push_back(static_cast<uint32_t>(b.ids.size()));
        if (b.offsets.size() > 1000) flush(b);
    }

    // :82-88  single-threaded bulk insert of a known group with a count
    inline void insertGroup(TranscriptGroup g, uint32_t count) {
        pending_ids_.insert(pending_ids_.end(), g.txps.begin(), g.txps.end());
        pending_off_.push_back(static_cast<uint32_t>(pending_ids_.size()));
        pending_cnt_.push_back(count);
    }

    // :64-80  every mapping thread has joined by now (src/SailfishQuantify.cpp:941): their last partial batches go in
    bool finish() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (auto& b : batches_) flush(*b);
        }
        if (!pending_cnt_.empty()) {
            DeviceBuf<uint32_t> ids(pending_ids_), off(pending_off_); DeviceBuf<uint64_t> cnt(pending_cnt_);
            check(sfgpu_eq_add_weighted_device(h_, ids.get(), off.get(), cnt.get(), static_cast<uint32_t>(pending_cnt_.size())),
                  "sfgpu_eq_add_weighted_device");
            pending_ids_.clear(); pending_off_.assign(1, 0); pending_cnt_.clear();
        }
        check(sfgpu_eq_finish(h_, &n_classes_, &nnz_, &total_reads_), "sfgpu_eq_finish");
        active_ = false;
        return true;
    }

    // :110-112  the classes on the host, canonical order (first id, hash, length, label) -- the reference's order is
    // the cuckoo table's and changes from run to run.  Only writeEquivCounts needs this copy; optimize() does not.
    std::vector<std::pair<const TranscriptGroup, TGValue>>& eqVec() {
        if (vec_.empty() && n_classes_) {
            std::vector<uint32_t> rowptr(n_classes_ + 1), ids(nnz_ ? nnz_ : 1);
            std::vector<uint64_t> counts(n_classes_), hashes(n_classes_);
            check(sfgpu_eq_export_host(h_, rowptr.data(), ids.data(), counts.data(), hashes.data()), "sfgpu_eq_export_host");
            vec_.reserve(n_classes_);
            for (uint64_t c = 0; c < n_classes_; ++c) {
                std::vector<uint32_t> lab(ids.begin() + rowptr[c], ids.begin() + rowptr[c + 1]);
                const size_t k = lab.size();
                vec_.emplace_back(TranscriptGroup(std::move(lab), static_cast<size_t>(hashes[c])), TGValue(std::vector<double>(k, 1.0), counts[c]));
            }
        }
        return vec_;
    }

    uint64_t numClasses() const { return n_classes_; }
    uint64_t numNonzeros() const { return nnz_; }
    uint64_t totalReads() const { return total_reads_; }       // what finish() logs as "Counted ... total reads" (:77-78)
    sfgpu_eq* handle() const { return h_; }

  private:
    struct Batch { std::vector<uint32_t> ids; std::vector<uint32_t> offsets{0}; };
    static uint64_t next_id() { static std::atomic<uint64_t> n{1}; return n.fetch_add(1); }
    Batch& my_batch() {
        thread_local std::vector<std::pair<uint64_t, Batch*>> mine;       // keyed by a never-reused builder id
        for (auto& e : mine) if (e.first == id_) return *e.second;
        std::lock_guard<std::mutex> lk(mu_);
        batches_.emplace_back(new Batch());
        mine.emplace_back(id_, batches_.back().get());
        return *batches_.back();
    }
    void flush(Batch& b) {
        const uint32_t n = static_cast<uint32_t>(b.offsets.size() - 1);
        if (n) check(sfgpu_eq_add_batch_host(h_, b.ids.empty() ? &zero_ : b.ids.data(), b.offsets.data(), n), "sfgpu_eq_add_batch_host");
        b.ids.clear(); b.offsets.assign(1, 0);
    }
    sfgpu_eq* h_ = nullptr;
    const uint64_t id_ = next_id();
    Logger logger_;
    bool active_ = false;
    std::mutex mu_;
    std::vector<std::unique_ptr<Batch>> batches_;
    std::vector<uint32_t> pending_ids_, pending_off_{0}; std::vector<uint64_t> pending_cnt_;
    uint64_t n_classes_ = 0, nnz_ = 0, total_reads_ = 0;
    std::vector<std::pair<const TranscriptGroup, TGValue>> vec_;
    uint32_t zero_ = 0;
};

// ---- include/Transcript.hpp (the members the path reads and writes) ---------------------------------------
class Transcript {
  public:
    Transcript(size_t idIn, const char* name, uint32_t len) : RefName(name), RefLength(len), EffectiveLength(-1.0), id(static_cast<uint32_t>(idIn)) {}
    void setEstCount(double sc) { estCount_ = sc; }
    double estCount() const { return estCount_; }
    void setMass(double m) { mass_ = m; }
    double mass() const { return mass_; }
    void setActive() { active_ = true; }
    bool getActive() const { return active_; }
    std::string RefName;
    uint32_t RefLength;
    double EffectiveLength;
    uint32_t id;
  private:
    double mass_ = 0.0, estCount_ = 0.0;
    bool active_ = false;
};

// ---- include/SailfishOpts.hpp:9-41 (the members the path consults) ----------------------------------------
struct SailfishOpts {
    uint32_t numThreads = 1;
    bool useVBOpt = false;
    bool noEffectiveLengthCorrection = false;
    uint32_t numBootstraps = 0;
    uint32_t numGibbsSamples = 0;
    bool biasCorrect = false, gcBiasCorrect = false;
    uint32_t gcSampFactor = 1;      // --gcSizeSamp
    uint32_t pdfSampFactor = 1;     // --gcSpeedSamp
    Logger jointLog;
};

// ---- include/ReadExperiment.hpp:65-99, 236-257 ------------------------------------------------------------
class ReadExperiment {
  public:
    explicit ReadExperiment(Logger log = nullptr) : eqBuilder_(std::move(log)) {}
    std::vector<Transcript>& transcripts() { return transcripts_; }
    EquivalenceClassBuilder& equivalenceClassBuilder() { return eqBuilder_; }
    uint64_t numMappedFragments() const { return numMappedFragments_.load(); }
    std::atomic<uint64_t>& numMappedFragmentsAtomic() { return numMappedFragments_; }
    std::atomic<uint64_t>& numObservedFragmentsAtomic() { return numObservedFragments_; }
    double mappingRate() const {
        const double obs = static_cast<double>(numObservedFragments_.load());
        return obs > 0.0 ? static_cast<double>(numMappedFragments_.load()) / obs : 0.0;
    }
    // what bias correction reads and writes (:93-97, 160-212, 240-255)
    void setSequences(std::string seq, std::vector<uint64_t> txpOffsets) { seq_ = std::move(seq); seqOff_ = std::move(txpOffsets); }   // RapMapSAIndex::seq / txpOffsets (:108-117)
    const std::string& sequences() const { return seq_; }
    const std::vector<uint64_t>& sequenceOffsets() const { return seqOff_; }
    void setFragLengthDist(const std::vector<int32_t>& fldIn) { fld_.assign(fldIn.begin(), fldIn.end()); }
    const std::vector<uint32_t>& fragLengthCounts() const { return fld_; }
    std::vector<uint32_t>& readBias() { return readBias_; }                      // ReadKmerDist<6>::counts, pseudo-count 1
    std::vector<uint32_t>& observedGC() { return observedGC_; }                  // 101 bins, pseudo-count 1
    std::vector<double>& expectedSeqBias() { return expectedSeqBias_; }
    std::vector<double>& expectedGCBias() { return expectedGC_; }
    void addNumFwd(int32_t n) { numFwd_ += n; }
    void addNumRC(int32_t n) { numRC_ += n; }
    int64_t numFwd() const { return numFwd_.load(); }
    int64_t numRC() const { return numRC_.load(); }
  private:
    std::vector<Transcript> transcripts_;
    EquivalenceClassBuilder eqBuilder_;
    std::atomic<uint64_t> numMappedFragments_{0}, numObservedFragments_{0};
    std::string seq_; std::vector<uint64_t> seqOff_;
    std::vector<uint32_t> fld_;
    std::vector<uint32_t> readBias_ = std::vector<uint32_t>(4096, 1), observedGC_ = std::vector<uint32_t>(101, 1);
    std::vector<double> expectedSeqBias_ = std::vector<double>(4096, 1.0), expectedGC_ = std::vector<double>(101, 1.0);
    std::atomic<int64_t> numFwd_{0}, numRC_{0};
};

namespace detail {
inline Logger* active_logger(Logger* set = nullptr, bool clear = false) {
    static Logger* cur = nullptr;
    if (set) cur = set;
    if (clear) cur = nullptr;
    return cur;
}
inline void log_trampoline(int level, const char* msg) { if (Logger* l = active_logger()) if (*l) (*l)(level, msg); }
struct LoggerScope {       // the library logs what the reference logs (iteration lines, class counts) through jointLog
    explicit LoggerScope(Logger& l) { if (l) { active_logger(&l); sfgpu_set_logger(log_trampoline); } }
    ~LoggerScope() { active_logger(nullptr, true); sfgpu_set_logger(nullptr); }
};

// the device-side problem of one experiment: lengths + the builder's classes (never leave HBM)
struct DeviceProblem {
    DeviceBuf<double> len;
    DeviceBuf<uint32_t> rowptr, ids;
    DeviceBuf<uint64_t> counts;
    sfgpu_problem prob{};
    DeviceProblem(ReadExperiment& readExp, const SailfishOpts& sopt) {
        auto& txps = readExp.transcripts();
        auto& eq = readExp.equivalenceClassBuilder();
        std::vector<double> h(txps.size());
        for (size_t i = 0; i < txps.size(); ++i)                                   // src/CollapsedEMOptimizer.cpp:736-737
            h[i] = sopt.noEffectiveLengthCorrection ? static_cast<double>(txps[i].RefLength) : txps[i].EffectiveLength;
        len.resize(h.size()); len.upload(h);
        rowptr.resize(eq.numClasses() + 1); ids.resize(eq.numNonzeros()); counts.resize(eq.numClasses());
        check(sfgpu_eq_export_device(eq.handle(), rowptr.get(), ids.get(), counts.get(), nullptr), "sfgpu_eq_export_device");
        prob = sfgpu_problem{txps.size(), len.get(), eq.numClasses(), rowptr.get(), ids.get(), counts.get(), readExp.numMappedFragments()};
    }
};
}  // namespace detail

// ---- include/CollapsedEMOptimizer.hpp:20-35 ----------------------------------------------------------------
class CollapsedEMOptimizer {
  public:
    CollapsedEMOptimizer() = default;

    // src/CollapsedEMOptimizer.cpp:711-893.  false where the reference logs an error and returns false
    // ("no transcripts expressed" :794-798, "total alpha weight was too small" :877-881).
    bool optimize(ReadExperiment& readExp, SailfishOpts& sopt, double tolerance = 0.01, uint32_t maxIter = 1000) {
        const bool doBiasCorrect = sopt.biasCorrect || sopt.gcBiasCorrect;         // :717
        detail::LoggerScope scope(sopt.jointLog);
        auto& txps = readExp.transcripts();
        const uint64_t M = txps.size();
        detail::DeviceProblem dp(readExp, sopt);
        DeviceBuf<double> alpha(M), mass(M);
        sfgpu_em* em = nullptr;
        check(sfgpu_em_create(&em, &dp.prob, nullptr), "sfgpu_em_create");
        sfgpu_em_opts o{sopt.useVBOpt ? 1 : 0, tolerance, /*minIter :716*/ 50, maxIter, /*check_mode*/ 0, 0};
        sfgpu_em_stats st{};
        int rc;
        std::vector<double> newEffLens;
        lastRecomputes = 0;
        if (doBiasCorrect) {
            // everything updateEffectiveLengths reads from the experiment (src/SailfishUtils.cpp:611-690), then the
            // loop with the recompute hook (:814-840); the corrected lengths come back for :888
            std::vector<char> seq(readExp.sequences().begin(), readExp.sequences().end());
            std::vector<uint32_t> refLens(M); std::vector<double> txpEff(M);
            for (uint64_t i = 0; i < M; ++i) { refLens[i] = txps[i].RefLength; txpEff[i] = txps[i].EffectiveLength; }
            DeviceBuf<char> dSeq(seq); DeviceBuf<uint64_t> dOff(readExp.sequenceOffsets());
            DeviceBuf<uint32_t> dRef(refLens); DeviceBuf<double> dTxpEff(txpEff), dEffOut(M);
            sfgpu_bias_inputs bi{};
            bi.M = M; bi.d_seq = dSeq.get(); bi.d_seq_off = dOff.get(); bi.d_ref_len = dRef.get(); bi.d_txp_eff_len = dTxpEff.get();
            bi.h_fl_counts = readExp.fragLengthCounts().data(); bi.max_frag_len = static_cast<uint32_t>(readExp.fragLengthCounts().size());
            bi.gc_speed_samp = sopt.pdfSampFactor; bi.h_read_bias = readExp.readBias().data(); bi.h_observed_gc = readExp.observedGC().data();
            bi.num_fwd = readExp.numFwd(); bi.num_rc = readExp.numRC();
            bi.seq_bias = sopt.biasCorrect; bi.gc_bias = sopt.gcBiasCorrect; bi.gc_size_samp = sopt.gcSampFactor;
            sfgpu_bias* bias = nullptr;
            rc = sfgpu_bias_create(&bias, &bi, nullptr);
            if (rc != SFGPU_OK) { sfgpu_em_destroy(em); check(rc, "sfgpu_bias_create"); }
            rc = sfgpu_em_optimize_bias(em, &o, bias, alpha.get(), mass.get(), dEffOut.get(), &lastRecomputes, &st);
            if (rc == SFGPU_OK) {
                (void)sfgpu_bias_expected(bias, readExp.expectedSeqBias().data(), readExp.expectedGCBias().data());
                newEffLens = dEffOut.download();
            }
            sfgpu_bias_destroy(bias);
        } else {
            rc = sfgpu_em_optimize(em, &o, alpha.get(), mass.get(), &st);
        }
        sfgpu_em_destroy(em);
        lastIterations = st.iters;
        if (rc == SFGPU_ERR_NO_ACTIVE || rc == SFGPU_ERR_ALPHA_SUM) return false;
        check(rc, "sfgpu_em_optimize");
        for (size_t i = 0; i < newEffLens.size(); ++i) txps[i].EffectiveLength = newEffLens[i];   // :888
        const std::vector<double> a = alpha.download(), m = mass.download();
        std::vector<uint32_t> members = dp.ids.download();
        for (uint32_t t : members) txps[t].setActive();                            // :774-782
        for (size_t i = 0; i < txps.size(); ++i) { txps[i].setEstCount(a[i]); txps[i].setMass(m[i]); }   // :885-891
        return true;
    }

    // src/CollapsedEMOptimizer.cpp:557-709 (doBootstrap :438-525).  The reference seeds from std::random_device.
    bool gatherBootstraps(ReadExperiment& readExp, SailfishOpts& sopt,
                          std::function<bool(const std::vector<double>&)>& writeBootstrap,
                          double relDiffTolerance, uint32_t maxIter) {
        detail::LoggerScope scope(sopt.jointLog);
        detail::DeviceProblem dp(readExp, sopt);
        sfgpu_em* em = nullptr;
        check(sfgpu_em_create(&em, &dp.prob, nullptr), "sfgpu_em_create");
        sfgpu_em_opts o{sopt.useVBOpt ? 1 : 0, relDiffTolerance, /*min_iter*/ 0, maxIter, /*check_mode :499*/ 1, 0};
        struct Ctx { std::function<bool(const std::vector<double>&)>* w; } ctx{&writeBootstrap};
        auto cb = [](const double* a, uint64_t M, void* user) -> int {
            std::vector<double> v(a, a + M);
            return (*static_cast<Ctx*>(user)->w)(v) ? 1 : 0;
        };
        std::random_device rd;
        const uint64_t seed = (static_cast<uint64_t>(rd()) << 32) | rd();
        const int rc = sfgpu_bootstrap(em, &o, sopt.numBootstraps, seed, nullptr, cb, &ctx, nullptr);
        sfgpu_em_destroy(em);
        if (rc == SFGPU_ERR_NO_ACTIVE || rc == SFGPU_ERR_ALPHA_SUM) return false;
        check(rc, "sfgpu_bootstrap");
        return true;
    }

    uint32_t lastIterations = 0;       // the N of the reference's log line "iteration = N | max rel diff. = x" (:871-872)
    uint32_t lastRecomputes = 0;       // how often "recomputing effective lengths" (:827) happened
};

// ---- include/CollapsedGibbsSampler.hpp:22-32 ---------------------------------------------------------------
class CollapsedGibbsSampler {
  public:
    CollapsedGibbsSampler() = default;
    // src/CollapsedGibbsSampler.cpp:187-291; reads Transcript::mass() as optimize() left it (it does not overwrite
    // it, unlike :219-221)
    bool sample(ReadExperiment& readExp, SailfishOpts& sopt, std::function<bool(const std::vector<int>&)>& writeBootstrap,
                uint32_t numSamples = 500) {
        detail::LoggerScope scope(sopt.jointLog);
        auto& txps = readExp.transcripts();
        detail::DeviceProblem dp(readExp, sopt);
        std::vector<double> m(txps.size());
        for (size_t i = 0; i < txps.size(); ++i) m[i] = txps[i].mass();
        DeviceBuf<double> mass(m);
        struct Ctx { std::function<bool(const std::vector<int>&)>* w; } ctx{&writeBootstrap};
        auto cb = [](const int32_t* c, uint64_t M, void* user) -> int {
            std::vector<int> v(c, c + M);
            return (*static_cast<Ctx*>(user)->w)(v) ? 1 : 0;
        };
        std::random_device rd;
        const uint64_t seed = (static_cast<uint64_t>(rd()) << 32) | rd();
        const int rc = sfgpu_gibbs_sample(&dp.prob, mass.get(), numSamples, 0, seed, nullptr, cb, &ctx, nullptr);
        if (rc != SFGPU_OK) { if (sopt.jointLog) sopt.jointLog(2, sfgpu_last_error()); return false; }
        return true;
    }
};

}  // namespace gpu
}  // namespace sailfish
#endif  // SFGPU_SAILFISH_HPP
