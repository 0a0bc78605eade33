// Reference-style C++ host code against include/sfgpu_sailfish.hpp: what src/SailfishQuantify.cpp does with the two
// classes on the hot path, written as its maintainers would -- mapping threads calling addGroup, finish(), optimize(),
// the samplers with their std::function writers -- and checked against the known answers of SURVEY.md 8c (outputs of
// the reference's own optimize()) and against a std::map.  Compiled by tests/test_abi.py; run on the GPU box.
#include <cmath>
#include <cstdio>
#include <map>
#include <thread>

#include "sfgpu_sailfish.hpp"

using namespace sailfish::gpu;

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)
static bool close_to(double a, double b, double rel) { return std::fabs(a - b) <= rel * std::fmax(1.0, std::fabs(b)); }

static void toy(ReadExperiment& exp) {          // lens [1000,2000,500,1500], EffectiveLength = len - 199
    const uint32_t lens[4] = {1000, 2000, 500, 1500};
    for (size_t i = 0; i < 4; ++i) { exp.transcripts().emplace_back(i, ("t" + std::to_string(i)).c_str(), lens[i]); exp.transcripts().back().EffectiveLength = lens[i] - 199.0; }
}

int main() {
    std::vector<std::string> lines;
    SailfishOpts sopt;
    sopt.jointLog = [&lines](int, const std::string& m) { lines.push_back(m); };

    // ---- builder known answer: reads {1,2,3}x3, {5}x2, {2,9}x1 -> 3 classes, counts 3/2/1
    {
        EquivalenceClassBuilder b;
        b.start();
        std::vector<double> w;
        const std::vector<std::vector<uint32_t>> reads = {{1, 2, 3}, {5}, {1, 2, 3}, {2, 9}, {5}, {1, 2, 3}};
        for (auto& r : reads) { w.assign(r.size(), 1.0); b.addGroup(TranscriptGroup(r), w); }
        EXPECT(b.finish());
        auto& v = b.eqVec();
        EXPECT(v.size() == 3 && b.totalReads() == 6);
        std::map<std::vector<uint32_t>, uint64_t> got;
        for (auto& kv : v) got[kv.first.txps] = kv.second.count;
        EXPECT((got[{1, 2, 3}] == 3 && got[{5}] == 2 && got[{2, 9}] == 1));
        for (auto& kv : v) if (kv.first.txps == std::vector<uint32_t>{1, 2, 3}) EXPECT(kv.first.hash == 0xb5148cb100a911fcull);   // XXH64, SURVEY 8c
    }

    // ---- mapping threads: 8 threads x 20000 reads through addGroup, against a std::map
    {
        ReadExperiment exp;
        auto& b = exp.equivalenceClassBuilder();
        b.start();
        const int T = 8, N = 20000;
        std::vector<std::map<std::vector<uint32_t>, uint64_t>> local(T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
            std::mt19937_64 g(1234 + t);
            std::vector<double> w;
            for (int i = 0; i < N; ++i) {
                const uint32_t base = g() % 500, k = 1 + g() % 5;
                std::vector<uint32_t> lab;
                for (uint32_t j = 0; j < k; ++j) lab.push_back(base + 7 * j);
                ++local[t][lab];
                w.assign(k, 1.0);
                b.addGroup(TranscriptGroup(lab), w);
                exp.numMappedFragmentsAtomic()++;
            }
        });
        for (auto& x : th) x.join();
        EXPECT(b.finish());
        std::map<std::vector<uint32_t>, uint64_t> want;
        for (auto& m : local) for (auto& kv : m) want[kv.first] += kv.second;
        auto& v = b.eqVec();
        EXPECT(v.size() == want.size() && b.totalReads() == (uint64_t)T * N && exp.numMappedFragments() == (uint64_t)T * N);
        bool same = true;
        for (auto& kv : v) same = same && want.count(kv.first.txps) && want[kv.first.txps] == kv.second.count;
        EXPECT(same);
    }

    // ---- optimize(): the toy of SURVEY 8c through insertGroup, EM and VBEM
    const double em_want[4] = {417.47751898139057, 0.0, 67.522481018609454, 0.0};
    const double vb_want[4] = {417.52276075723859, 0.0, 67.497239242761381, 0.0};
    for (int vb = 0; vb < 2; ++vb) {
        ReadExperiment exp(sopt.jointLog);
        toy(exp);
        auto& b = exp.equivalenceClassBuilder();
        b.start();
        b.insertGroup(TranscriptGroup({0}), 100); b.insertGroup(TranscriptGroup({0, 1}), 300); b.insertGroup(TranscriptGroup({1, 2}), 50);
        b.insertGroup(TranscriptGroup({0, 1, 2}), 25); b.insertGroup(TranscriptGroup({2}), 10);
        EXPECT(b.finish() && b.numClasses() == 5 && b.totalReads() == 485);
        exp.numMappedFragmentsAtomic() = 485;
        sopt.useVBOpt = vb != 0;
        CollapsedEMOptimizer opt;
        EXPECT(opt.optimize(exp, sopt, 0.01, 10000));
        EXPECT(opt.lastIterations == 50);
        for (int i = 0; i < 4; ++i) EXPECT(close_to(exp.transcripts()[i].estCount(), (vb ? vb_want : em_want)[i], vb ? 1e-12 : 1e-14));
        EXPECT(close_to(exp.transcripts()[0].mass(), vb ? 417.52276075723859 / (417.52276075723859 + 67.497239242761381) : 0.86077838965235165, 1e-12));
        EXPECT(exp.transcripts()[0].getActive() && exp.transcripts()[1].getActive() && !exp.transcripts()[3].getActive());
        if (!vb) {
            // the samplers with the reference's writer signatures
            sopt.numBootstraps = 5;
            int n_boot = 0; double boot_sum = 0.0;
            std::function<bool(const std::vector<double>&)> bw = [&](const std::vector<double>& a) { ++n_boot; for (double x : a) boot_sum += x; return true; };
            EXPECT(opt.gatherBootstraps(exp, sopt, bw, 0.01, 10000));
            EXPECT(n_boot == 5 && close_to(boot_sum, 5 * 485.0, 1e-9));
            int n_gibbs = 0; long gibbs_sum = 0;
            std::function<bool(const std::vector<int>&)> gw = [&](const std::vector<int>& c) { ++n_gibbs; for (int x : c) gibbs_sum += x; return true; };
            CollapsedGibbsSampler sampler;
            EXPECT(sampler.sample(exp, sopt, gw, 6));
            EXPECT(n_gibbs == 6 && gibbs_sum == 6 * 485);
        }
    }
    bool saw_iter_line = false, saw_classes_line = false;
    for (auto& l : lines) { saw_iter_line = saw_iter_line || l.find("iteration = 50") != std::string::npos; saw_classes_line = saw_classes_line || l.find("Optimizing over 5 equivalence classes") != std::string::npos; }
    EXPECT(saw_iter_line && saw_classes_line);                     // what the reference logs through jointLog (:790, :871)

    // ---- optimize() with doBiasCorrect (:717, :814-840, :888): pairs of transcripts sharing most fragments, the second
    //      of each pair without unique evidence (the EM runs well past iteration 50), both bias models
    for (int model = 0; model < 2; ++model) {
        ReadExperiment exp;
        std::mt19937_64 g(99);
        const int M = 40;
        std::string seq; std::vector<uint64_t> off;
        for (int t = 0; t < M; ++t) {
            const uint32_t L = 600 + g() % 1500;
            off.push_back(seq.size());
            for (uint32_t i = 0; i < L; ++i) seq.push_back("ACGT"[g() % 4]);
            seq.push_back('$');
            exp.transcripts().emplace_back(t, ("t" + std::to_string(t)).c_str(), L);
            exp.transcripts().back().EffectiveLength = L - 199.0;
        }
        exp.setSequences(seq, off);
        std::vector<int32_t> fld(1000, 0);
        for (int i = 0; i < 1000; ++i) { const int d = i - 200; fld[i] = (d > -150 && d < 150) ? 150 - std::abs(d) : 0; }
        exp.setFragLengthDist(fld);
        for (int i = 0; i < 4096; ++i) exp.readBias()[i] = 1 + (g() % 50);
        for (int i = 0; i < 101; ++i) exp.observedGC()[i] = 1 + ((i > 30 && i < 70) ? 200 + g() % 50 : g() % 3);
        exp.addNumFwd(600); exp.addNumRC(400);
        auto& b = exp.equivalenceClassBuilder();
        b.start();
        uint64_t total = 0;
        for (uint32_t j = 0; j < M / 2; ++j) {
            const uint32_t s = 3000 + g() % 2000, u = 150 + g() % 100;
            b.insertGroup(TranscriptGroup({2 * j, 2 * j + 1}), s); b.insertGroup(TranscriptGroup({2 * j}), u);
            total += s + u;
        }
        EXPECT(b.finish());
        exp.numMappedFragmentsAtomic() = total;
        SailfishOpts bo; bo.biasCorrect = model == 0; bo.gcBiasCorrect = model == 1;
        CollapsedEMOptimizer opt;
        EXPECT(opt.optimize(exp, bo, 0.01, 10000));
        EXPECT(opt.lastIterations > 50 && opt.lastRecomputes >= 1);
        int changed = 0; double est = 0.0, exp_sum = 0.0;
        for (int t = 0; t < M; ++t) { changed += exp.transcripts()[t].EffectiveLength != exp.transcripts()[t].RefLength - 199.0; est += exp.transcripts()[t].estCount(); }
        for (double v : (model == 0 ? exp.expectedSeqBias() : exp.expectedGCBias())) exp_sum += v;
        EXPECT(changed > 0 && close_to(est, (double)total, 1e-9) && exp_sum > (model == 0 ? 4096.0 : 101.0) + 1.0);
        std::printf("bias model %d: %u iterations, %u recomputes, %d lengths corrected\n", model, opt.lastIterations, opt.lastRecomputes, changed);
    }

    // ---- optimize() returns false where the reference does: no transcript is expressed (:794-798)
    {
        ReadExperiment exp; toy(exp);
        exp.equivalenceClassBuilder().start(); exp.equivalenceClassBuilder().finish();
        SailfishOpts quiet;
        CollapsedEMOptimizer opt;
        EXPECT(!opt.optimize(exp, quiet, 0.01, 10000));
    }
    std::printf(failures ? "cpp host FAILED (%d)\n" : "cpp host ok\n", failures);
    return failures ? 1 : 0;
}
