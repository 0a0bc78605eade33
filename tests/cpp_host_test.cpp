// Reference-style C++ host code against include/sfgpu_sailfish.hpp: what src/SailfishQuantify.cpp does with the two
// classes on the hot path, written as its maintainers would -- mapping threads calling addGroup, finish(), optimize(),
// the samplers with their std::function writers -- and checked against the known answers of SURVEY.md 8c (outputs of
// the reference's own optimize()) and against a std::map.  Compiled by tests/test_abi.py; run on the GPU box.
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <map>
#include <mutex>
#include <random>
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

// ---- SURVEY 8e through the C ABI alone: two ranks (threads) share this box's GPU.  Each builds the table of its half of
// the reads; owner-partitioned exchange (pack -> "all-to-all" -> fold -> export -> "all-gather" -> merge); then the sharded
// EM, classes cut in two, alphaOut summed over the ranks by the caller's all-reduce between sweep and update.  The
// transport here is device pointers handed over at a barrier -- a multi-GPU host puts ncclSend / ncclAllReduce there.
namespace {
struct Barrier {
    std::mutex m; std::condition_variable cv; int n, count = 0, gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() { std::unique_lock<std::mutex> lk(m); const int g = gen; if (++count == n) { count = 0; ++gen; cv.notify_all(); } else cv.wait(lk, [&] { return gen != g; }); }
};
struct Table { std::vector<uint32_t> rowptr, ids; std::vector<uint64_t> counts, hashes; };
struct Shared {
    Barrier bar{2};
    const void* blk[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // blk[src][dst]
    uint64_t sz[2][2][2] = {};                                          // [src][dst] -> (classes, ids)
    const void* part[2] = {nullptr, nullptr}; uint64_t part_c[2] = {0, 0}, part_l[2] = {0, 0};
    std::vector<double> red[2];
    Table merged[2]; std::vector<double> alpha[2]; sfgpu_em_stats st[2];
};
struct ReduceCtx { Shared* sh; int rank; };
int allreduce_two_threads(double* d_buf, uint64_t n, void* user, sfgpu_stream stream) {
    ReduceCtx* c = static_cast<ReduceCtx*>(user);
    if (hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)) != hipSuccess) return 1;
    c->sh->red[c->rank].resize(n);
    if (hipMemcpy(c->sh->red[c->rank].data(), d_buf, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    c->sh->bar.wait();
    std::vector<double> sum(n);
    for (uint64_t i = 0; i < n; ++i) sum[i] = c->sh->red[0][i] + c->sh->red[1][i];       // same order on both ranks: same bits
    c->sh->bar.wait();
    return hipMemcpy(d_buf, sum.data(), n * 8, hipMemcpyHostToDevice) == hipSuccess ? 0 : 3;
}
}  // namespace

static int two_ranks_on_one_gpu() {
    const int before = failures;
    const uint32_t M = 3000;
    // reads: labels {base + 7 j}, a few thousand distinct; every label shows up on both ranks
    std::vector<std::vector<uint32_t>> reads;
    std::mt19937_64 g(2026);
    for (int i = 0; i < 200000; ++i) {
        const uint32_t base = g() % 2500, k = 1 + g() % 9;
        std::vector<uint32_t> lab;
        for (uint32_t j = 0; j < k; ++j) lab.push_back((base + 7 * j) % M);
        std::sort(lab.begin(), lab.end()); lab.erase(std::unique(lab.begin(), lab.end()), lab.end());
        reads.push_back(lab);
    }
    auto pack = [&](size_t lo, size_t hi, std::vector<uint32_t>& ids, std::vector<uint32_t>& off) {
        ids.clear(); off.assign(1, 0);
        for (size_t r = lo; r < hi; ++r) { ids.insert(ids.end(), reads[r].begin(), reads[r].end()); off.push_back((uint32_t)ids.size()); }
    };
    auto build = [&](sfgpu_eq* eq, const std::vector<uint32_t>& ids, const std::vector<uint32_t>& off) {
        check(sfgpu_eq_start(eq), "start");
        check(sfgpu_eq_add_batch_host(eq, ids.data(), off.data(), (uint32_t)off.size() - 1), "add");
    };
    struct Dev { DeviceBuf<uint32_t> rowptr, ids; DeviceBuf<uint64_t> counts, hashes; uint64_t C = 0, L = 0; };
    auto finish_export = [&](sfgpu_eq* eq, Dev& d) {
        uint64_t tot = 0;
        check(sfgpu_eq_finish(eq, &d.C, &d.L, &tot), "finish");
        d.rowptr.resize(d.C + 1); d.ids.resize(d.L); d.counts.resize(d.C); d.hashes.resize(d.C);
        check(sfgpu_eq_export_device(eq, d.rowptr.get(), d.ids.get(), d.counts.get(), d.hashes.get()), "export");
        check_hip(hipDeviceSynchronize(), "sync");
    };
    // the single-builder table over all reads, and its EM
    Table want; std::vector<double> want_alpha(M); sfgpu_em_stats want_st{};
    std::vector<double> len(M);
    for (uint32_t t = 0; t < M; ++t) len[t] = 300.0 + (double)((t * 2654435761u) % 3000);
    DeviceBuf<double> d_len(len);
    sfgpu_em_opts opts{}; opts.use_vbem = 0; opts.tol = 0.01; opts.min_iter = 50; opts.max_iter = 10000; opts.check_mode = 0; opts.iters_per_launch = 32;
    {
        sfgpu_eq* eq = nullptr; check(sfgpu_eq_create(&eq, 0, nullptr), "create");
        std::vector<uint32_t> ids, off; pack(0, reads.size(), ids, off); build(eq, ids, off);
        Dev d; finish_export(eq, d);
        want.rowptr = d.rowptr.download(); want.ids = d.ids.download(); want.counts = d.counts.download(); want.hashes = d.hashes.download();
        sfgpu_problem pr{}; pr.M = M; pr.d_len = d_len.get(); pr.C = d.C; pr.d_rowptr = d.rowptr.get(); pr.d_ids = d.ids.get(); pr.d_counts = d.counts.get();
        pr.num_mapped = reads.size();
        sfgpu_em* em = nullptr; check(sfgpu_em_create(&em, &pr, nullptr), "em_create");
        DeviceBuf<double> a(M), ms(M);
        check(sfgpu_em_optimize(em, &opts, a.get(), ms.get(), &want_st), "optimize");
        want_alpha = a.download();
        // the RCCL callback of the library itself (sfgpu_comm_*: librccl bound at run time), a communicator of one rank: the
        // sharded loop with ncclAllReduce enqueued on its stream must give what optimize() gives
        if (sfgpu_comm_available()) {
            unsigned char id[SFGPU_COMM_ID_BYTES];
            sfgpu_comm* comm = nullptr;
            check(sfgpu_comm_unique_id(id), "comm_unique_id");
            check(sfgpu_comm_create(&comm, id, 1, 0), "comm_create");
            double us = 0.0;
            DeviceBuf<double> buf(M);
            check(sfgpu_comm_time_allreduce(comm, buf.get(), M, 20, nullptr, &us), "comm_time_allreduce");
            sfgpu_em_stats st1{};
            DeviceBuf<double> a1(M), ms1(M);
            check(sfgpu_em_optimize_sharded(em, &opts, sfgpu_comm_allreduce_fn(), comm, 8, a1.get(), ms1.get(), &st1), "optimize_sharded(rccl)");
            const std::vector<double> got = a1.download();
            bool ok = st1.iters == want_st.iters;
            for (uint32_t t = 0; ok && t < M; ++t) ok = (want_alpha[t] > 0) == (got[t] > 0) && close_to(got[t], want_alpha[t], 1e-9);
            EXPECT(ok);
            std::printf("rccl callback: world 1, all-reduce of %u doubles %.1f us, stop iteration %u\n", M, us, st1.iters);
            sfgpu_comm_destroy(comm);
        } else std::printf("rccl callback: librccl.so not loadable, skipped\n");
        sfgpu_em_destroy(em); sfgpu_eq_destroy(eq);
    }
    Shared sh;
    auto rank_main = [&](int me) {
        try {
            sfgpu_eq *local = nullptr, *part = nullptr;
            check(sfgpu_eq_create(&local, 0, nullptr), "create"); check(sfgpu_eq_create(&part, 0, nullptr), "create");
            std::vector<uint32_t> ids, off; pack(me ? reads.size() / 2 : 0, me ? reads.size() : reads.size() / 2, ids, off); build(local, ids, off);
            Dev d; finish_export(local, d);
            // 1. my classes by owner
            uint64_t hc[2], hi[2], boff[3];
            check(sfgpu_eqvec_owner_sizes(d.rowptr.get(), d.hashes.get(), d.C, 2, hc, hi, nullptr), "owner_sizes");
            DeviceBuf<unsigned char> blocks(SFGPU_BLOCK_BYTES(hc[0], hi[0]) + SFGPU_BLOCK_BYTES(hc[1], hi[1]) + 8);
            check(sfgpu_eqvec_pack_by_owner(d.rowptr.get(), d.ids.get(), d.counts.get(), d.hashes.get(), d.C, 2, hc, hi, blocks.get(), boff, nullptr), "pack");
            for (int dst = 0; dst < 2; ++dst) { sh.blk[me][dst] = blocks.get() + boff[dst]; sh.sz[me][dst][0] = hc[dst]; sh.sz[me][dst][1] = hi[dst]; }
            sh.bar.wait();                                                  // "all-to-all": block [src][me] is mine
            // 2. the owner adds up the copies
            check(sfgpu_eq_start(part), "start");
            for (int src = 0; src < 2; ++src) check(sfgpu_eq_add_block_device(part, sh.blk[src][me], sh.sz[src][me][0], sh.sz[src][me][1], nullptr), "add_block");
            Dev p; finish_export(part, p);
            // 3. my partition of the merged table as one block
            DeviceBuf<unsigned char> pblock(SFGPU_GATHER_BYTES(p.C, p.L) + 8);
            check(sfgpu_eqvec_export_block(p.rowptr.get(), p.ids.get(), p.counts.get(), p.hashes.get(), p.C, p.L, pblock.get(), nullptr), "export_block");
            check_hip(hipDeviceSynchronize(), "sync");
            sh.part[me] = pblock.get(); sh.part_c[me] = p.C; sh.part_l[me] = p.L;
            sh.bar.wait();                                                  // "all-gather"
            // 4. the union, canonical order
            const uint64_t n = sh.part_c[0] + sh.part_c[1], l = sh.part_l[0] + sh.part_l[1];
            Dev m; m.C = n; m.L = l; m.rowptr.resize(n + 1); m.ids.resize(l); m.counts.resize(n); m.hashes.resize(n);
            int tie = -1;
            check(sfgpu_eqvec_merge_disjoint(sh.part, sh.part_c, sh.part_l, 2, m.rowptr.get(), m.ids.get(), m.counts.get(), m.hashes.get(), &tie, nullptr), "merge");
            if (tie != 0) throw std::runtime_error("unexpected key tie");
            sh.merged[me].rowptr = m.rowptr.download(); sh.merged[me].ids = m.ids.download(); sh.merged[me].counts = m.counts.download(); sh.merged[me].hashes = m.hashes.download();
            sh.bar.wait();                                                  // nobody frees a block another rank still reads
            // 5. sharded EM: my half of the classes (cut at half the nonzeros)
            const std::vector<uint32_t>& rp = sh.merged[me].rowptr;
            uint64_t cut = 0; while (cut < n && rp[cut] < l / 2) ++cut;
            const uint64_t c0 = me ? cut : 0, c1 = me ? n : cut;
            std::vector<uint32_t> rp_loc(c1 - c0 + 1); for (uint64_t c = c0; c <= c1; ++c) rp_loc[c - c0] = rp[c] - rp[c0];
            DeviceBuf<uint32_t> d_rp(rp_loc);
            sfgpu_problem pr{}; pr.M = M; pr.d_len = d_len.get(); pr.C = c1 - c0; pr.d_rowptr = d_rp.get(); pr.d_ids = m.ids.get() + rp[c0]; pr.d_counts = m.counts.get() + c0;
            pr.num_mapped = reads.size();
            sfgpu_em* em = nullptr; check(sfgpu_em_create(&em, &pr, nullptr), "em_create");
            DeviceBuf<double> a(M), ms(M);
            ReduceCtx ctx{&sh, me};
            check(sfgpu_em_optimize_sharded(em, &opts, allreduce_two_threads, &ctx, 8, a.get(), ms.get(), &sh.st[me]), "optimize_sharded");
            sh.alpha[me] = a.download();
            sfgpu_em_destroy(em); sfgpu_eq_destroy(local); sfgpu_eq_destroy(part);
        } catch (const std::exception& e) { std::fprintf(stderr, "rank %d: %s\n", me, e.what()); ++failures; }
    };
    std::thread t0(rank_main, 0), t1(rank_main, 1);
    t0.join(); t1.join();
    if (failures == before) {
        for (int r = 0; r < 2; ++r) {
            EXPECT(sh.merged[r].rowptr == want.rowptr && sh.merged[r].ids == want.ids && sh.merged[r].counts == want.counts && sh.merged[r].hashes == want.hashes);
            EXPECT(sh.st[r].iters == want_st.iters && sh.st[r].converged == want_st.converged);
            bool ok = sh.alpha[r].size() == M;
            for (uint32_t t = 0; ok && t < M; ++t) ok = (want_alpha[t] > 0) == (sh.alpha[r][t] > 0) && close_to(sh.alpha[r][t], want_alpha[t], 1e-9);
            EXPECT(ok);
        }
        EXPECT(sh.alpha[0] == sh.alpha[1]);                                 // the all-reduce leaves both ranks with the same bits
        std::printf("two ranks through the ABI: %zu classes merged, sharded EM stopped at iteration %u (single GPU: %u)\n",
                    want.counts.size(), sh.st[0].iters, want_st.iters);
    }
    return failures - before;
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
    failures += two_ranks_on_one_gpu();
    std::printf(failures ? "cpp host FAILED (%d)\n" : "cpp host ok\n", failures);
    return failures ? 1 : 0;
}
