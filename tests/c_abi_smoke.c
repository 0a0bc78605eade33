/* Plain-C host for libsfgpu: proves the boundary is a C ABI usable without Python or torch.
 * Builds a tiny experiment from host buffers (the path a C++ Sailfish host would take):
 * add_batch_host -> finish -> export -> effective lengths -> EM -> TPM, and prints the result.
 * Compiled by tests/test_abi.py (gcc, links libsfgpu + the HIP runtime); run by the GPU tests. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "sfgpu.h"

#define CHECK(x) do { int _rc = (x); if (_rc) { fprintf(stderr, "%s -> %d: %s\n", #x, _rc, sfgpu_last_error()); return 1; } } while (0)
#define HIPCHECK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e)); return 1; } } while (0)

static int n_boot_cb = 0, n_gibbs_cb = 0;
static int on_bootstrap(const double* alpha, uint64_t M, void* user) { (void)user; n_boot_cb += (M == 4 && alpha[0] >= 0.0); return 1; }
static int on_gibbs(const int32_t* counts, uint64_t M, void* user) {
    long tot = 0; for (uint64_t i = 0; i < M; ++i) tot += counts[i];
    n_gibbs_cb += (tot == *(long*)user); return 1;
}

int main(void) {
    /* SURVEY 8c toy: lens [1000,2000,500,1500], classes {0}:100 {0,1}:300 {1,2}:50 {0,1,2}:25 {2}:10 */
    const uint32_t ref_len[4] = {1000, 2000, 500, 1500};
    const uint32_t lab[][3] = {{0}, {0, 1}, {1, 2}, {0, 1, 2}, {2}};
    const uint32_t lab_n[] = {1, 2, 2, 3, 1};
    const uint32_t reps[] = {100, 300, 50, 25, 10};
    uint32_t ids[2048], off[512]; uint32_t n = 0, h = 0;
    for (int c = 0; c < 5; ++c) for (uint32_t r = 0; r < reps[c]; ++r) { off[n++] = h; for (uint32_t k = 0; k < lab_n[c]; ++k) ids[h++] = lab[c][k]; }
    off[n] = h;

    sfgpu_eq* eq = NULL;
    CHECK(sfgpu_eq_create(&eq, 0, NULL));
    CHECK(sfgpu_eq_start(eq));
    CHECK(sfgpu_eq_add_batch_host(eq, ids, off, n));
    uint64_t C, L, total;
    CHECK(sfgpu_eq_finish(eq, &C, &L, &total));
    if (C != 5 || total != 485) { fprintf(stderr, "classes %llu total %llu\n", (unsigned long long)C, (unsigned long long)total); return 1; }

    uint32_t *d_rowptr, *d_ids, *d_ref; uint64_t* d_counts; double *d_eff, *d_alpha, *d_mass, *d_tpm;
    HIPCHECK(hipMalloc((void**)&d_rowptr, (C + 1) * 4)); HIPCHECK(hipMalloc((void**)&d_ids, L * 4));
    HIPCHECK(hipMalloc((void**)&d_counts, C * 8)); HIPCHECK(hipMalloc((void**)&d_ref, 16));
    HIPCHECK(hipMalloc((void**)&d_eff, 32)); HIPCHECK(hipMalloc((void**)&d_alpha, 32));
    HIPCHECK(hipMalloc((void**)&d_mass, 32)); HIPCHECK(hipMalloc((void**)&d_tpm, 32));
    CHECK(sfgpu_eq_export_device(eq, d_rowptr, d_ids, d_counts, NULL));
    HIPCHECK(hipMemcpy(d_ref, ref_len, 16, hipMemcpyHostToDevice));
    /* the KAT uses EffectiveLength = len - 199: feed it as a 1-entry "table" via direct lengths */
    double eff[4]; for (int i = 0; i < 4; ++i) eff[i] = ref_len[i] - 199.0;
    HIPCHECK(hipMemcpy(d_eff, eff, 32, hipMemcpyHostToDevice));

    sfgpu_problem prob = {4, d_eff, C, d_rowptr, d_ids, d_counts, total};
    sfgpu_em* em = NULL;
    CHECK(sfgpu_em_create(&em, &prob, NULL));
    sfgpu_em_opts o = {0, 0.01, 50, 10000, 0, 0};
    sfgpu_em_stats st;
    CHECK(sfgpu_em_optimize(em, &o, d_alpha, d_mass, &st));
    CHECK(sfgpu_tpm(d_alpha, d_eff, 4, (double)total, d_tpm, NULL));
    double alpha[4], tpm[4];
    HIPCHECK(hipMemcpy(alpha, d_alpha, 32, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(tpm, d_tpm, 32, hipMemcpyDeviceToHost));
    printf("iters %u alpha %.17g %.17g %.17g %.17g tpm_sum %.6f\n", st.iters, alpha[0], alpha[1], alpha[2], alpha[3],
           tpm[0] + tpm[1] + tpm[2] + tpm[3]);

    /* the rest of the ABI from C: posterior samplers with their writer hooks, the effective-length helpers and
     * the hit-filtering stage in front of the path */
    CHECK(sfgpu_bootstrap(em, &o, 3, 7, NULL, on_bootstrap, NULL, NULL));
    long n_frags = (long)total;
    CHECK(sfgpu_gibbs_sample(&prob, d_mass, 4, 0, 7, NULL, on_gibbs, &n_frags, NULL));
    double cf[1000];
    CHECK(sfgpu_cf_gaussian(1000, 200, 80, cf));
    CHECK(sfgpu_efflen_smoothed(d_ref, 4, cf, 1000, d_eff, NULL));
    HIPCHECK(hipMemcpy(eff, d_eff, 32, hipMemcpyDeviceToHost));
    uint32_t flc[1000]; memset(flc, 0, sizeof flc); flc[180] = 30; flc[220] = 50; flc[400] = 20;
    CHECK(sfgpu_efflen_empirical(flc, 1000, d_ref, 4, d_alpha /* reuse */, NULL));
    double eff_emp[4]; HIPCHECK(hipMemcpy(eff_emp, d_alpha, 32, hipMemcpyDeviceToHost));
    sfgpu_hit recs[4] = {{3, 10, 200, 240, 50, 50, 1, 0, 3, 0}, {9, 10, 200, 240, 50, 50, 0, 1, 3, 0},   /* read 0: ISF keeps tid 3 */
                         {4, 10, 200, 260, 50, 50, 1, 0, 3, 0},                                          /* read 1: unique pair */
                         {6, 0, 0, 0, 50, 50, 1, 0, 1, 0}};                                              /* read 2: an orphan, dropped */
    uint32_t roff[4] = {0, 2, 3, 4};
    sfgpu_hit* d_recs; uint32_t *d_roff, *d_fids, *d_foff, *d_fl;
    HIPCHECK(hipMalloc((void**)&d_recs, sizeof recs)); HIPCHECK(hipMalloc((void**)&d_roff, 16));
    HIPCHECK(hipMalloc((void**)&d_fids, 16)); HIPCHECK(hipMalloc((void**)&d_foff, 16)); HIPCHECK(hipMalloc((void**)&d_fl, 4000));
    HIPCHECK(hipMemcpy(d_recs, recs, sizeof recs, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(d_roff, roff, 16, hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(d_fl, 0, 4000));
    sfgpu_filter_opts fo = {200, 1000, 1, 1, 0, 0, 0, {1, 2, 0, 0}};       /* paired library, orphans discarded, ISF */
    sfgpu_filter_stats fs; memset(&fs, 0, sizeof fs);
    int64_t budget = 10;
    CHECK(sfgpu_filter_hits(d_recs, d_roff, 3, &fo, d_fids, d_foff, d_fl, &budget, &fs, NULL));
    uint32_t fids[4], foff[4];
    HIPCHECK(hipMemcpy(fids, d_fids, 16, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(foff, d_foff, 16, hipMemcpyDeviceToHost));
    /* bias-aware effective lengths: a handle over the (synthetic) transcript sequences, one
     * updateEffectiveLengths call per model, and optimize() with the recompute hook */
    static char seq[5004]; uint64_t soff[4]; uint64_t sp = 0; uint32_t lcg = 12345u;
    for (int t = 0; t < 4; ++t) { soff[t] = sp; for (uint32_t i = 0; i < ref_len[t]; ++i) { lcg = lcg * 1664525u + 1013904223u; seq[sp++] = "ACGT"[lcg >> 30]; } seq[sp++] = '$'; }
    char* d_seq; uint64_t* d_soff; double *d_teff, *d_bout;
    HIPCHECK(hipMalloc((void**)&d_seq, sizeof seq)); HIPCHECK(hipMalloc((void**)&d_soff, 32));
    HIPCHECK(hipMalloc((void**)&d_teff, 32)); HIPCHECK(hipMalloc((void**)&d_bout, 32));
    HIPCHECK(hipMemcpy(d_seq, seq, sizeof seq, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(d_soff, soff, 32, hipMemcpyHostToDevice));
    for (int i = 0; i < 4; ++i) eff[i] = ref_len[i] - 199.0;
    HIPCHECK(hipMemcpy(d_teff, eff, 32, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_eff, eff, 32, hipMemcpyHostToDevice));            /* the problem's lengths again */
    uint32_t fld[1000], rbias[4096], ogc[101];
    for (int i = 0; i < 1000; ++i) { int dd = i - 200; fld[i] = (dd > -150 && dd < 150) ? (uint32_t)(150 - (dd < 0 ? -dd : dd)) : 0; }
    for (int i = 0; i < 4096; ++i) rbias[i] = 1 + (uint32_t)(i % 7);
    for (int i = 0; i < 101; ++i) ogc[i] = 1 + (uint32_t)(i > 30 && i < 70 ? 50 : 0);
    int bias_ok = 1;
    for (int model = 0; model < 2; ++model) {
        sfgpu_bias_inputs bi; memset(&bi, 0, sizeof bi);
        bi.M = 4; bi.d_seq = d_seq; bi.d_seq_off = d_soff; bi.d_ref_len = d_ref; bi.d_txp_eff_len = d_teff;
        bi.h_fl_counts = fld; bi.max_frag_len = 1000; bi.gc_speed_samp = 1; bi.h_read_bias = rbias; bi.h_observed_gc = ogc;
        bi.num_fwd = 60; bi.num_rc = 40; bi.seq_bias = model == 0; bi.gc_bias = model == 1; bi.gc_size_samp = 1;
        sfgpu_bias* bias = NULL; sfgpu_bias_stats bs;
        CHECK(sfgpu_bias_create(&bias, &bi, NULL));
        CHECK(sfgpu_bias_update(bias, d_teff, d_alpha, d_bout, &bs, NULL));  /* d_alpha: leftovers, any finite values do */
        double es[4096], eg[101], beff[4]; uint32_t hooks = 0;
        CHECK(sfgpu_em_optimize_bias(em, &o, bias, d_alpha, d_mass, d_bout, &hooks, &st));
        CHECK(sfgpu_bias_expected(bias, es, eg));
        HIPCHECK(hipMemcpy(beff, d_bout, 32, hipMemcpyDeviceToHost));
        bias_ok = bias_ok && bs.status == 0 && bs.n_corrected + bs.n_uncorrected == 4 && st.iters >= 50 && hooks == (uint32_t)((st.iters > 50) + (st.iters > 500) + (st.iters > 1000)) &&
                  beff[0] > 1.0 && beff[1] > 1.0 && es[0] >= 1.0 && eg[50] >= 1.0;
        printf("bias model %d: iters %u hooks %u effLen %.3f %.3f %.3f %.3f\n", model, st.iters, hooks, beff[0], beff[1], beff[2], beff[3]);
        CHECK(sfgpu_bias_destroy(bias));
    }
    /* the samples the hit loop collects for the bias models, and the piecewise form of the recompute hook */
    uint32_t *d_gcp, *d_rb, *d_og;
    HIPCHECK(hipMalloc((void**)&d_gcp, sizeof seq * 4)); HIPCHECK(hipMalloc((void**)&d_rb, 4096 * 4)); HIPCHECK(hipMalloc((void**)&d_og, 101 * 4));
    HIPCHECK(hipMemset(d_rb, 0, 4096 * 4)); HIPCHECK(hipMemset(d_og, 0, 101 * 4));
    CHECK(sfgpu_gc_prefix(d_seq, d_soff, d_ref, 4, d_gcp, NULL));
    sfgpu_hit srec[2] = {{0, 10, 160, 200, 50, 50, 1, 0, 3, 0}, {1, 300, 450, 200, 50, 50, 0, 1, 3, 0}};   /* two proper pairs */
    uint32_t sroff[3] = {0, 1, 2};
    HIPCHECK(hipMemcpy(d_recs, srec, sizeof srec, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(d_roff, sroff, 12, hipMemcpyHostToDevice));
    int64_t bias_budget = 5;
    sfgpu_bias_sampler smp; memset(&smp, 0, sizeof smp);
    smp.d_seq = d_seq; smp.d_seq_off = d_soff; smp.d_ref_len = d_ref; smp.d_read_bias = d_rb; smp.remaining_bias_samples = &bias_budget;
    smp.d_observed_gc = d_og; smp.d_gc_prefix = d_gcp; smp.gc_size_samp = 1;
    CHECK(sfgpu_sample_bias(d_recs, d_roff, 2, &fo, &smp, NULL));
    int sample_ok = smp.n_bias_sampled == 2 && smp.n_gc_sampled == 2 && bias_budget == 3;
    CHECK(sfgpu_em_begin(em, &o)); CHECK(sfgpu_em_init(em));
    CHECK(sfgpu_em_set_bounds(em, 3, 3));
    for (int i = 0; i < 4; ++i) { CHECK(sfgpu_em_sweep(em)); CHECK(sfgpu_em_update(em)); }   /* the fourth pair is a no-op: stopped at 3 */
    int em_done = 0; CHECK(sfgpu_em_poll(em, &em_done, &st));
    CHECK(sfgpu_em_rebase(em, sfgpu_em_lengths(em)));                     /* same lengths: x is rebuilt from alpha */
    CHECK(sfgpu_em_set_bounds(em, 5, 5));
    for (int i = 0; i < 2; ++i) { CHECK(sfgpu_em_sweep(em)); CHECK(sfgpu_em_update(em)); }
    int em_done2 = 0; sfgpu_em_stats st2; CHECK(sfgpu_em_poll(em, &em_done2, &st2));
    CHECK(sfgpu_em_finish(em, d_alpha, d_mass, &st2));
    int piecewise_ok = em_done == 1 && st.iters == 3 && em_done2 == 1 && st2.iters == 5 && sfgpu_em_alpha(em) != NULL;
    printf("samples %s (6-mers %llu, fragments %llu)  piecewise %s (iters %u then %u)\n", sample_ok ? "ok" : "FAILED",
           (unsigned long long)smp.n_bias_sampled, (unsigned long long)smp.n_gc_sampled, piecewise_ok ? "ok" : "FAILED", st.iters, st2.iters);
    int extras_ok = sample_ok && piecewise_ok && bias_ok && n_boot_cb == 3 && n_gibbs_cb == 4 && eff[0] > 790.0 && eff[0] < 810.0 && eff_emp[1] > 1700.0 && eff_emp[1] < 1800.0 &&
                    foff[0] == 0 && foff[1] == 1 && foff[2] == 2 && foff[3] == 2 && fids[0] == 3 && fids[1] == 4 &&
                    fs.n_observed == 3 && fs.n_mapped == 2 && fs.fl_sampled == 1 && budget == 9;
    printf("extras %s (boot %d gibbs %d eff %.3f emp %.3f mapped %llu)\n", extras_ok ? "ok" : "FAILED", n_boot_cb, n_gibbs_cb, eff[0], eff_emp[1],
           (unsigned long long)fs.n_mapped);
    sfgpu_em_destroy(em); sfgpu_eq_destroy(eq);
    return extras_ok ? 0 : 2;
}
