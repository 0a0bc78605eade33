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
    sfgpu_em_destroy(em); sfgpu_eq_destroy(eq);
    return 0;
}
