/* dev probe: throughput of the HOST entry point of the builder -- T mapper-like threads hand over
 * 1000-read batches through sfgpu_eq_add_batch_host (the call the reference-side adaptor of
 * INTEGRATION.md makes), then finish().
 * gcc -O2 -pthread tools/host_path_probe.c -Iinclude -Lsailfish_amd/csrc -lsfgpu -o tools/host_path_probe */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "sfgpu.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct { sfgpu_eq* eq; const uint32_t* ids; const uint32_t* off; uint32_t r0, r1, batch; } job_t;

static void* worker(void* p) {
    job_t* j = (job_t*)p;
    for (uint32_t r = j->r0; r < j->r1; r += j->batch) {
        uint32_t n = (j->r1 - r < j->batch) ? j->r1 - r : j->batch;
        /* offsets are passed as they are (base != 0): the library rebases them */
        if (sfgpu_eq_add_batch_host(j->eq, j->ids, j->off + r, n) != SFGPU_OK) { fprintf(stderr, "%s\n", sfgpu_last_error()); exit(1); }
    }
    return NULL;
}

int main(int argc, char** argv) {
    const uint32_t R = argc > 1 ? (uint32_t)atol(argv[1]) : 20000000u;
    const int T = argc > 2 ? atoi(argv[2]) : 8;
    const uint32_t batch = argc > 3 ? (uint32_t)atol(argv[3]) : 1000u;
    const uint32_t P = 400000, M = 80000;
    /* label pool: P labels of 1..8 nearby ids; reads pick labels */
    uint32_t* plen = malloc(P * 4); uint32_t* pbase = malloc(P * 4);
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = 0; i < P; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; plen[i] = 1 + (uint32_t)(s % 8); pbase[i] = (uint32_t)((s >> 20) % (M - 64)); }
    uint32_t* off = malloc(((size_t)R + 1) * 4); uint32_t* ids = malloc((size_t)R * 8 * 4);
    uint64_t h = 0;
    for (uint32_t r = 0; r < R; ++r) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t a = (uint32_t)(s % P), b = (uint32_t)((s >> 32) % P), l = a < b ? a : b;
        off[r] = (uint32_t)h;
        for (uint32_t k = 0; k < plen[l]; ++k) ids[h++] = pbase[l] + 7 * k;
    }
    off[R] = (uint32_t)h;
    sfgpu_eq* eq = NULL;
    if (sfgpu_eq_create(&eq, 0, NULL) != SFGPU_OK) { fprintf(stderr, "%s\n", sfgpu_last_error()); return 1; }
    for (int rep = 0; rep < 3; ++rep) {
        sfgpu_eq_start(eq);
        double t0 = now();
        pthread_t th[64]; job_t jb[64];
        for (int t = 0; t < T; ++t) {
            jb[t] = (job_t){eq, ids, off, (uint32_t)((uint64_t)R * t / T), (uint32_t)((uint64_t)R * (t + 1) / T), batch};
            pthread_create(&th[t], NULL, worker, &jb[t]);
        }
        for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
        uint64_t nc, nnz, tot;
        if (sfgpu_eq_finish(eq, &nc, &nnz, &tot) != SFGPU_OK) { fprintf(stderr, "%s\n", sfgpu_last_error()); return 1; }
        double dt = now() - t0;
        printf("host path: %u reads (%llu ids) in %d threads x %u-read batches: %.1f ms = %.1f M reads/s, %.2f GB/s; %llu classes, %llu reads counted\n",
               R, (unsigned long long)h, T, batch, dt * 1e3, R / dt / 1e6, (h * 4.0 + R * 4.0) / dt / 1e9,
               (unsigned long long)nc, (unsigned long long)tot);
    }
    sfgpu_eq_destroy(eq);
    return 0;
}
