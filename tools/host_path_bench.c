/* Throughput of the one-chunk host path from C, the way libhdf5 drives the H5Z callback: malloc'd buffer
 * in, malloc'd buffer out, one chunk of W x L int16 per call (default shapes: what one H5Z call sees for the
 * reference's README example 20 x 7000, BASELINE config #1's 100 x 7000, the Nab chunk 2000 x 7000 and the nEDM
 * chunk 32 x 81920, docs/Performance.md:16,27).  Also the bare PCIe time of the same bytes (pageable host memory,
 * one hipMemcpy each way) so that the codec's share of a call can be read off.
 * build: gcc -O2 -Iinclude -I/opt/rocm/include tools/host_path_bench.c -o /tmp/host_path_bench -Ldeltarice_amd -ldeltarice_hip \
 *        -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/deltarice_amd -Wl,-rpath,/opt/rocm/lib -lm -D__HIP_PLATFORM_AMD__
 * usage: host_path_bench [W L]... */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <hip/hip_runtime_api.h>
#include "deltarice_hip.h"

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int cmp_d(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }
static double median(double *v, int n) { qsort(v, n, sizeof *v, cmp_d); return v[n / 2]; }

static int run(drx_ctx *ctx, size_t W, size_t L) {
    const size_t n = W * L;
    int16_t *x = malloc(n * 2);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {  /* sum of 12 uniforms: roughly Gaussian, sigma 10 */
        double a = 0;
        for (int j = 0; j < 12; ++j) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a += (double)(s >> 11) / 9007199254740992.0; }
        x[i] = (int16_t)((a - 6.0) * 10.0);
    }
    const unsigned cd[2] = {8, (unsigned)L};
    void *enc = NULL, *dec = NULL;
    size_t enc_bytes = 0, dec_bytes = 0;
    enum { REPS = 15 };
    double te[REPS], td[REPS], th2d[REPS], td2h[REPS];
    void *dbuf = NULL;
    if (hipMalloc(&dbuf, n * 2) != hipSuccess) return 1;
    for (int r = 0; r < REPS + 2; ++r) {
        void *in = malloc(n * 2);  /* HDF5 hands the filter a buffer it allocated */
        memcpy(in, x, n * 2);
        double t0 = now_ms();
        if (drx_filter_chunk_host(ctx, 0, 2, cd, in, n * 2, &enc, &enc_bytes) != DRX_OK) { fprintf(stderr, "encode: %s\n", drx_ctx_last_error(ctx)); return 1; }
        double t1 = now_ms();
        free(in);
        if (drx_filter_chunk_host(ctx, 1, 2, cd, enc, enc_bytes, &dec, &dec_bytes) != DRX_OK) { fprintf(stderr, "decode: %s\n", drx_ctx_last_error(ctx)); return 1; }
        double t2 = now_ms();
        if (dec_bytes != n * 2 || memcmp(dec, x, n * 2)) { fprintf(stderr, "round trip mismatch\n"); return 1; }
        /* the same bytes across PCIe and nothing else: raw chunk up, raw chunk down (pageable memory, as the callback has) */
        double t3 = now_ms();
        if (hipMemcpy(dbuf, x, n * 2, hipMemcpyHostToDevice) != hipSuccess) return 1;
        double t4 = now_ms();
        if (hipMemcpy(dec, dbuf, n * 2, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        double t5 = now_ms();
        free(enc); free(dec);
        if (r >= 2) { te[r - 2] = t1 - t0; td[r - 2] = t2 - t1; th2d[r - 2] = t4 - t3; td2h[r - 2] = t5 - t4; }
    }
    (void)hipFree(dbuf);
    const double e = median(te, REPS), d = median(td, REPS), up = median(th2d, REPS), down = median(td2h, REPS);
    const double ratio = (double)enc_bytes / (n * 2);
    /* what crosses PCIe in a call: encode = raw up + encoded down, decode = encoded up + raw down */
    printf("chunk %5zu x %6zu (%7.2f MB, ratio %.4f): encode %7.3f ms = %6.2f GB/s | decode %7.3f ms = %6.2f GB/s | "
           "bare copies of the raw chunk: H2D %6.3f ms, D2H %6.3f ms -> encode's copies ~%.3f ms, decode's ~%.3f ms\n",
           W, L, n * 2 / 1e6, ratio, e, n * 2 / e / 1e6, d, n * 2 / d / 1e6, up, down, up + ratio * down, ratio * up + down);
    free(x);
    return 0;
}

int main(int argc, char **argv) {
    drx_ctx *ctx = NULL;
    if (drx_ctx_create(0, NULL, &ctx) != DRX_OK) { fprintf(stderr, "no GPU\n"); return 1; }
    static const size_t def[][2] = {{20, 7000}, {100, 7000}, {2000, 7000}, {32, 81920}};
    int rc = 0;
    if (argc >= 3) {
        for (int i = 1; i + 1 < argc && !rc; i += 2) rc = run(ctx, strtoull(argv[i], 0, 10), strtoull(argv[i + 1], 0, 10));
    } else {
        for (size_t i = 0; i < sizeof def / sizeof def[0] && !rc; ++i) rc = run(ctx, def[i][0], def[i][1]);
    }
    drx_ctx_destroy(ctx);
    return rc;
}
