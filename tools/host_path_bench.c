/* Throughput of the one-chunk host path from C, the way libhdf5 drives the H5Z callback: malloc'd buffer
 * in, malloc'd buffer out, one 2000 x 7000 int16 chunk per call.
 * build: gcc -O2 -Iinclude tools/host_path_bench.c -o /tmp/host_path_bench -Ldeltarice_amd -ldeltarice_hip -Wl,-rpath,$PWD/deltarice_amd -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "deltarice_hip.h"

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

int main(void) {
    const size_t W = 2000, L = 7000, n = W * L;
    int16_t *x = malloc(n * 2);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {  /* sum of 12 uniforms: roughly Gaussian, sigma 10 */
        double a = 0;
        for (int j = 0; j < 12; ++j) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a += (double)(s >> 11) / 9007199254740992.0; }
        x[i] = (int16_t)((a - 6.0) * 10.0);
    }
    drx_ctx *ctx = NULL;
    if (drx_ctx_create(0, NULL, &ctx) != DRX_OK) { fprintf(stderr, "no GPU\n"); return 1; }
    const unsigned cd[2] = {8, (unsigned)L};
    void *enc = NULL, *dec = NULL;
    size_t enc_bytes = 0, dec_bytes = 0;
    double te = 0, td = 0;
    const int reps = 10;
    for (int r = 0; r < reps + 2; ++r) {
        void *in = malloc(n * 2);  /* HDF5 hands the filter a buffer it allocated */
        memcpy(in, x, n * 2);
        double t0 = now_ms();
        if (drx_filter_chunk_host(ctx, 0, 2, cd, in, n * 2, &enc, &enc_bytes) != DRX_OK) { fprintf(stderr, "encode: %s\n", drx_ctx_last_error(ctx)); return 1; }
        double t1 = now_ms();
        free(in);
        if (drx_filter_chunk_host(ctx, 1, 2, cd, enc, enc_bytes, &dec, &dec_bytes) != DRX_OK) { fprintf(stderr, "decode: %s\n", drx_ctx_last_error(ctx)); return 1; }
        double t2 = now_ms();
        if (dec_bytes != n * 2 || memcmp(dec, x, n * 2)) { fprintf(stderr, "round trip mismatch\n"); return 1; }
        free(enc); free(dec);
        if (r >= 2) { te += t1 - t0; td += t2 - t1; }
    }
    printf("host path, one 2000x7000 chunk per call (ratio %.4f): encode %.3f ms = %.2f GB/s, decode %.3f ms = %.2f GB/s\n",
           (double)enc_bytes / (n * 2), te / reps, n * 2 / (te / reps) / 1e6, td / reps, n * 2 / (td / reps) / 1e6);
    drx_ctx_destroy(ctx);
    return 0;
}
