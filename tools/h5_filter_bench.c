/* H5Dwrite / H5Dread through filter 32025 timed INSIDE one process (file on tmpfs), in the units of the reference's
 * docs/Performance.md:10,24-25 (MB/s of uncompressed data): process start-up, HIP initialisation and the plugin load
 * are paid by an untimed warm-up dataset first, so the figures are what a long-running writer/reader sees.
 * The filter is found through HDF5_PLUGIN_PATH (deltarice_amd/plugin) like any dynamically loaded filter.
 * build: gcc -O2 tools/h5_filter_bench.c -o /tmp/h5_filter_bench -I$HDF5/include -L$HDF5/lib -lhdf5 -Wl,-rpath,$HDF5/lib
 * usage: h5_filter_bench file.h5 [rows=20000 cols=7000 chunk_rows=2000 M=8 L=cols reps=5] */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CHECK(x) do { if ((x) < 0) { fprintf(stderr, "HDF5 call failed: %s (line %d)\n", #x, __LINE__); return 2; } } while (0)

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int cmp_d(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }

static int write_ds(const char *path, const char *name, const short *x, hsize_t rows, hsize_t cols, hsize_t crows,
                    unsigned M, unsigned L, double *ms, hsize_t *stored) {
    hid_t file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(file);
    hsize_t dims[2] = {rows, cols}, chunk[2] = {crows, cols};
    hid_t space = H5Screate_simple(2, dims, NULL), dcpl = H5Pcreate(H5P_DATASET_CREATE);
    const unsigned cd[2] = {M, L};
    CHECK(H5Pset_chunk(dcpl, 2, chunk));
    CHECK(H5Pset_filter(dcpl, 32025, H5Z_FLAG_MANDATORY, 2, cd));
    hid_t dset = H5Dcreate(file, name, H5T_NATIVE_SHORT, space, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    CHECK(dset);
    const double t0 = now_ms();
    CHECK(H5Dwrite(dset, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, x));
    CHECK(H5Dclose(dset));   /* flushes the chunk cache through the filter */
    CHECK(H5Fflush(file, H5F_SCOPE_GLOBAL));
    *ms = now_ms() - t0;
    dset = H5Dopen(file, name, H5P_DEFAULT);
    *stored = H5Dget_storage_size(dset);
    H5Dclose(dset); H5Pclose(dcpl); H5Sclose(space);
    CHECK(H5Fclose(file));
    return 0;
}

static int read_ds(const char *path, const char *name, short *y, double *ms) {
    hid_t file = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    CHECK(file);
    hid_t dset = H5Dopen(file, name, H5P_DEFAULT);
    CHECK(dset);
    const double t0 = now_ms();
    CHECK(H5Dread(dset, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, y));
    *ms = now_ms() - t0;
    H5Dclose(dset);
    CHECK(H5Fclose(file));
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const hsize_t rows = argc > 2 ? strtoull(argv[2], 0, 10) : 20000, cols = argc > 3 ? strtoull(argv[3], 0, 10) : 7000,
                  crows = argc > 4 ? strtoull(argv[4], 0, 10) : 2000;
    const unsigned M = argc > 5 ? (unsigned)atoi(argv[5]) : 8, L = argc > 6 ? (unsigned)atoi(argv[6]) : (unsigned)cols;
    const int reps = argc > 7 ? atoi(argv[7]) : 5;
    const size_t n = rows * cols;
    short *x = malloc(n * 2), *y = malloc(n * 2);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {  /* sum of 12 uniforms: roughly Gaussian, sigma 10 */
        double a = 0;
        for (int j = 0; j < 12; ++j) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a += (double)(s >> 11) / 9007199254740992.0; }
        x[i] = (short)((a - 6.0) * 10.0);
    }
    double ms;
    hsize_t stored = 0;
    /* warm-up: plugin load, HIP context, the codec's plan and staging buffers (one chunk's worth) */
    if (write_ds(argv[1], "warm", x, crows, cols, crows, M, L, &ms, &stored)) return 2;
    if (read_ds(argv[1], "warm", y, &ms)) return 2;
    double tw[64], tr[64];
    for (int r = 0; r < reps && r < 64; ++r) {
        if (write_ds(argv[1], "test", x, rows, cols, crows, M, L, &tw[r], &stored)) return 2;
        memset(y, 0, n * 2);
        if (read_ds(argv[1], "test", y, &tr[r])) return 2;
        if (memcmp(x, y, n * 2)) { fprintf(stderr, "round trip mismatch\n"); return 3; }
    }
    qsort(tw, reps, sizeof *tw, cmp_d);
    qsort(tr, reps, sizeof *tr, cmp_d);
    const double mb = n * 2 / 1e6;
    printf("H5Dwrite/H5Dread through filter 32025, in-process, %llu x %llu int16 = %.0f MB, chunks %llu x %llu, cd = {%u, %u}, "
           "stored %.1f %%: write %.1f ms = %.0f MB/s, read %.1f ms = %.0f MB/s (median of %d)\n",
           (unsigned long long)rows, (unsigned long long)cols, mb, (unsigned long long)crows, (unsigned long long)cols, M, L,
           100.0 * stored / (n * 2.0), tw[reps / 2], mb / tw[reps / 2] * 1e3, tr[reps / 2], mb / tr[reps / 2] * 1e3, reps);
    free(x); free(y);
    return 0;
}
