/*
 * h5_direct.c -- direct-chunk HDF5 file <-> VRAM path (include/deltarice_h5io.h).
 * Host-side caller of the hot path: HDF5 chunk I/O + one PCIe copy + one batched drx_* call.
 */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "deltarice_h5io.h"

#define FILTER_ID 32025

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* H5Dcreate refuses a MANDATORY filter it cannot find, although H5Dwrite_chunk never runs it: make
 * filter 32025 known to this process' HDF5 by registering the plugin that sits next to this library. */
static void ensure_filter_registered(void) {
    static int done;
    if (done || H5Zfilter_avail(FILTER_ID) > 0) { done = 1; return; }
    Dl_info info;
    if (dladdr((void *)&ensure_filter_registered, &info) && info.dli_fname) {
        char path[4096];
        snprintf(path, sizeof path, "%s", info.dli_fname);
        char *slash = strrchr(path, '/');
        if (slash) {
            snprintf(slash + 1, sizeof path - (size_t)(slash + 1 - path), "plugin/libh5deltarice.so");
            void *h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                /* register through THIS library's libhdf5 (the plugin itself does not link one) */
                const void *(*info_fn)(void) = (const void *(*)(void))dlsym(h, "H5PLget_plugin_info");
                if (info_fn) (void)H5Zregister(info_fn());
            }
        }
    }
    done = 1;
}

/* The raw hipMalloc / hipMemcpy calls below must land on the context's device whatever device the calling thread has
 * current (include/deltarice_hip.h: every call runs on the context's device and restores the caller's). */
static int enter_device(const drx_ctx *ctx, int *prev) {
    const int want = drx_ctx_device(ctx);
    if (hipGetDevice(prev) != hipSuccess) *prev = -1;
    if (*prev == want) { *prev = -1; return 0; }  /* nothing to restore */
    return hipSetDevice(want) == hipSuccess ? 0 : -1;
}
static void leave_device(int prev) { if (prev >= 0) (void)hipSetDevice(prev); }

static int log2_m(unsigned m, unsigned *k) {
    if (m == 0 || (m & (m - 1)) || m > 32768) return -1;
    *k = 0;
    while ((1u << *k) != m) ++*k;
    return 0;
}

drx_status drx_h5_read(drx_ctx *ctx, const char *file, const char *name, int16_t *d_out,
                       uint64_t out_cap_samples, drx_h5_stats *st) {
    if (!ctx || !file || !name || !d_out) return DRX_ERR_ARG;
    drx_h5_stats s;
    memset(&s, 0, sizeof s);
    drx_status rc = DRX_ERR_ARG;
    hid_t f = -1, d = -1, sp = -1, pl = -1;
    void *h_words = NULL, *d_words = NULL;
    uint64_t *h_off = NULL, *d_off = NULL;
    drx_plan *plan = NULL, *plan_edge = NULL;
    void *d_edge = NULL;
    double t0 = now();
    int prev_dev = -1;
    if (enter_device(ctx, &prev_dev) != 0) return DRX_ERR_DEVICE;

    if ((f = H5Fopen(file, H5F_ACC_RDONLY, H5P_DEFAULT)) < 0) goto out;
    if ((d = H5Dopen2(f, name, H5P_DEFAULT)) < 0) goto out;
    sp = H5Dget_space(d);
    pl = H5Dget_create_plist(d);
    hsize_t dims[2], chunk[2];
    if (H5Sget_simple_extent_ndims(sp) != 2 || H5Sget_simple_extent_dims(sp, dims, NULL) < 0) goto out;
    if (H5Pget_chunk(pl, 2, chunk) != 2 || chunk[1] != dims[1]) { rc = DRX_ERR_UNSUPPORTED; goto out; }
    {
        hid_t ty = H5Dget_type(d);
        /* 16-bit integers in the byte order the codec computes in; signed or unsigned alike (the reference reinterprets the
         * bytes as int16 whatever the type, tests/test.py:72-83) */
        const int ok = H5Tget_class(ty) == H5T_INTEGER && H5Tget_size(ty) == 2 && H5Tget_order(ty) == H5T_ORDER_LE;
        H5Tclose(ty);
        if (!ok) { rc = DRX_ERR_UNSUPPORTED; goto out; }
    }
    unsigned cd[3 + DRX_MAX_TAPS], flags = 0, fcfg = 0;
    size_t ncd = 3 + DRX_MAX_TAPS;
    char fname[8];
    if (H5Pget_nfilters(pl) != 1 ||
        H5Pget_filter_by_id2(pl, FILTER_ID, &flags, &ncd, cd, sizeof fname, fname, &fcfg) < 0) {
        rc = DRX_ERR_UNSUPPORTED;  /* other filters in the pipeline */
        goto out;
    }
    drx_opts o;
    if (drx_parse_cd_values(ncd, cd, &o) != DRX_OK) goto out;
    const unsigned k = o.rice_k, L = o.wave_len < 0 ? 0u : (unsigned)o.wave_len;
    /* HDF5 stores the last chunk full size when the rows do not divide: it is decoded into scratch and the
     * rows that exist are copied out */
    const uint64_t n_full = dims[0] / chunk[0], edge_rows = dims[0] % chunk[0];
    s.rows = dims[0]; s.cols = dims[1]; s.chunk_rows = chunk[0]; s.n_chunks = n_full + (edge_rows ? 1 : 0);
    s.raw_bytes = dims[0] * dims[1] * 2;
    if (dims[0] * dims[1] > out_cap_samples) { rc = DRX_ERR_CAPACITY; goto out; }
    if (chunk[0] * chunk[1] > 0x7fffffffull) goto out;

    /* sizes -> offsets -> one pinned buffer -> raw chunk reads */
    h_off = (uint64_t *)malloc((s.n_chunks + 1) * sizeof(uint64_t));
    if (!h_off) { rc = DRX_ERR_NOMEM; goto out; }
    uint64_t words = 0;
    for (uint64_t c = 0; c < s.n_chunks; ++c) {
        hsize_t off[2] = {c * chunk[0], 0}, nb = 0;
        if (H5Dget_chunk_storage_size(d, off, &nb) < 0 || (nb & 3)) { rc = DRX_ERR_CORRUPT; goto out; }
        if (nb == 0) { rc = DRX_ERR_UNSUPPORTED; goto out; }  /* a chunk that was never written (fill value): not a stored stream */
        h_off[c] = words;
        words += nb / 4;
    }
    h_off[s.n_chunks] = words;
    s.stored_bytes = words * 4;
    if (hipHostMalloc(&h_words, words * 4, hipHostMallocDefault) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
    for (uint64_t c = 0; c < s.n_chunks; ++c) {
        hsize_t off[2] = {c * chunk[0], 0};
        uint32_t mask = 0;
        if (H5Dread_chunk(d, H5P_DEFAULT, off, &mask, (uint32_t *)h_words + h_off[c]) < 0 || mask) { rc = DRX_ERR_CORRUPT; goto out; }
    }
    s.t_file = now() - t0;

    t0 = now();
    hipStream_t stream = (hipStream_t)drx_ctx_stream(ctx);
    if (hipMalloc(&d_words, words * 4) != hipSuccess || hipMalloc((void **)&d_off, (s.n_chunks + 1) * 8) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
    if (hipMemcpyAsync(d_words, h_words, words * 4, hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(d_off, h_off, (s.n_chunks + 1) * 8, hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) { rc = DRX_ERR_DEVICE; goto out; }
    s.t_pcie = now() - t0;

    t0 = now();
    const uint32_t chunk_samples = (uint32_t)(chunk[0] * chunk[1]);
    if (n_full) {
        if ((rc = drx_plan_create_uniform(ctx, n_full, chunk_samples, L, k, &plan)) != DRX_OK) goto out;
        if ((rc = drx_plan_set_filter(plan, o.n_taps, o.taps)) != DRX_OK) goto out;
        if ((rc = drx_decode(plan, (const uint32_t *)d_words, words, d_off, d_out)) != DRX_OK) goto out;
        if ((rc = drx_plan_finish(plan, NULL)) != DRX_OK) goto out;
    }
    if (edge_rows) {
        if (hipMalloc(&d_edge, (size_t)chunk_samples * 2) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
        if ((rc = drx_plan_create_uniform(ctx, 1, chunk_samples, L, k, &plan_edge)) != DRX_OK) goto out;
        if ((rc = drx_plan_set_filter(plan_edge, o.n_taps, o.taps)) != DRX_OK) goto out;
        if ((rc = drx_decode(plan_edge, (const uint32_t *)d_words, words, d_off + n_full, (int16_t *)d_edge)) != DRX_OK) goto out;
        if ((rc = drx_plan_finish(plan_edge, NULL)) != DRX_OK) goto out;
        if (hipMemcpyAsync(d_out + n_full * chunk_samples, d_edge, (size_t)(edge_rows * dims[1]) * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { rc = DRX_ERR_DEVICE; goto out; }
    }
    rc = DRX_OK;
    s.t_gpu = now() - t0;
out:
    if (plan) drx_plan_destroy(plan);
    if (plan_edge) drx_plan_destroy(plan_edge);
    if (d_edge) (void)hipFree(d_edge);
    if (d_words) (void)hipFree(d_words);
    if (d_off) (void)hipFree(d_off);
    if (h_words) (void)hipHostFree(h_words);
    free(h_off);
    if (pl >= 0) H5Pclose(pl);
    if (sp >= 0) H5Sclose(sp);
    if (d >= 0) H5Dclose(d);
    if (f >= 0) H5Fclose(f);
    leave_device(prev_dev);
    if (st) *st = s;
    return rc;
}

drx_status drx_h5_write(drx_ctx *ctx, const char *file, const char *name, const int16_t *d_in,
                        uint64_t rows, uint64_t cols, uint64_t chunk_rows, unsigned rice_m,
                        unsigned wave_len, drx_h5_stats *st) {
    return drx_h5_write_filtered(ctx, file, name, d_in, rows, cols, chunk_rows, rice_m, wave_len, 0, NULL, st);
}

drx_status drx_h5_write_filtered(drx_ctx *ctx, const char *file, const char *name, const int16_t *d_in,
                                 uint64_t rows, uint64_t cols, uint64_t chunk_rows, unsigned rice_m,
                                 unsigned wave_len, unsigned n_taps, const int32_t *taps, drx_h5_stats *st) {
    if (!ctx || !file || !name || !d_in || !rows || !cols || !chunk_rows || chunk_rows > rows) return DRX_ERR_ARG;  /* HDF5: chunk <= dataset */
    if (n_taps > DRX_MAX_TAPS || (n_taps && !taps)) return DRX_ERR_ARG;
    drx_h5_stats s;
    memset(&s, 0, sizeof s);
    unsigned k;
    if (log2_m(rice_m, &k) || chunk_rows * cols > 0x7fffffffull) return DRX_ERR_ARG;
    /* rows that do not divide: HDF5 stores the last chunk full size, padded with the fill value (0) */
    const uint64_t n_full = rows / chunk_rows, edge_rows = rows % chunk_rows;
    const uint32_t chunk_samples = (uint32_t)(chunk_rows * cols);
    s.rows = rows; s.cols = cols; s.chunk_rows = chunk_rows; s.n_chunks = n_full + (edge_rows ? 1 : 0);
    s.raw_bytes = rows * cols * 2;
    drx_status rc = DRX_ERR_DEVICE;
    drx_plan *plan = NULL, *plan_edge = NULL;
    void *d_words = NULL, *h_words = NULL, *d_edge = NULL, *d_words_e = NULL;
    uint64_t *d_off = NULL, *h_off = NULL;
    hid_t f = -1, d = -1, sp = -1, pl = -1;
    hipStream_t stream = (hipStream_t)drx_ctx_stream(ctx);
    int prev_dev = -1;
    if (enter_device(ctx, &prev_dev) != 0) return DRX_ERR_DEVICE;

    double t0 = now();
    uint64_t words = 0, words_e = 0, cap = 0, cap_e = 0;
    if (hipMalloc((void **)&d_off, (s.n_chunks + 3) * 8) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
    if (n_full) {
        if ((rc = drx_plan_create_uniform(ctx, n_full, chunk_samples, wave_len, k, &plan)) != DRX_OK) goto out;
        if (n_taps && (rc = drx_plan_set_filter(plan, n_taps, taps)) != DRX_OK) goto out;
        cap = drx_plan_max_encoded_words(plan);
        if (hipMalloc(&d_words, cap * 4) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
        if ((rc = drx_encode(plan, d_in, (uint32_t *)d_words, cap, d_off)) != DRX_OK) goto out;
        if ((rc = drx_plan_finish(plan, &words)) != DRX_OK) goto out;
    }
    if (edge_rows) {
        if ((rc = drx_plan_create_uniform(ctx, 1, chunk_samples, wave_len, k, &plan_edge)) != DRX_OK) goto out;
        if (n_taps && (rc = drx_plan_set_filter(plan_edge, n_taps, taps)) != DRX_OK) goto out;
        cap_e = drx_plan_max_encoded_words(plan_edge);
        if (hipMalloc(&d_edge, (size_t)chunk_samples * 2) != hipSuccess || hipMalloc(&d_words_e, cap_e * 4) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
        if (hipMemsetAsync(d_edge, 0, (size_t)chunk_samples * 2, stream) != hipSuccess ||
            hipMemcpyAsync(d_edge, d_in + n_full * chunk_samples, (size_t)(edge_rows * cols) * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess) { rc = DRX_ERR_DEVICE; goto out; }
        if ((rc = drx_encode(plan_edge, (const int16_t *)d_edge, (uint32_t *)d_words_e, cap_e, d_off + n_full + 1)) != DRX_OK) goto out;
        if ((rc = drx_plan_finish(plan_edge, &words_e)) != DRX_OK) goto out;
    }
    s.t_gpu = now() - t0;
    s.stored_bytes = (words + words_e) * 4;

    t0 = now();
    rc = DRX_ERR_NOMEM;
    h_off = (uint64_t *)malloc((s.n_chunks + 3) * 8);
    if (!h_off || hipHostMalloc(&h_words, (words + words_e + 1) * 4, hipHostMallocDefault) != hipSuccess) goto out;
    rc = DRX_ERR_DEVICE;
    h_off[0] = 0;
    if (n_full && (hipMemcpyAsync(h_words, d_words, words * 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                   hipMemcpyAsync(h_off, d_off, (n_full + 1) * 8, hipMemcpyDeviceToHost, stream) != hipSuccess)) goto out;
    if (edge_rows && hipMemcpyAsync((uint32_t *)h_words + words, d_words_e, words_e * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) goto out;
    if (hipStreamSynchronize(stream) != hipSuccess) goto out;
    if (edge_rows) h_off[n_full + 1] = words + words_e;  /* h_off[n_full] == words (or 0 without full chunks) */
    s.t_pcie = now() - t0;

    t0 = now();
    rc = DRX_ERR_ARG;
    ensure_filter_registered();
    if ((f = H5Fcreate(file, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)) < 0) goto out;
    hsize_t dims[2] = {rows, cols}, chunk[2] = {chunk_rows, cols};
    sp = H5Screate_simple(2, dims, NULL);
    pl = H5Pcreate(H5P_DATASET_CREATE);
    unsigned cd[3 + DRX_MAX_TAPS] = {rice_m, wave_len ? wave_len : 0xffffffffu, n_taps};  /* src/deltaRice.c:248-291 */
    for (unsigned j = 0; j < n_taps; ++j) cd[3 + j] = (unsigned)taps[j];
    if (H5Pset_chunk(pl, 2, chunk) < 0 || H5Pset_filter(pl, FILTER_ID, H5Z_FLAG_MANDATORY, n_taps ? 3 + n_taps : 2, cd) < 0) goto out;
    if ((d = H5Dcreate2(f, name, H5T_NATIVE_SHORT, sp, H5P_DEFAULT, pl, H5P_DEFAULT)) < 0) goto out;
    for (uint64_t c = 0; c < s.n_chunks; ++c) {
        hsize_t off[2] = {c * chunk_rows, 0};
        if (H5Dwrite_chunk(d, H5P_DEFAULT, 0, off, (size_t)(h_off[c + 1] - h_off[c]) * 4,
                           (const uint32_t *)h_words + h_off[c]) < 0) goto out;
    }
    rc = DRX_OK;
out:
    if (d >= 0) H5Dclose(d);
    if (pl >= 0) H5Pclose(pl);
    if (sp >= 0) H5Sclose(sp);
    if (f >= 0) { if (H5Fclose(f) < 0 && rc == DRX_OK) rc = DRX_ERR_ARG; }
    s.t_file = now() - t0;
    if (plan) drx_plan_destroy(plan);
    if (plan_edge) drx_plan_destroy(plan_edge);
    if (d_words) (void)hipFree(d_words);
    if (d_words_e) (void)hipFree(d_words_e);
    if (d_edge) (void)hipFree(d_edge);
    if (d_off) (void)hipFree(d_off);
    if (h_words) (void)hipHostFree(h_words);
    free(h_off);
    leave_device(prev_dev);
    if (st) *st = s;
    return rc;
}
