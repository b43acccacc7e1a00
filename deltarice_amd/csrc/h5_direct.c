/*
 * h5_direct.c -- direct-chunk HDF5 file <-> VRAM path (include/deltarice_h5io.h).
 * Host-side caller of the hot path: HDF5 chunk I/O + one PCIe copy + one batched drx_* call.
 */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <hdf5.h>
#include <pthread.h>
#include <stdio.h>
#include <unistd.h>
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

/* A second stream for the copies of this path (one per device, made on first use, never destroyed): chunk data crosses
 * PCIe on it while the calling thread is inside HDF5. */
static hipStream_t copy_stream_of(int dev) {
    static hipStream_t cs[64];
    if (dev < 0 || dev >= 64) return NULL;
    if (!cs[dev] && hipStreamCreateWithFlags(&cs[dev], hipStreamNonBlocking) != hipSuccess) cs[dev] = NULL;
    return cs[dev];
}

/* The chunks of a dataset in SLABS of consecutive chunks, at least 8 MB each and at most kMaxSlabs of them: the unit in which
 * data moves between the file and the device while the other side works on the slab before / behind it. */
enum { kMaxSlabs = 64 };
typedef struct { uint64_t c0, c1; } slab_t;
static int make_slabs(const uint64_t *off_words, uint64_t n_chunks, slab_t *slabs) {
    const uint64_t total = off_words[n_chunks] * 4;
    uint64_t target = total / kMaxSlabs + 1;
    if (target < (8u << 20)) target = 8u << 20;
    int n = 0;
    uint64_t c = 0;
    while (c < n_chunks) {
        uint64_t e = c + 1;
        while (e < n_chunks && (off_words[e] - off_words[c]) * 4 < target) ++e;
        if (n == kMaxSlabs - 1) e = n_chunks;
        slabs[n].c0 = c; slabs[n].c1 = e;
        ++n;
        c = e;
    }
    return n;
}

/* file -> pinned memory by plain pread(2) from several threads, where HDF5 tells where the chunks lie (H5Dget_chunk_info_by_coord,
 * 1.10.5+) and the file is one the operating system can read as it is (sec2 driver, no user block, opened read-only): four
 * threads copy out of the page cache at several times the rate of one, and H5Dread_chunk is one thread by construction. */
typedef struct {
    int fd;
    const uint64_t *addr, *off_words;  /* per chunk: file address, word offset in the staging buffer */
    uint8_t *dst;
    const slab_t *slabs;
    int n_slabs, n_threads, failed;
    int done[kMaxSlabs];
    pthread_mutex_t mu;
    pthread_cond_t cv;
} raw_reader;
typedef struct { raw_reader *r; int t; } raw_arg;
static void *raw_reader_main(void *p) {
    raw_arg *a = (raw_arg *)p;
    raw_reader *r = a->r;
    for (int s = a->t; s < r->n_slabs; s += r->n_threads) {
        int ok = 1;
        for (uint64_t c = r->slabs[s].c0; c < r->slabs[s].c1 && ok; ++c) {
            uint64_t left = (r->off_words[c + 1] - r->off_words[c]) * 4, at = 0;
            while (left) {
                const ssize_t got = pread(r->fd, r->dst + r->off_words[c] * 4 + at, left, (off_t)(r->addr[c] + at));
                if (got <= 0) { ok = 0; break; }
                left -= (uint64_t)got; at += (uint64_t)got;
            }
        }
        pthread_mutex_lock(&r->mu);
        r->done[s] = 1;
        if (!ok) r->failed = 1;
        pthread_cond_broadcast(&r->cv);
        pthread_mutex_unlock(&r->mu);
    }
    return NULL;
}

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
    void *h_words = NULL, *d_words = NULL;  /* (h_words: the context's staging buffer, not freed here) */
    uint64_t *h_off = NULL, *d_off = NULL, *h_addr = NULL;
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

    /* sizes -> offsets -> the context's pinned staging buffer <- the stored chunks, slab by slab, each slab on its way to
     * the device while the next one is read */
    h_off = (uint64_t *)malloc((s.n_chunks + 1) * sizeof(uint64_t));
    h_addr = (uint64_t *)malloc((s.n_chunks + 1) * sizeof(uint64_t));
    if (!h_off || !h_addr) { rc = DRX_ERR_NOMEM; goto out; }
    uint64_t words = 0;
    int raw_ok = 1;
    for (uint64_t c = 0; c < s.n_chunks; ++c) {
        hsize_t off[2] = {c * chunk[0], 0}, nb = 0;
#if H5_VERSION_GE(1, 10, 5)
        unsigned fmask = 0;
        haddr_t addr = HADDR_UNDEF;
        if (H5Dget_chunk_info_by_coord(d, off, &fmask, &addr, &nb) < 0 || (nb & 3)) { rc = DRX_ERR_CORRUPT; goto out; }
        if (addr == HADDR_UNDEF || fmask) raw_ok = 0;
        h_addr[c] = (uint64_t)addr;
        if (addr == HADDR_UNDEF) nb = 0;
#else
        raw_ok = 0;
        if (H5Dget_chunk_storage_size(d, off, &nb) < 0 || (nb & 3)) { rc = DRX_ERR_CORRUPT; goto out; }
#endif
        if (nb == 0) { rc = DRX_ERR_UNSUPPORTED; goto out; }  /* a chunk that was never written (fill value): not a stored stream */
        h_off[c] = words;
        words += nb / 4;
    }
    h_off[s.n_chunks] = words;
    s.stored_bytes = words * 4;
    if (raw_ok) {  /* a file the operating system can read as it is? */
        hid_t fapl = H5Fget_access_plist(f), fcpl = H5Fget_create_plist(f);
        hsize_t ub = 1;
        unsigned intent = H5F_ACC_RDWR;
        if (fapl < 0 || fcpl < 0 || H5Pget_driver(fapl) != H5FD_SEC2 || H5Pget_userblock(fcpl, &ub) < 0 || ub != 0 ||
            H5Fget_intent(f, &intent) < 0 || intent != H5F_ACC_RDONLY || getenv("DRX_H5_NO_RAW"))
            raw_ok = 0;
        if (fapl >= 0) H5Pclose(fapl);
        if (fcpl >= 0) H5Pclose(fcpl);
    }
    if (drx_ctx_host_staging(ctx, words * 4, &h_words) != DRX_OK) { rc = DRX_ERR_NOMEM; goto out; }
    hipStream_t stream = (hipStream_t)drx_ctx_stream(ctx);
    if (hipMalloc(&d_words, words * 4) != hipSuccess || hipMalloc((void **)&d_off, (s.n_chunks + 1) * 8) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
    slab_t slabs[kMaxSlabs];
    const int n_slabs = make_slabs(h_off, s.n_chunks, slabs);
    int raw_fd = raw_ok ? open(file, O_RDONLY | O_CLOEXEC) : -1;
    double t_wait = 0;
    if (raw_fd >= 0) {
        raw_reader rr;
        memset(&rr, 0, sizeof rr);
        rr.fd = raw_fd; rr.addr = h_addr; rr.off_words = h_off; rr.dst = (uint8_t *)h_words; rr.slabs = slabs; rr.n_slabs = n_slabs;
        rr.n_threads = n_slabs < 4 ? n_slabs : 4;
        pthread_mutex_init(&rr.mu, NULL);
        pthread_cond_init(&rr.cv, NULL);
        pthread_t th[4];
        raw_arg args[4];
        int started = 0;
        for (int t = 0; t < rr.n_threads; ++t) {
            args[t].r = &rr; args[t].t = t;
            if (pthread_create(&th[t], NULL, raw_reader_main, &args[t]) != 0) break;
            ++started;
        }
        int bad = started != rr.n_threads;
        if (bad) {  /* (the slabs of a thread that did not start would never be done) */
            pthread_mutex_lock(&rr.mu);
            rr.failed = 1;
            pthread_mutex_unlock(&rr.mu);
        }
        for (int sl = 0; sl < n_slabs && !bad; ++sl) {
            pthread_mutex_lock(&rr.mu);
            while (!rr.done[sl] && !rr.failed) pthread_cond_wait(&rr.cv, &rr.mu);
            bad = rr.failed;
            pthread_mutex_unlock(&rr.mu);
            if (bad) break;
            const uint64_t w0 = h_off[slabs[sl].c0], w1 = h_off[slabs[sl].c1];
            if (hipMemcpyAsync((uint32_t *)d_words + w0, (uint32_t *)h_words + w0, (w1 - w0) * 4, hipMemcpyHostToDevice, stream) != hipSuccess) bad = 1;
        }
        for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
        pthread_mutex_destroy(&rr.mu);
        pthread_cond_destroy(&rr.cv);
        close(raw_fd);
        if (bad) { (void)hipStreamSynchronize(stream); rc = DRX_ERR_CORRUPT; goto out; }
    } else {
        for (int sl = 0; sl < n_slabs; ++sl) {
            for (uint64_t c = slabs[sl].c0; c < slabs[sl].c1; ++c) {
                hsize_t off[2] = {c * chunk[0], 0};
                uint32_t mask = 0;
                if (H5Dread_chunk(d, H5P_DEFAULT, off, &mask, (uint32_t *)h_words + h_off[c]) < 0 || mask) { (void)hipStreamSynchronize(stream); rc = DRX_ERR_CORRUPT; goto out; }
            }
            const uint64_t w0 = h_off[slabs[sl].c0], w1 = h_off[slabs[sl].c1];
            if (hipMemcpyAsync((uint32_t *)d_words + w0, (uint32_t *)h_words + w0, (w1 - w0) * 4, hipMemcpyHostToDevice, stream) != hipSuccess) { rc = DRX_ERR_DEVICE; goto out; }
        }
    }
    s.t_file = now() - t0;

    t0 = now();
    if (hipMemcpyAsync(d_off, h_off, (s.n_chunks + 1) * 8, hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) { rc = DRX_ERR_DEVICE; goto out; }
    t_wait = now() - t0;
    s.t_pcie = t_wait;  /* (what of the copies was left to wait for behind the last slab's read) */

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
    free(h_off);
    free(h_addr);
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
    void *d_words = NULL, *h_words = NULL, *d_edge = NULL, *d_words_e = NULL;  /* (h_words: the context's staging buffer, not freed here) */
    uint64_t *d_off = NULL, *h_off = NULL;
    hid_t f = -1, d = -1, sp = -1, pl = -1;
    hipStream_t stream = (hipStream_t)drx_ctx_stream(ctx);
    int prev_dev = -1;
    if (enter_device(ctx, &prev_dev) != 0) return DRX_ERR_DEVICE;

    double t0 = now();
    const int timing = getenv("DRX_H5_TIMING") != NULL;
#define STAMP(what) do { if (timing) fprintf(stderr, "  h5 write %-28s %8.3f ms\n", what, (now() - t0) * 1e3); } while (0)
    uint64_t words = 0, words_e = 0, cap = 0, cap_e = 0;
    hipEvent_t ev[kMaxSlabs];
    int n_ev = 0;
    if (hipMalloc((void **)&d_off, (s.n_chunks + 3) * 8) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
    /* the encoders are launched ... */
    if (n_full) {
        if ((rc = drx_plan_create_uniform(ctx, n_full, chunk_samples, wave_len, k, &plan)) != DRX_OK) goto out;
        if (n_taps && (rc = drx_plan_set_filter(plan, n_taps, taps)) != DRX_OK) goto out;
        STAMP("plan created");
        cap = drx_plan_max_encoded_words(plan);
        if (hipMalloc(&d_words, cap * 4) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
        STAMP("output buffer allocated");
        if ((rc = drx_encode(plan, d_in, (uint32_t *)d_words, cap, d_off)) != DRX_OK) goto out;
        STAMP("encode launched");
    }
    if (edge_rows) {
        if ((rc = drx_plan_create_uniform(ctx, 1, chunk_samples, wave_len, k, &plan_edge)) != DRX_OK) goto out;
        if (n_taps && (rc = drx_plan_set_filter(plan_edge, n_taps, taps)) != DRX_OK) goto out;
        cap_e = drx_plan_max_encoded_words(plan_edge);
        if (hipMalloc(&d_edge, (size_t)chunk_samples * 2) != hipSuccess || hipMalloc(&d_words_e, cap_e * 4) != hipSuccess) { rc = DRX_ERR_NOMEM; goto out; }
        if (hipMemsetAsync(d_edge, 0, (size_t)chunk_samples * 2, stream) != hipSuccess ||
            hipMemcpyAsync(d_edge, d_in + n_full * chunk_samples, (size_t)(edge_rows * cols) * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess) { rc = DRX_ERR_DEVICE; goto out; }
        if ((rc = drx_encode(plan_edge, (const int16_t *)d_edge, (uint32_t *)d_words_e, cap_e, d_off + n_full + 1)) != DRX_OK) goto out;
    }
    /* ... and run while the file and the dataset are created */
    double t1 = now();
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
    s.t_file = now() - t1;
    STAMP("file and dataset created");
    t1 = now();
    if (n_full && (rc = drx_plan_finish(plan, &words)) != DRX_OK) goto out;
    STAMP("encode finished");
    if (edge_rows && (rc = drx_plan_finish(plan_edge, &words_e)) != DRX_OK) goto out;
    s.t_gpu = (t1 - t0 - s.t_file) + (now() - t1);  /* launches + what was left of the kernels behind the file's creation */
    s.stored_bytes = (words + words_e) * 4;

    /* chunk offsets, then the chunks themselves slab by slab on the copy stream into the context's pinned staging buffer,
     * each slab written to the file (H5Dwrite_chunk) while the slabs behind it cross PCIe */
    t0 = now();
    rc = DRX_ERR_NOMEM;
    h_off = (uint64_t *)malloc((s.n_chunks + 3) * 8);
    if (!h_off || drx_ctx_host_staging(ctx, (words + words_e + 1) * 4, &h_words) != DRX_OK) goto out;
    rc = DRX_ERR_DEVICE;
    h_off[0] = 0;
    if (n_full && (hipMemcpyAsync(h_off, d_off, (n_full + 1) * 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)) goto out;
    if (edge_rows) h_off[n_full + 1] = words + words_e;  /* h_off[n_full] == words (or 0 without full chunks) */
    hipStream_t cs = copy_stream_of(drx_ctx_device(ctx));
    if (!cs) cs = stream;
    slab_t slabs[kMaxSlabs];
    const int n_slabs = make_slabs(h_off, s.n_chunks, slabs);
    for (int sl = 0; sl < n_slabs; ++sl) {
        /* (the edge chunk, if any, is the last chunk and lies in its own device buffer) */
        const uint64_t c1 = slabs[sl].c1 > n_full ? n_full : slabs[sl].c1;
        if (slabs[sl].c0 < c1 && hipMemcpyAsync((uint32_t *)h_words + h_off[slabs[sl].c0], (const uint32_t *)d_words + h_off[slabs[sl].c0],
                                                (h_off[c1] - h_off[slabs[sl].c0]) * 4, hipMemcpyDeviceToHost, cs) != hipSuccess) goto out;
        if (slabs[sl].c1 > n_full && hipMemcpyAsync((uint32_t *)h_words + words, d_words_e, words_e * 4, hipMemcpyDeviceToHost, cs) != hipSuccess) goto out;
        if (hipEventCreateWithFlags(&ev[n_ev], hipEventDisableTiming) != hipSuccess) goto out;
        ++n_ev;
        if (hipEventRecord(ev[n_ev - 1], cs) != hipSuccess) goto out;
    }
    STAMP("copies issued");
    double t_wait = now() - t0, t_h5 = 0;
    for (int sl = 0; sl < n_slabs; ++sl) {
        t1 = now();
        if (hipEventSynchronize(ev[sl]) != hipSuccess) goto out;
        t_wait += now() - t1;
        t1 = now();
        for (uint64_t c = slabs[sl].c0; c < slabs[sl].c1; ++c) {
            hsize_t off[2] = {c * chunk_rows, 0};
            if (H5Dwrite_chunk(d, H5P_DEFAULT, 0, off, (size_t)(h_off[c + 1] - h_off[c]) * 4,
                               (const uint32_t *)h_words + h_off[c]) < 0) { rc = DRX_ERR_ARG; goto out; }
        }
        t_h5 += now() - t1;
    }
    s.t_pcie = t_wait;  /* (the copies the calling thread had to WAIT for; the rest ran under H5Dwrite_chunk) */
    s.t_file += t_h5;
    t0 = now();
    rc = DRX_OK;
out:
    if (d >= 0) H5Dclose(d);
    if (pl >= 0) H5Pclose(pl);
    if (sp >= 0) H5Sclose(sp);
    if (f >= 0) { if (H5Fclose(f) < 0 && rc == DRX_OK) rc = DRX_ERR_ARG; }
    if (rc == DRX_OK) s.t_file += now() - t0;  /* (closing the dataset and the file) */
    if (n_ev) {  /* (an error path may leave copies in flight into the staging buffer and out of the device buffers) */
        (void)hipStreamSynchronize(copy_stream_of(drx_ctx_device(ctx)) ? copy_stream_of(drx_ctx_device(ctx)) : stream);
        for (int i = 0; i < n_ev; ++i) (void)hipEventDestroy(ev[i]);
    }
    if (plan) drx_plan_destroy(plan);
    if (plan_edge) drx_plan_destroy(plan_edge);
    if (d_words) (void)hipFree(d_words);
    if (d_words_e) (void)hipFree(d_words_e);
    if (d_edge) (void)hipFree(d_edge);
    if (d_off) (void)hipFree(d_off);
    free(h_off);
    leave_device(prev_dev);
    if (st) *st = s;
    return rc;
}
