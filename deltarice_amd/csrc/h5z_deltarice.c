/*
 * h5z_deltarice.c -- HDF5 filter 32025 ("deltarice") backed by the MI355X codec.
 *
 * Host side of the drop-in boundary, in C like the reference's filter
 * (/root/reference/src/deltaRice.c:19-32,468-501, src/deltaRice_h5plugin.c).
 * Per chunk HDF5 hands this callback a host buffer; it crosses PCIe to the GPU,
 * is encoded/decoded by the HIP kernels behind include/deltarice_hip.h and comes
 * back in a malloc'ed buffer that replaces *buf (same ownership protocol as
 * src/deltaRice.c:433-436,336-340).  No CPU codec lives here.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

#include "deltarice_h5filter.h"
#include "deltarice_hip.h"
#include "H5PLextern.h"

H5Z_class_t H5Z_DELTARICE[1] = {{
    H5Z_CLASS_T_VERS,                     /* H5Z_class_t version */
    (H5Z_filter_t)H5Z_FILTER_DELTARICE,   /* filter id 32025 */
    1,                                    /* encoder present */
    1,                                    /* decoder present */
    "deltarice",                          /* name, as the reference's (src/deltaRice.c:24) */
    NULL,                                 /* can_apply: none, like the reference */
    NULL,                                 /* set_local: none, like the reference */
    (H5Z_func_t)H5Z_filter_deltarice,
}};

static drx_ctx *g_ctx;
static drx_status g_ctx_status;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void ctx_init(void) {
    const char *dev = getenv("DELTARICE_DEVICE");
    g_ctx_status = drx_ctx_create(dev ? atoi(dev) : 0, NULL, &g_ctx);
}


size_t H5Z_filter_deltarice(unsigned flags, size_t cd_nelmts, const unsigned cd_values[],
                            size_t nbytes, size_t *buf_size, void **buf) {
    if (!buf || !*buf || !buf_size) return 0;
    pthread_once(&g_once, ctx_init);
    if (!g_ctx) {
        fprintf(stderr, "deltarice: no usable MI355X device (%s)\n", drx_status_str(g_ctx_status));
        return 0;
    }
    void *out = NULL;
    size_t out_bytes = 0;
    const int reverse = (flags & H5Z_FLAG_REVERSE) != 0;
    drx_status st = drx_filter_chunk_host(g_ctx, reverse, cd_nelmts, cd_values, *buf, nbytes, &out, &out_bytes);
    if (st != DRX_OK) {
        fprintf(stderr, "deltarice: %s failed: %s (%s)\n", reverse ? "de-compression" : "compression",
                drx_status_str(st), drx_ctx_last_error(g_ctx));
        return 0; /* HDF5 contract (H5Zpublic.h): 0 = failure, buffers untouched */
    }
    free(*buf);
    *buf = out;
    *buf_size = out_bytes;
    return out_bytes;
}

/* ---- registration ---------------------------------------------------------- */
typedef herr_t (*h5zregister_fn)(const void *);
static h5zregister_fn g_h5zregister;

int init_filter(const char *libname) {
    void *h = dlopen(libname, RTLD_LAZY | RTLD_LOCAL);
    if (!h) return -1;
    h5zregister_fn f = (h5zregister_fn)dlsym(h, "H5Zregister");
    if (!f) { dlclose(h); return -1; }  /* not a library that carries HDF5: try the next one (src/h5.pyx:43-51) */
    g_h5zregister = f;
    return 0;
}

int deltarice_register_h5filter(void) {
    h5zregister_fn f = g_h5zregister;
    if (!f) f = (h5zregister_fn)dlsym(RTLD_DEFAULT, "H5Zregister");
    if (!f) {
        fprintf(stderr, "deltarice_register_h5filter: no libhdf5 in this process (call init_filter)\n");
        return -1;
    }
    int ret = (int)f(H5Z_DELTARICE);
    if (ret < 0) fprintf(stderr, "deltarice_register_h5filter: can't register deltarice filter\n");
    return ret;
}

/* ---- dynamically loaded plugin (HDF5_PLUGIN_PATH) -------------------------- */
H5PL_type_t H5PLget_plugin_type(void) { return H5PL_TYPE_FILTER; }
const void *H5PLget_plugin_info(void) { return H5Z_DELTARICE; }
