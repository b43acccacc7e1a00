/* Sanitizer driver (CPU container, no GPU): the failure paths of the HDF5 filter callback and of the registration entry
 * points in deltarice_amd/csrc/h5z_deltarice.c, compiled together with this file under -fsanitize=address,undefined
 * (`make -C oracle asan`).  The callback must report failure the HDF5 way -- return 0, *buf and *buf_size untouched, nothing
 * freed, nothing leaked (the reference returns (size_t)-1 and has already freed or leaked buffers on some of these paths,
 * /root/reference/src/deltaRice.c:394-397,477,486).  Prints "failpath ok" and exits 0; any sanitizer report aborts. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "deltarice_h5filter.h"
#include "H5PLextern.h"

int main(void) {
    unsigned cd[2] = {8, 1024};
    size_t size = 4096;
    unsigned char *p = malloc(size);
    memset(p, 7, size);
    void *buf = p;
    /* no usable device here: 0, buffer and size as they were (and still ours to free) */
    for (unsigned flags = 0; flags <= 0x100; flags += 0x100) {
        size_t r = H5Z_filter_deltarice(flags, 2, cd, 2048, &size, &buf);
        if (r != 0 || buf != (void *)p || size != 4096) { fprintf(stderr, "failure contract broken (flags %u)\n", flags); return 1; }
        for (size_t i = 0; i < 4096; ++i) if (p[i] != 7) { fprintf(stderr, "buffer modified\n"); return 1; }
    }
    /* argument checks come before anything else */
    if (H5Z_filter_deltarice(0, 2, cd, 2048, &size, NULL) != 0) return 1;
    if (H5Z_filter_deltarice(0, 2, cd, 2048, NULL, &buf) != 0) return 1;
    void *none = NULL;
    if (H5Z_filter_deltarice(0, 2, cd, 2048, &size, &none) != 0) return 1;
    free(p);
    /* registration without an HDF5 library in the process / with a library that is not HDF5 */
    if (init_filter("/nonexistent/libhdf5.so") == 0) return 1;
    if (init_filter("libm.so.6") == 0) return 1; /* loads, but carries no H5Zregister */
    if (deltarice_register_h5filter() >= 0) return 1;
    if (H5PLget_plugin_type() != H5PL_TYPE_FILTER || H5PLget_plugin_info() != (const void *)H5Z_DELTARICE) return 1;
    puts("failpath ok");
    return 0;
}
