/*
 * deltarice_h5filter.h -- the HDF5-facing surface of the MI355X Delta-Rice filter.
 *
 * Same names, C linkage and meaning as the reference's header
 * (/root/reference/src/deltaRice.h:7-15) and plugin shim
 * (/root/reference/src/deltaRice_h5plugin.c:4-5), so that code written against
 * the reference links against this library unchanged:
 *
 *   H5Z_FILTER_DELTARICE          src/deltaRice.h:7        filter id 32025
 *   superint                      src/deltaRice.h:8        the 64-bit bit-accumulator type of the CPU codec
 *   H5Z_DELTARICE[1]              src/deltaRice.c:19-28    H5Z_class2_t record
 *   H5Z_filter_deltarice          src/deltaRice.c:468-490  the H5Z_func_t callback
 *   deltarice_register_h5filter   src/deltaRice.c:494-501  explicit registration
 *   H5PLget_plugin_type/_info     src/deltaRice_h5plugin.c:4-5  dynamic loading
 *   init_filter                   src/hdf5_dl.c:194-267    bind to a given libhdf5
 *
 * compression_opts / cd_values keep the reference's meaning (src/deltaRice.c:248-291):
 *   ()            M = 8, the whole chunk is one waveform
 *   (M)           RiceParameter M = 2^k, 1 <= M <= 32768
 *   (M, L)        + WaveformLength L samples (-1: whole chunk)
 *   (M, L, n, t0..t{n-1})  + prediction filter taps (default [1,-1] = delta)
 *
 * Differences from the reference, all on paths where the reference misbehaves
 * (SURVEY.md Appendix B): H5PLget_plugin_info returns the class record (the
 * reference returns the integer 32025 and HDF5 crashes); failures return 0 and
 * leave *buf untouched as H5Zpublic.h requires (the reference returns (size_t)-1);
 * encoded input is validated before it is decoded.
 *
 * The arithmetic runs on the GPU through include/deltarice_hip.h; there is no
 * CPU implementation behind this surface.
 */
#ifndef DELTARICE_H5FILTER_H
#define DELTARICE_H5FILTER_H

#define H5Z_class_t_vers 2
#include "hdf5.h"

#define H5Z_FILTER_DELTARICE 32025
/* Exported by the reference's public header (src/deltaRice.h:8; the accumulator of
 * compressWithRiceCoding, src/deltaRice.c:194).  Nothing here uses it -- the bit packing happens in the
 * HIP kernels -- but code that includes the reference's header may name the type. */
typedef unsigned long long int superint;

#ifdef __cplusplus
extern "C" {
#endif

extern H5Z_class_t H5Z_DELTARICE[1];

size_t H5Z_filter_deltarice(unsigned flags, size_t cd_nelmts, const unsigned cd_values[],
                            size_t nbytes, size_t *buf_size, void **buf);

/* H5Zregister(H5Z_DELTARICE); < 0 on failure. */
int deltarice_register_h5filter(void);

/* Take H5Zregister from the given shared library (e.g. the libhdf5 an h5py build
 * links), 0 on success, -1 on failure.  Without it the symbol is looked up in the
 * process image. */
int init_filter(const char *libname);

#ifdef __cplusplus
}
#endif
#endif /* DELTARICE_H5FILTER_H */
