/*
 * deltarice_h5io.h -- direct-chunk HDF5 file <-> VRAM path.
 *
 * The H5Z callback (deltarice_h5filter.h) is handed one chunk at a time in host memory, so
 * every chunk crosses PCIe twice and costs a kernel launch of its own; the reference's authors
 * measured the same ceiling for their GPU prototype ("File -> VRAM", docs/Performance.md:74,87,89).
 * This interface moves whole datasets instead: the stored (filtered) bytes of all chunks go
 * between the file and ONE pinned staging buffer with H5Dread_chunk / H5Dwrite_chunk (HDF5 >=
 * 1.10.3, no filter pipeline involved), one PCIe copy, and ONE drx_decode / drx_encode over the
 * whole batch.  Files are bit-compatible both ways with the filter path and with the reference.
 *
 * Datasets: 2-D, 16-bit little-endian integers (signed or unsigned: the bytes are what is coded), chunked as
 * (chunk_rows x all columns), filter 32025 alone in the pipeline with any cd_values the filter accepts.  A chunk
 * that was never written (fill value only) is DRX_ERR_UNSUPPORTED, another element type or byte order likewise.
 * This library links libhdf5 (the application's); the codec itself stays in libdeltarice_hip.so.
 */
#ifndef DELTARICE_H5IO_H
#define DELTARICE_H5IO_H

#include <stdint.h>

#include "deltarice_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t rows, cols, chunk_rows, n_chunks;
    uint64_t raw_bytes, stored_bytes;  /* int16 payload / bytes in the file's chunks */
    double t_file, t_pcie, t_gpu;      /* seconds: HDF5 chunk I/O, host<->device copies, codec kernels */
} drx_h5_stats;

/* File -> VRAM: decodes dataset `name` of `file` into d_out (device int16[out_cap_samples]). */
drx_status drx_h5_read(drx_ctx *ctx, const char *file, const char *name, int16_t *d_out,
                       uint64_t out_cap_samples, drx_h5_stats *stats);

/* VRAM -> file: encodes d_in (device int16[rows*cols]) and writes it as dataset `name` (file is
 * created/truncated).  rice_m, wave_len: compression_opts (RiceParameter, WaveformLength). */
drx_status drx_h5_write(drx_ctx *ctx, const char *file, const char *name, const int16_t *d_in,
                        uint64_t rows, uint64_t cols, uint64_t chunk_rows, unsigned rice_m,
                        unsigned wave_len, drx_h5_stats *stats);

/* The same with a general prediction filter (cd_values[2..] = n_taps, taps..., src/deltaRice.c:277-289); n_taps = 0:
 * the delta filter, cd_values = (RiceParameter, WaveformLength) as drx_h5_write stores them. */
drx_status drx_h5_write_filtered(drx_ctx *ctx, const char *file, const char *name, const int16_t *d_in,
                                 uint64_t rows, uint64_t cols, uint64_t chunk_rows, unsigned rice_m,
                                 unsigned wave_len, unsigned n_taps, const int32_t *taps, drx_h5_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
