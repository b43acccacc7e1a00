/*
 * deltarice_hip.h -- C ABI of the MI355X (gfx950) Delta-Rice codec.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * It replaces the arithmetic of the reference's per-chunk filter
 * (/root/reference/src/deltaRice.c) with HIP kernels; the HDF5-facing surface
 * (H5Z_filter_deltarice, H5Z_DELTARICE, registration, plugin entry points) is
 * declared in deltarice_h5filter.h and is a thin host wrapper over this file.
 *
 * Reference interface replaced by each entry point:
 *   drx_parse_cd_values      parseCD_VALUES + determinePowerOf2  src/deltaRice.c:248-291,114-136
 *   drx_encode               writeWholeCompressedByteString      src/deltaRice.c:383-441
 *                            (perWaveCompression :365-381, encodeWaveform :49-63,
 *                             compressWithRiceCoding :191-244), for a batch of chunks
 *   drx_decode               readWholeCompressedByteString       src/deltaRice.c:301-341
 *                            (perWaveDecompression :293-297, decompressWithRiceCoding
 *                             :138-189, decodeWaveform :78-90), for a batch of chunks
 *   drx_filter_chunk_host    the body of H5Z_filter_deltarice    src/deltaRice.c:468-490
 *                            (host buffer in, malloc'ed host buffer out, one chunk)
 *
 * Data model.  A *batch* is a list of HDF5 chunks.  Chunk c holds n_samples[c]
 * int16 samples cut into waveforms of wave_len[c] samples (the last one may be
 * shorter, src/deltaRice.c:399-403).  Raw chunks lie back to back in one int16
 * device buffer.  The encoded batch is one uint32 device buffer holding each
 * chunk's filtered bytes exactly as the reference's filter would emit them
 *   u32 n_samples | { u32 n_i | u32 payload_i[n_i] } per waveform
 * plus a table chunk_word_off[n_chunks+1] of where each chunk starts (the role
 * HDF5's chunk index plays in a file).
 *
 * All device pointers are on the context's device; every call is ordered on the
 * context's stream and returns without waiting.  Errors found on the device
 * (capacity, corrupt stream) are collected by drx_plan_finish().
 * There is no CPU fallback anywhere behind this ABI.
 */
#ifndef DELTARICE_HIP_H
#define DELTARICE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRX_FILTER_ID 32025 /* H5Z_FILTER_DELTARICE, src/deltaRice.h:7 */
#define DRX_MAX_TAPS 64

typedef struct drx_ctx drx_ctx;   /* one device + one stream + scratch */
typedef struct drx_plan drx_plan; /* geometry of one batch, device resident */

typedef enum {
    DRX_OK = 0,
    DRX_ERR_ARG = 1,         /* bad argument / compression_opts (reference: stderr + -1, :116-135,394-397) */
    DRX_ERR_DEVICE = 2,      /* HIP runtime failure, or no usable GPU */
    DRX_ERR_CAPACITY = 3,    /* encoded batch does not fit out_cap_words */
    DRX_ERR_CORRUPT = 4,     /* encoded input fails validation (header chain, sizes) */
    DRX_ERR_UNSUPPORTED = 5, /* valid for the reference, not (yet) on the device */
    DRX_ERR_NOMEM = 6
} drx_status;

/* Parsed compression_opts = cd_values (src/deltaRice.c:248-291). */
typedef struct {
    uint32_t rice_k;   /* log2(RiceParameter), 0..15 */
    int64_t wave_len;  /* WaveformLength; -1 = the whole chunk is one waveform */
    uint32_t n_taps;   /* prediction filter; default 2 taps [1,-1] = delta */
    int32_t taps[DRX_MAX_TAPS];
} drx_opts;

const char *drx_version(void);
const char *drx_status_str(drx_status s);
int drx_device_count(void);

/* cd_values -> options.  Defaults: M=8, wave_len=-1, taps [1,-1]. */
drx_status drx_parse_cd_values(size_t cd_nelmts, const unsigned *cd_values, drx_opts *out);

/* hip_stream: a hipStream_t to borrow (e.g. the caller's), or NULL to create one. */
drx_status drx_ctx_create(int device, void *hip_stream, drx_ctx **out);
void drx_ctx_destroy(drx_ctx *ctx);
drx_status drx_ctx_synchronize(drx_ctx *ctx);
const char *drx_ctx_last_error(const drx_ctx *ctx);
void *drx_ctx_stream(const drx_ctx *ctx);
int drx_ctx_device(const drx_ctx *ctx); /* the HIP device the context was created on (-1 for NULL) */
/* A pinned host buffer of at least `bytes` that belongs to the context and is kept across calls (grown when a call asks for
 * more; freed with the context): staging for callers that move batches between host memory and the device, such as the
 * direct-chunk HDF5 path -- a fresh hipHostMalloc per call costs more than the copy it serves (7 ms for 113 MB).
 * One user at a time, like the context itself. */
drx_status drx_ctx_host_staging(drx_ctx *ctx, size_t bytes, void **host_out);

/* Plans.  chunk_wave_len[c] == 0 means "whole chunk" (WaveformLength = -1).
 * Allocates the per-waveform tables and every scratch buffer the batch's geometry can need on the device;
 * drx_encode / drx_decode allocate nothing.
 *
 * Threading: a context and its plans are for one thread at a time (calls are ordered on the context's stream);
 * drx_filter_chunk_host alone takes the context's lock and may be called from any thread (HDF5 does).
 * Every call runs on the context's device and restores the calling thread's current device before returning. */
drx_status drx_plan_create(drx_ctx *ctx, uint64_t n_chunks, const uint32_t *chunk_samples,
                           const uint32_t *chunk_wave_len, uint32_t rice_k, drx_plan **out);
drx_status drx_plan_create_uniform(drx_ctx *ctx, uint64_t n_chunks, uint32_t chunk_samples,
                                   uint32_t wave_len, uint32_t rice_k, drx_plan **out);
/* Prediction filter of the plan (cd_values[2..], src/deltaRice.c:277-289).  Default: the delta filter [1,-1].  Any other
 * taps (1 <= n_taps <= DRX_MAX_TAPS, taps[0] != 0) run on the GPU as well:
 *   up to four taps            the single-pass encoders' own kernels (forward filter in packed 16-bit math);
 *   ... and taps[0] = +-1      the fast decoders: a lane per waveform with the recurrence in the lane, or -- few long waveforms --
 *                              the block decoder with the inverse filter inside it (DRX_PATH_IIR_FUSED) or as a parallel pass of
 *                              its own behind it (DRX_PATH_IIR);
 *   anything else              the two-pass encoder and the simple decode kernel (DRX_PATH_SIMPLE): a lane per waveform, serial. */
drx_status drx_plan_set_filter(drx_plan *plan, uint32_t n_taps, const int32_t *taps);
void drx_plan_destroy(drx_plan *plan);
uint64_t drx_plan_n_chunks(const drx_plan *plan);
uint64_t drx_plan_total_samples(const drx_plan *plan);
uint64_t drx_plan_total_waves(const drx_plan *plan);
/* Worst-case size of the encoded batch in uint32 words (25 bits per sample + headers). */
uint64_t drx_plan_max_encoded_words(const drx_plan *plan);

/* Encode.  d_in: int16[total_samples]; d_out: uint32[out_cap_words];
 * d_chunk_word_off: uint64[n_chunks+1], written (word index of each chunk's first
 * word in d_out; last entry = total words). */
drx_status drx_encode(drx_plan *plan, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap_words,
                      uint64_t *d_chunk_word_off);

/* Decode.  d_in: uint32[in_words]; d_chunk_word_off: uint64[n_chunks+1], read;
 * d_out: int16[total_samples].  in_words is also what the decoder of few long waveforms sizes its blocks by
 * (32 in_words / total_samples bits per sample): stating the encoded size rather than a buffer's capacity
 * costs nothing and is worth up to 1.5x there; it never affects the result. */
drx_status drx_decode(drx_plan *plan, const uint32_t *d_in, uint64_t in_words,
                      const uint64_t *d_chunk_word_off, int16_t *d_out);

/* Decode with a side-band: d_wave_words = the n_i of every waveform (uint32[total_waves], e.g. a copy of what
 * drx_plan_wave_words() shows after the drx_encode that produced d_in).  The reference's format has no index, so drx_decode
 * must find every waveform's header by walking or searching the stream (src/deltaRice.c:320-325); a device-resident pipeline
 * that still has the encoder's table can hand it back and skip that.  NOT part of the HDF5 drop-in surface (a file holds no
 * such table) and never used by bench.py.  The table is checked against the stream (every header word must equal its n_i,
 * the chain must end at the chunk's end): a table that does not belong to the stream is DRX_ERR_CORRUPT. */
drx_status drx_decode_with_wave_words(drx_plan *plan, const uint32_t *d_in, uint64_t in_words,
                                      const uint64_t *d_chunk_word_off, const uint32_t *d_wave_words, int16_t *d_out);

/* RiceParameter optimiser (docs/Optimization.md:5-19 of the reference describes one, the tree does not
 * contain it): exact number of uint32 words drx_encode would emit for this batch with RiceParameter
 * 2^k, for every k = 0..15 (host array of 16), in one pass over the samples.  Synchronous. */
drx_status drx_estimate_words(drx_plan *plan, const int16_t *d_in, uint64_t words_out[16]);

/* Waits for the plan's last encode/decode, reports device-side errors and (for
 * encode) the number of words produced.  total_words may be NULL. */
drx_status drx_plan_finish(drx_plan *plan, uint64_t *total_words);

/* Device-side tables of the last call (valid until the next call on the plan):
 * per-waveform payload word counts n_i and header word offsets.  For tests/tools. */
const uint32_t *drx_plan_wave_words(const drx_plan *plan);
const uint64_t *drx_plan_wave_word_off(const drx_plan *plan);
/* Which decoders the plan's last drx_decode used (for tests and tools): the batch's geometry and filter choose among them. */
#define DRX_PATH_LANES_FUSED 1u /* a lane per waveform, header walk inside the launch */
#define DRX_PATH_LANES 2u       /* a lane per waveform behind a separate walk */
#define DRX_PATH_BLOCKS 4u      /* a workgroup per block of a waveform's stream (few long waveforms) */
#define DRX_PATH_LONG 8u        /* a workgroup per waveform */
#define DRX_PATH_SIMPLE 16u     /* the simple kernel (filters the fast kernels do not take) */
#define DRX_PATH_IIR 32u        /* residuals first, then the general filter's inverse in place, parallel inside a waveform */
#define DRX_PATH_IIR_FUSED 64u  /* the general filter's inverse inside the block decoder: one kernel, samples straight to the output */
uint32_t drx_plan_last_decode_path(const drx_plan *plan);
/* ... and which encoder its last drx_encode used (one value; bench.py names the kernel it prices by this, and the tests
 * hold the dispatch to it: the headline batch must take DRX_ENC_STREAM whatever in_words its decodes were given) */
#define DRX_ENC_TWO_PASS 1u    /* sizes, scan, pack (encode_impl 0; more than four taps) */
#define DRX_ENC_SEGMENTS 2u    /* the two-pass segment encoder (the geometries only it serves) */
#define DRX_ENC_FUSED 3u       /* k_encode_fused: a wavefront per waveform, one pass */
#define DRX_ENC_PIECES 4u      /* k_encode_pieces: runs of short waveforms / segments of long ones */
#define DRX_ENC_STREAM 5u      /* k_encode_stream: persistent, a ring per wavefront, a scanner */
#define DRX_ENC_STREAM_SEGS 6u /* k_encode_stream_segs: the same over segments of long waveforms */
uint32_t drx_plan_last_encode_path(const drx_plan *plan);
/* Copies n_i of every waveform to host memory (waits for the stream). */
drx_status drx_plan_read_wave_words(drx_plan *plan, uint32_t *host_out);

/* One chunk, host memory, filter semantics (the H5Z callback body):
 * reverse == 0: in = nbytes of int16 samples -> *out = filtered bytes
 * reverse != 0: in = nbytes of filtered bytes -> *out = int16 samples
 * *out is allocated with malloc() (HDF5's allocator for filter buffers,
 * src/deltaRice.c:412,434); the caller owns it.  `in` is not freed. */
drx_status drx_filter_chunk_host(drx_ctx *ctx, int reverse, size_t cd_nelmts,
                                 const unsigned *cd_values, const void *in, size_t nbytes,
                                 void **out, size_t *out_bytes);

/* Kernel times of the plan's last call, measured with HIP events on the context's
 * stream (needs the context option "profile" = 1 before the call; waits for it).
 *   after drx_encode: ms = { size pass, offset scan, pack pass, whole call }
 *   after drx_decode: ms = { header-chain walk, decode kernel, 0, whole call } */
drx_status drx_plan_last_timings(drx_plan *plan, float ms[4]);

/* Tuning / diagnostics.  Returns DRX_ERR_ARG for unknown keys or values.
 *   "profile"      1: bracket the kernels with HIP events (drx_plan_last_timings)
 *   "encode_impl"  2 (default): single pass; one wavefront per waveform runs in the persistent form (wavefronts on their own,
 *                  a scanner workgroup for the prefix sum) wherever the standard geometry applies;
 *                  1: single pass with a look-back per workgroup everywhere (round 3's form);  0: size pass + scan + pack pass
 *   "decode_impl"  variant of the lane-per-waveform decode; each is bit-exact and covered by the parity tests:
 *        8 (default)  behind the parallel header walks where a batch takes them (chunks of 8 ... 8192 waveforms longer than 2048
 *                     samples, whatever their number: 64 chains per chunk are chased at once; few chunks of short waveforms);
 *                     otherwise the header walk inside the launch where the batch is large enough to hide it
 *        7            always behind a separate walk kernel
 *        0            simple kernel (also: general filters the staged kernel does not take)
 *        (5 / 1: one sample per ring access, the form 8 / 7 superseded in round 1 -- only in builds made with -DDRX_LEGACY)
 *   "debug_flags"  dispatch overrides that force an alternative (still bit-exact) path, for tests and A/B timing:
 *        256 never the long-waveform paths, 512 long waveforms one workgroup each, 2048 never the parallel header walks,
 *        4096 never the pieces encoder, 8192 always the segment encoder, 32768 the pieces encoder also where one wavefront per
 *        waveform is the default, 65536 never the single-pass encoder's larger-buffer geometries (RiceParameter above 8),
 *        262144 the persistent encoder on three workgroups (every wavefront codes many waveforms of a small batch),
 *        524288 the persistent encoder (encode_impl 2) whatever the batch's size and expected code length,
 *        2097152 general filters behind the block decoder: always the separate inverse-filter pass,
 *        4194304 the persistent encoder's segment form (long waveforms) wherever the batch is uniform, in segments of ~1024
 *        samples, 8388608 the chunk-wide header walk by reading the whole chunk instead of chasing 64 chains, 16777216 by
 *        chains also where the scan form is the default (one to four chunks).  (Ablation switches inside the kernels exist only in -DDRX_ABLATION builds.) */
drx_status drx_ctx_set_option(drx_ctx *ctx, const char *key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* DELTARICE_HIP_H */
