// drx_internal.h -- types shared by the kernels and the C ABI glue (not installed).
#ifndef DRX_INTERNAL_H
#define DRX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace drx {

// One HDF5 chunk of the batch (device resident table, ragged batches only).
struct ChunkDesc {
    uint64_t sample_off;  // first sample of the chunk in the raw int16 batch
    uint64_t wave_base;   // global index of the chunk's first waveform
    uint32_t n_samples;   // N  (u32 header word of the encoded chunk, src/deltaRice.c:415)
    uint32_t wave_len;    // L  (>= 1; "whole chunk" already resolved to N)
    uint32_t n_waves;     // ceil(N / L)
    uint32_t pad_;
};

// Passed to kernels by value.
struct Geom {
    const ChunkDesc *chunks;  // device pointer, n_chunks entries (unused when uniform)
    uint64_t n_chunks;
    uint64_t total_waves;
    uint32_t uniform;  // all chunks share (n_samples, wave_len): pure arithmetic mapping
    uint32_t u_n_samples, u_wave_len, u_n_waves;
    uint32_t k;  // log2(M)
    // prediction filter (src/deltaRice.c:49-76,78-103).  n_taps == 0: the default [1,-1] (delta), which
    // the fast kernels implement; otherwise taps (device pointer, n_taps entries) select the general
    // FIR / IIR kernels.
    uint32_t n_taps;
    const int32_t *taps;
    // general filters the fast kernels take: at most 4 taps with taps[0] = +-1 (the inverse filter then has no
    // division); fast_t0neg = (taps[0] == -1), fast_nt[j-1] = -taps[j] (0 beyond n_taps)
    uint32_t fast_taps, fast_t0neg;
    uint32_t fast_nt[3];
    // forward filter in the single-pass encoder: at most 4 taps, any taps[0]; enc_t[j] = taps[j] mod 2^16
    uint32_t enc_fast;
    uint32_t enc_t[4];
    uint64_t total_samples;  // of the batch
    // pinned host word (device-visible) that every encoder writes the batch's encoded word count to beside DevStatus: the host
    // reads it -- without waiting for anything -- when it chooses the NEXT encode's kernel (drx_api.hip, stream_encoder_suits())
    uint64_t *host_words;
    uint32_t dbg;  // "debug_flags" context option; 0 in normal use.  Dispatch overrides (host side, always available, every
                   // forced path is bit-exact and the tests use them to reach it):
                   //   256 never take the long-waveform paths   512 long waveforms: one workgroup per waveform only
                   //  2048 never take the parallel header walks of small batches   8192 always the segment encoder
                   //  4096 never the pieces encoder   32768 the pieces encoder wherever its geometry allows
                   //  131072 ragged batches: one decode launch behind both header walks instead of one behind each
                   //  262144 k_encode_stream on three workgroups (tests: every wavefront goes around its ring)
                   //  524288 k_encode_stream (encode_impl 2) whatever the batch (else: where stream_encoder_suits())
                   //  2097152 general filters behind the block decoder: always the separate k_iir_tiles pass
                   //  4194304 k_encode_stream_segs wherever the batch is uniform, segments of kEsSegMinLen samples
                   //  8388608 the chunk-wide walk by reading the chunk (k_pw_scan + k_walk_parallel) instead of k_walk_sparse
                   // 16777216 k_walk_sparse also for one to four chunks and for ragged chunks of few long waveforms
                   // Ablation switches INSIDE the kernels, compiled only with -DDRX_ABLATION (results invalid):
                   //   decode:   1 skip the output stores   2 skip the stream loads
                   //             4 request pieces without counting on the round's minimum consumption   16384 long: no stores
                   //   encode:  16 per-code LDS emission instead of the lane-local concatenation
                   //            32 no emission   64 no copy-out   128 no look-back (positions wrong)
    // ragged batches, walk inside the decode launch: chunk indices, short-waveform chunks first
    // (walk_short[n_short], then walk_long[n_long]), and the largest ceil(n_waves / 64) of any chunk
    const uint32_t *walk_short, *walk_long;
    uint32_t n_short, n_long, max_groups;
    uint64_t max_wave_len64;  // ragged batches: 64 x the longest WaveformLength (how far apart a wavefront's 64 lines can lie)
    // ragged batches of few long waveforms: the block-parallel decoder takes them (decided when the plan is made, from the
    // host's chunk table: drx_blocks.hip); lanes per block and look-back slots per waveform for the longest of them
    uint32_t rag_blocks, rag_blk_nt, rag_blk_slots;
    // ... launched once per class of WaveformLengths (floor(log2 L): the waveforms of a launch differ by less than 2x, so
    // that run-major tickets over them are rarely empty): rag_blk_list = waveform indices class by class (device),
    // class c = entries [rag_blk_class_off[c], rag_blk_class_off[c + 1]), its longest WaveformLength in rag_blk_class_len[c]
    const uint32_t *rag_blk_list;
    uint32_t rag_blk_classes;
    uint32_t rag_blk_class_off[33], rag_blk_class_len[32];
    // general prediction filter behind the block decoder (drx_iir.hip): tables, tiles and their look-back state
    const uint32_t *iir_tab;
    const uint32_t *blk_iir_tab;  // ... and inside the block decoder (FUSE): blocks_iir_tables()
    const uint64_t *iir_chunk_tile_base;  // ragged: first tile of every chunk, n_chunks + 1 entries
    uint64_t iir_n_tiles;
    uint64_t *iir_state;                  // uint64[iir_n_tiles + 1]
    // ragged batches that the segment encoder takes (some chunk has short or long waveforms): first unit (waveform x
    // segment slot) of every chunk, n_chunks + 1 entries, and their total
    const uint64_t *seg_unit_base;
    uint64_t seg_units;
    // ragged batches small enough for the parallel header walks (drx_walk.h): set when the plan is made;
    // rag_bw_blocks_max = 4096-word blocks of the largest short-waveform chunk at 25 bits per sample
    uint32_t rag_par, rag_bw_blocks_max;
    uint32_t rag_bw_min_len;  // ... and the smallest WaveformLength among those chunks (bounds the headers of a block)
    uint32_t rag_pw_min_waves;  // ... and the fewest waveforms any LONG-waveform chunk has (k_walk_sparse: from 64 on)
    // ragged batches, lane-per-waveform decode outside the fused launch: {chunk, group of 64 waveforms} of every
    // wavefront, longest WaveformLength first; rag_groups entries
    const uint2 *rag_order;
    uint32_t rag_groups;
    uint32_t rag_groups_long;  // ... of which the first ones belong to chunks of WaveformLength > 2048 (the chunk-wide walk's)
    // ragged batches the pieces encoder takes (drx_pieces.hip): first workgroup of every chunk, n_chunks + 1 entries
    const uint32_t *pc_wg_base;
    uint32_t pc_super;  // ... and every chunk's WaveformLength is above kPcMaxLen (waveforms over several workgroups)
    uint32_t pc_packed;  // ... or every chunk's WaveformLength is piece_packable()
};

struct DevStatus {
    uint32_t err;  // kErr* bits, OR-ed by kernels
    uint32_t pad_;
    uint64_t total_words;  // encode: words of the encoded batch
};

constexpr uint32_t kErrCapacity = 1u;
constexpr uint32_t kErrCorrupt = 2u;
constexpr uint32_t kErrInternal = 4u;

hipError_t launch_encode(const Geom &G, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                         uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint32_t *d_wave_rel,
                         uint64_t *d_chunk_words, DevStatus *d_status, hipEvent_t *ev, hipStream_t s);

// launch_decode() takes the workgroup-per-waveform decoder for uniform batches whose lane-per-waveform decode would leave most of
// the 98 304 lane slots empty and the block decoder does not take: at most 16 384 waveforms of at least 65 536 samples, or at most
// 4 096 of at least 16 384 (measured crossover at 350 M samples: L = 32 768 lanes 2.2 ms / long 2.9 ms, L = 65 536 3.6 / 2.4 ms);
// the same test sends such batches to the segment encoder where the pieces encoder does not apply
__host__ __device__ inline bool long_waveform_batch(uint64_t total_waves, uint32_t wave_len) {
    return (wave_len >= 65536u && total_waves <= 16384u) || (wave_len >= 16384u && total_waves <= 4096u);
}
static inline unsigned blocks_for(uint64_t items, unsigned per_block) { return (unsigned)((items + per_block - 1) / per_block); }
static inline void mark(hipEvent_t *ev, int i, hipStream_t s) {  // (optional profiling events of a launch)
    if (ev) (void)hipEventRecord(ev[i], s);
}

hipError_t launch_sideband_tables(const Geom &G, const uint32_t *d_in, uint64_t in_words, const uint64_t *d_chunk_word_off,
                                  const uint32_t *d_n, uint64_t *d_wave_off, uint32_t *d_wave_words, DevStatus *d_status, hipStream_t s);
hipError_t launch_estimate_words(const Geom &G, const int16_t *d_in, unsigned long long *d_words16, hipStream_t s);

hipError_t launch_encode_fused(const Geom &G, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                               uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint64_t *d_scan,
                               DevStatus *d_status, hipEvent_t *ev, hipStream_t s);

// the persistent form of the single pass (drx_encode_stream.hip): a ring of kEsRingWords LDS words per wavefront, a scanner
// wavefront.  d_scan: uint64[2 * total_waves + 48]
#ifndef DRX_ES_RING
#define DRX_ES_RING 2496
#endif
constexpr uint32_t kEsRingWords = DRX_ES_RING;
hipError_t launch_encode_stream(const Geom &G, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                                uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint64_t *d_scan,
                                DevStatus *d_status, hipEvent_t *ev, hipStream_t s);

// ... and its form for LONG waveforms (k_encode_stream_segs): the unit a wavefront codes into its ring is a SEGMENT of a
// waveform, a ticket is kEsSegWaves consecutive segments of ONE waveform, places are bit positions.
// d_scan: uint64[3 * tickets + 48].  The shape is a function of WaveformLength and the segment length the caller aims at
// (what a ring holds with room for most of the next segment, from the bits per sample the plan expects).
constexpr uint32_t kEsSegWaves = 4;
constexpr uint32_t kEsSegMinLen = 1024, kEsSegMaxLen = 7168;  // samples per segment the host may aim at
struct EsSegShape { uint32_t seg_len, nseg, tpw; };           // samples per segment (a multiple of 8), segments and tickets per waveform
__host__ __device__ inline EsSegShape es_seg_shape(uint32_t L, uint32_t seg_target) {
    EsSegShape sh;
    uint32_t n = (L + seg_target - 1u) / seg_target;
    n = n ? n : 1u;
    // whole tickets: fewer, longer segments or more, shorter ones -- whichever costs less.  Measured on 50 chunks of 14 M samples
    // (profiles/r04_notes.md section 10): segments 17 % above the target (a ring then holds little of the next one) +9.5 %, 22 %
    // below it (a ticket's fixed costs per fewer samples) +9 %, 41 % below +16 %; never above 67 / 57 of the target (what
    // k_encode_stream accepts of a ring, stream_encoder_suits()).
    const uint32_t dn = n / kEsSegWaves * kEsSegWaves, up = (n + kEsSegWaves - 1u) / kEsSegWaves * kEsSegWaves;
    n = up;
    if (dn && dn != up) {
        const uint64_t sd = (L + dn - 1u) / dn, su = (L + up - 1u) / up;  // (sd > seg_target >= su)
        const uint64_t pen_dn = sd > seg_target ? (sd - seg_target) * 55u : 0u, pen_up = su < seg_target ? (seg_target - su) * 40u : 0u;
        if (sd * 57u <= (uint64_t)seg_target * 67u && pen_dn < pen_up) n = dn;
    }
    sh.seg_len = (((L + n - 1u) / n) + 7u) & ~7u;                     // equal segments, 16-byte steps
    sh.nseg = (L + sh.seg_len - 1u) / sh.seg_len;
    sh.tpw = (sh.nseg + kEsSegWaves - 1u) / kEsSegWaves;
    return sh;
}
hipError_t launch_encode_stream_segs(const Geom &G, uint32_t seg_target, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                                     uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint64_t *d_scan,
                                     DevStatus *d_status, hipEvent_t *ev, hipStream_t s);

// few long waveforms (WaveformLength = -1): a wavefront per 8192-sample segment, see drx_encode_kernels.hip
bool long_batch(const Geom &G);
uint64_t long_batch_units(const Geom &G);
hipError_t launch_encode_long(const Geom &G, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                              uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint32_t *d_wave_rel,
                              uint64_t *d_chunk_words, uint32_t *d_seg_bits, uint64_t *d_seg_pos, DevStatus *d_status,
                              hipEvent_t *ev, hipStream_t s);

// a second stream of the context, for kernels of one call that do not depend on each other (fork / join through events).
// One per CONTEXT, shared by all its plans: it relies on the rule of include/deltarice_hip.h that a context and its plans are
// driven by one thread at a time (a second thread decoding another plan of the same context would re-record fork / join
// in the middle of this call).  Every path out of launch_decode() after the fork joins, or synchronises the side stream.
struct SideStream {
    hipStream_t s;
    hipEvent_t fork, join;
};
// path_out (optional): which decoders the call used, DRX_PATH_* bits of include/deltarice_hip.h
hipError_t launch_decode(const Geom &G, const uint32_t *d_in, uint64_t in_words,
                         const uint64_t *d_chunk_word_off, int16_t *d_out, uint64_t *d_wave_off,
                         uint32_t *d_wave_words, uint64_t *d_granules, DevStatus *d_status, int impl,
                         void *d_pw, void *d_blk, const SideStream *side, hipEvent_t *ev, hipStream_t s, uint32_t *path_out = nullptr);
// block-parallel decoder for batches of few waveforms (drx_blocks.hip): a workgroup per block of a waveform's stream
bool blocks_batch(const Geom &G);
// ragged plans: decides rag_blocks / rag_blk_nt / rag_blk_slots from the host's chunk table (call once, when the plan is made)
// (list_out: the host copy of rag_blk_list, for the caller to upload)
void blocks_plan_ragged(Geom &G, const ChunkDesc *host_chunks, uint32_t *list_out);
uint64_t blocks_scratch_bytes(const Geom &G);
// resid: leave the residuals (not their running sums) in d_out: a general prediction filter's inverse follows (launch_iir)
// fused_out: a general filter's inverse ran inside the kernel (no k_iir_tiles pass is needed behind it)
hipError_t launch_decode_blocks(const Geom &G, const uint32_t *d_in, uint64_t in_words, const uint64_t *d_wave_off,
                                const uint32_t *d_wave_words, void *d_blk, DevStatus *d_status, int16_t *d_out,
                                const uint32_t **fail_out, const uint32_t **suspect_out, bool resid, hipStream_t s, bool *fused_out);
uint32_t blocks_iir_tab_words();
void blocks_iir_tables(const uint32_t fast_nt[3], uint32_t t0neg, uint32_t *tab);
// single-pass encoder for short and long waveforms (drx_pieces.hip): a wavefront takes a PIECE, either a run of whole
// short waveforms or one segment of a long one; the pieces of a chunk fill whole workgroups of kPcWaves wavefronts
constexpr uint32_t kPcWaves = 8;
constexpr uint32_t kPcRunSamples = 7168;  // samples of a run of short waveforms (9.1 bits per sample fit the LDS buffer)
constexpr uint32_t kPcMaxRun = 16;        // waveforms of a run
constexpr uint32_t kPcSegSamples = 8192;  // most samples of a segment
constexpr uint32_t kPcWholeLen = 10240;   // WaveformLengths up to here stay whole (6.4 bits per sample fit the buffer)
constexpr uint32_t kPcMinLen = 64;        // WaveformLengths it takes: from here
// A piece's code must fit its wavefront's 2048-word LDS buffer, or the piece is coded a second time, tile by tile, straight
// to its place (6 x slower: NOPTREX at sigma = 40, m = 32 -- 8.5 bits per sample -- encoded in 5.7 ms instead of 0.93; the
// headline shape at sigma = 80, m = 64 in 2.2 ms per 100 chunks instead of 1.25).  With the RiceParameter that suits the data a
// sample takes about k + 3.5 bits, so the pieces are sized for k + 4.5: the constants above are the measured geometry for
// k <= 3 (m <= 8), beyond that a segment holds 131072 / (2 k + 9) samples in whole tiles.
__host__ __device__ inline uint32_t pc_seg_samples(uint32_t k) {
    if (k <= 3u) return kPcSegSamples;
    const uint32_t s = (131072u / (2u * k + 9u)) & ~511u;
    return s < 512u ? 512u : s;
}
__host__ __device__ inline uint32_t pc_run_samples(uint32_t k) { return k <= 3u ? kPcRunSamples : pc_seg_samples(k) - 512u; }
// a waveform stays whole (one wavefront, and k_encode_fused's range stays k_encode_fused's) while it fits at k + 4 bits per
// sample: two segments of 3500 samples cost a quarter of the rate (100 chunks of 2000 x 7000, sigma = 40, m = 32: 1.84 against 1.26 ms)
__host__ __device__ inline uint32_t pc_whole_len(uint32_t k) { return k <= 3u ? kPcWholeLen : 65536u / (k + 4u); }
__host__ __device__ inline uint32_t pc_max_len(uint32_t k) { return pc_seg_samples(k) * kPcWaves; }  // beyond: several workgroups per waveform
struct PieceShape {
    uint32_t run;      // waveforms per piece (> 1: runs of short waveforms)
    uint32_t segs;     // pieces per waveform and workgroup (a power of two <= kPcWaves; > 1: long waveforms)
    uint32_t parts;    // workgroups per waveform (> 1: waveforms longer than kPcMaxLen, kPcWaves segments per workgroup)
    uint32_t seg_len;  // samples per segment (a multiple of 512)
    uint32_t pieces;   // of the chunk
    uint32_t wgs;      // workgroups of the chunk
};
// waveforms of fewer than 512 samples, a multiple of 8 (a lane's 8 samples never straddle two waveforms): PACKED runs, whose
// tiles span waveform boundaries
__host__ __device__ inline bool piece_packable(uint32_t L) { return L >= 8u && L < 512u && (L & 7u) == 0u; }
__host__ __device__ inline PieceShape piece_shape(uint32_t L, uint32_t W, uint32_t k, bool packed) {
    PieceShape s;
    s.parts = 1u;
    const uint32_t run_samples = pc_run_samples(k), seg_samples = pc_seg_samples(k), max_len = pc_max_len(k);
    if (packed) {
        // as many waveforms as fill ~7000 samples AND leave the 2048-word buffer room at 9 (k + 6) bits per sample + a header each
        const uint32_t bits = k <= 3u ? 9u : k + 6u;
        uint32_t by_samples = run_samples / L, by_words = 2000u / (1u + (bits * L + 31u) / 32u);
        by_samples = by_samples ? by_samples : 1u;
        by_words = by_words ? by_words : 1u;
        s.run = by_samples < by_words ? by_samples : by_words;
        s.segs = 1u;
        s.seg_len = L;
        s.pieces = (W + s.run - 1u) / s.run;
        s.wgs = (s.pieces + kPcWaves - 1u) / kPcWaves;
    } else if (L <= run_samples / 2u) {
        s.run = run_samples / L < kPcMaxRun ? run_samples / L : kPcMaxRun;
        s.segs = 1u;
        s.seg_len = L;
        s.pieces = (W + s.run - 1u) / s.run;
        s.wgs = (s.pieces + kPcWaves - 1u) / kPcWaves;
    } else if (L <= max_len) {
        const uint32_t need = L <= pc_whole_len(k) ? 1u : (L + seg_samples - 1u) / seg_samples;
        s.run = 1u;
        s.segs = 1u;
        while (s.segs < need) s.segs <<= 1;
        s.seg_len = s.segs == 1u ? L : ((((L + s.segs - 1u) / s.segs) + 511u) & ~511u);
        s.pieces = W * s.segs;
        s.wgs = (s.pieces + kPcWaves - 1u) / kPcWaves;
    } else {
        // a waveform over several workgroups ("parts"): kPcWaves segments each, as even as whole tiles allow (full 8192-sample
        // segments with a short last part instead were measured slower: nEDM 0.90 against 0.77 ms)
        s.run = 1u;
        s.segs = kPcWaves;
        s.parts = (uint32_t)(((uint64_t)L + max_len - 1u) / max_len);
        const uint32_t per_part = (uint32_t)(((uint64_t)L + s.parts - 1u) / s.parts);
        s.seg_len = (((per_part + kPcWaves - 1u) / kPcWaves) + 511u) & ~511u;
        s.pieces = 0u;  // (not used: every workgroup of the chunk is full)
        s.wgs = W * s.parts;  // (the plan checks that this fits 31 bits)
    }
    return s;
}
// k_encode_fused's own range of WaveformLengths (uniform batches), a RiceParameter beyond the measured geometry, and a waveform
// that would not stay whole in the standard 2048-word buffer: larger buffers, fewer waveforms per workgroup --
// 1 = 8 waveforms x 2496 words, 2 = 4 x 3072, 3 = 4 x 4096 -- where the waveform fits at k + 4.4 bits per sample; 0 = the
// standard geometry (or, beyond all of them, the pieces encoder's segments).  100 chunks of 2000 x 7000, sigma = 80, m = 64
// (9.5 bits per sample): coded twice 2.17 ms, two segments 1.87, 4 x 3072 1.56, 8 x 3072 (one workgroup per CU) 1.64, 6 x 3072 2.08.
constexpr uint32_t kEncWide1Words = 2496, kEncWide2Words = 3072, kEncWide3Words = 4096;
inline int fused_wide(const Geom &G) {
    if (!G.uniform || G.k <= 3u || (G.dbg & 65536u)) return 0;
    const uint32_t L = G.u_wave_len;
    if (L <= kPcRunSamples / 2u || L > kPcWholeLen || L <= pc_whole_len(G.k)) return 0;
    const uint64_t bits10 = (uint64_t)L * (10u * G.k + 44u);
    return bits10 <= 320ull * kEncWide1Words ? 1 : (bits10 <= 320ull * kEncWide2Words ? 2 : (bits10 <= 320ull * kEncWide3Words ? 3 : 0));
}
bool pieces_batch(const Geom &G);       // the batch takes this encoder
uint64_t pieces_workgroups(const Geom &G, const ChunkDesc *host_chunks);
uint64_t pieces_scan_words(const Geom &G, uint64_t total_wgs);  // uint64 words of its look-back state
hipError_t launch_encode_pieces(const Geom &G, const int16_t *d_in, uint64_t in_samples, uint32_t *d_out, uint64_t out_cap,
                                uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint64_t *d_scan, uint64_t total_wgs,
                                DevStatus *d_status, hipEvent_t *ev, hipStream_t s);
// in-place inverse of a general prediction filter over decoded residuals (drx_iir.hip): tiles of kIirThreads lanes x kIirRun
// samples, decoupled look-back over kIirWin tiles per poll, tables of kIirTabWords uint32 per filter
#ifndef DRX_IIR_THREADS
#define DRX_IIR_THREADS 512
#endif
#ifndef DRX_IIR_RUN
#define DRX_IIR_RUN 64
#endif
constexpr uint32_t kIirRun = DRX_IIR_RUN, kIirThreads = DRX_IIR_THREADS, kIirTile = kIirRun * kIirThreads, kIirWin = 128;
constexpr uint32_t kIirTabWords = (7 + 64 + (kIirWin + 1)) * 9 + 4;
void iir_tables(const uint32_t fast_nt[3], uint32_t t0neg, uint32_t *tab);
uint64_t iir_tiles(const Geom &G, const ChunkDesc *host_chunks, uint64_t *chunk_tile_base);  // (chunk_tile_base: n_chunks + 1, ragged only)
hipError_t launch_iir(const Geom &G, const uint64_t *d_chunk_tile_base, uint64_t n_tiles, const uint32_t *d_tab, uint64_t *d_state,
                      const uint32_t *d_skip, DevStatus *d_status, int16_t *d_out, hipStream_t s);
uint64_t par_walk_scratch_bytes(const Geom &G);
uint32_t bw_walk_blocks_max(const Geom &G);
constexpr uint32_t kWalkShortLenHost = 2048;  // keep equal to kWalkShortLen in drx_walk.h
// WaveformLengths the segment encoder takes in any batch: up to kSegShortLenHost, and from kSegLongLenHost
constexpr uint32_t kSegShortLenHost = 3072, kSegLongLenHost = 10240;
// limits of the parallel header walks (see k_walk_parallel / k_bw_blocks)
constexpr uint32_t kPwMaxWaves = 3584;   // waveforms per chunk the chunk-wide walk takes (leaves room for impostors)
#ifndef DRX_PW_MAX_CHUNKS
#define DRX_PW_MAX_CHUNKS 224
#endif
constexpr uint64_t kPwMaxChunks = DRX_PW_MAX_CHUNKS;  // the walks that READ the chunks (block-parallel; the chunk-wide walk's scan form,
                                                      // debug flag 8388608): more chunks hide the serial walk behind the decoding
// the chunk-wide walk by chains (k_walk_sparse) costs ~60 us per 512 chunks whatever their size: every uniform batch of
// long waveforms takes it, the headline's 500 chunks included (4.74 + 0.08 ms against 5.33 with the walk inside the launch)
constexpr uint64_t kSwMaxChunks = 1u << 20;
constexpr uint32_t kSwMinWaves = 8, kSwMaxWaves = 8192;  // ... chunks of 8 ... 8192 waveforms (its scan form: 64 ... kPwMaxWaves)

}  // namespace drx
#endif
