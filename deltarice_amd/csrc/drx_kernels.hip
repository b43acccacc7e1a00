// drx_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the Delta-Rice codec.
//
// Format contract (bit-exact with /root/reference/src/deltaRice.c):
//   chunk   := u32 N | { u32 n_i | u32 payload_i[n_i] }            (:415,379,427-433)
//   payload := MSB-first concatenation of one code per sample      (:229-241)
//   code(z) := (z>>k) zeros, '1', k bits      if (z>>k) < 8        (:215-222)
//              8 zeros, '1', 16 bits of z     otherwise            (:223-228)
//   z = zigzag(d), d_0 = x_0, d_j = x_j - x_{j-1} mod 2^16         (:51-63,207-211)
//
// Work decomposition (64-wide wavefronts, no MFMA: this is integer bit packing
// bounded by HBM bandwidth):
//   encode  one wavefront per waveform: 16-byte coalesced loads (8 samples/lane),
//           wave prefix scan of code lengths -> bit offsets, codes OR-ed into an LDS
//           staging tile, full words streamed out coalesced.
//   decode  the Rice parse is serial inside a waveform, so the unit of parallelism is
//           the waveform: one lane per waveform, 64 waveforms per wavefront; the 64
//           compressed streams are staged into per-lane LDS rings by coalesced
//           128-byte line loads and the decoded samples are transposed through LDS so
//           that HBM sees 16-byte stores of whole 128-byte runs per waveform.
//   walk    the only way to find waveform i+1 is the chained length header of
//           waveform i (:320-325); one lane per chunk chases it and validates it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "drx_internal.h"
#include "drx_device.h"
#include "drx_encode.h"

namespace drx {

// ---------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------
constexpr int kStageWords = 416;  // 512 * 25 bits / 32 = 400 words worst case, + carry word + slack

// Loads this lane's 8 consecutive samples of the tile starting at t0; returns how
// many of them exist.  vec_ok: the waveform starts on a 16-byte boundary.
__device__ __forceinline__ int load8(const int16_t *__restrict__ x, uint32_t len, uint32_t t0, int lane,
                                     bool vec_ok, int32_t v[8]) {
    const uint32_t i0 = t0 + 8u * (uint32_t)lane;
    const int nv = (i0 >= len) ? 0 : (int)((len - i0) < 8u ? (len - i0) : 8u);
    if (vec_ok && nv == 8) {
        const uint4 q = *reinterpret_cast<const uint4 *>(x + i0);
        v[0] = (int16_t)(q.x & 0xffffu); v[1] = (int16_t)(q.x >> 16);
        v[2] = (int16_t)(q.y & 0xffffu); v[3] = (int16_t)(q.y >> 16);
        v[4] = (int16_t)(q.z & 0xffffu); v[5] = (int16_t)(q.z >> 16);
        v[6] = (int16_t)(q.w & 0xffffu); v[7] = (int16_t)(q.w >> 16);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (j < nv) ? (int32_t)x[i0 + j] : 0;
    }
    return nv;
}

// residual -> (code bits without the leading zeros, total code length)
__device__ __forceinline__ void rice_code(int32_t d, uint32_t k, uint32_t &code, uint32_t &nbits) {
    const uint32_t z = (uint32_t)((d << 1) ^ (d >> 31));  // zig-zag, 0..65535 (:207-211)
    const uint32_t q = z >> k;
    const bool esc = q >= 8u;                              // "giveup" (:203,215)
    nbits = esc ? 25u : q + 1u + k;
    code = esc ? (0x10000u | z) : ((1u << k) | (z & ((1u << k) - 1u)));
}

// General prediction filter, forward (src/deltaRice.c:64-74): d[i] = sum_j taps[j] * x[i-j] over the
// samples that exist, every partial sum truncated to int16 -- i.e. the sum mod 2^16.
__device__ __forceinline__ int32_t fir_residual(const int16_t *__restrict__ x, uint32_t i, const Geom &G) {
    uint32_t acc = 0;
    for (uint32_t t = 0; t < G.n_taps && t <= i; ++t) acc += (uint32_t)((int32_t)x[i - t] * G.taps[t]);
    return (int32_t)(int16_t)(uint16_t)acc;
}

// Pass A: payload word count n_i of every waveform.
__global__ __launch_bounds__(256) void k_encode_sizes(Geom G, const int16_t *__restrict__ in,
                                                      uint32_t *__restrict__ wave_words) {
    const int lane = lane_id();
    const uint64_t g = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= G.total_waves) return;
    const WaveRef r = locate(G, g);
    const int16_t *x = in + r.sample_off;
    const bool vec_ok = ((uintptr_t)x & 15u) == 0;
    const uint32_t k = G.k;
    uint32_t bits = 0;  // per lane: <= len/64*25 + 200, fits
    int32_t carry = 0;  // x[-1] := 0 so that d_0 = x_0 (:53-54)
    for (uint32_t t0 = 0; t0 < r.len; t0 += kTile) {
        int32_t v[8];
        const int nv = load8(x, r.len, t0, lane, vec_ok, v);
        int32_t prev = __shfl_up(v[7], 1);
        if (lane == 0) prev = carry;
        carry = __shfl(v[7], 63);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int32_t d = (int32_t)(int16_t)(v[j] - prev);
            prev = v[j];
            if (G.n_taps) d = (j < nv) ? fir_residual(x, t0 + 8u * (uint32_t)lane + (uint32_t)j, G) : 0;
            uint32_t code, nb;
            rice_code(d, k, code, nb);
            bits += (j < nv) ? nb : 0u;
        }
    }
    const uint64_t total = wave_sum_u64(bits);
    if (lane == 0) wave_words[g] = (uint32_t)((total + 31u) >> 5);
}

// RiceParameter optimiser (the routine docs/Optimization.md:5-19 describes but the reference does not
// ship): exact size of the encoded batch for every k = 0..15 in one pass over the samples.
// words[k] += 1 + ceil(bits_k / 32) per waveform (+1 per chunk); one wavefront per waveform.
__global__ __launch_bounds__(256) void k_estimate_words(Geom G, const int16_t *__restrict__ in,
                                                        unsigned long long *__restrict__ words) {
    const int lane = lane_id();
    const uint64_t g = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= G.total_waves) return;
    const WaveRef r = locate(G, g);
    const int16_t *x = in + r.sample_off;
    const bool vec_ok = ((uintptr_t)x & 15u) == 0;
    uint32_t bits[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) bits[k] = 0;
    int32_t carry = 0;
    for (uint32_t t0 = 0; t0 < r.len; t0 += kTile) {
        int32_t v[8];
        const int nv = load8(x, r.len, t0, lane, vec_ok, v);
        int32_t prev = __shfl_up(v[7], 1);
        if (lane == 0) prev = carry;
        carry = __shfl(v[7], 63);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int32_t d = (int32_t)(int16_t)(v[j] - prev);
            prev = v[j];
            if (G.n_taps) d = (j < nv) ? fir_residual(x, t0 + 8u * (uint32_t)lane + (uint32_t)j, G) : 0;
            const uint32_t z = (uint32_t)((d << 1) ^ (d >> 31));
            if (j < nv) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t q = z >> k;
                    bits[k] += q < 8u ? q + 1u + (uint32_t)k : 25u;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint64_t total = wave_sum_u64(bits[k]);
        if (lane == 0) atomicAdd(words + k, (unsigned long long)(1u + ((total + 31u) >> 5) + (r.idx == 0 ? 1u : 0u)));
    }
}

// Per chunk: position of each waveform's header word relative to the chunk start
// (1 + exclusive prefix of (1 + n_i)) and the chunk's total word count.
__global__ __launch_bounds__(256) void k_chunk_scan(Geom G, const uint32_t *__restrict__ wave_words,
                                                    uint32_t *__restrict__ wave_rel,
                                                    uint64_t *__restrict__ chunk_words) {
    __shared__ uint32_t wsum[4];
    const uint64_t c = blockIdx.x;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint64_t base;
    uint32_t W;
    if (G.uniform) { base = c * G.u_n_waves; W = G.u_n_waves; }
    else { base = G.chunks[c].wave_base; W = G.chunks[c].n_waves; }
    uint64_t run = 1;  // the chunk header word
    for (uint32_t i0 = 0; i0 < W; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t v = (i < W) ? wave_words[base + i] + 1u : 0u;
        const uint32_t inc = wave_incl_scan_u32(v, lane);
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { before += (w < wv) ? wsum[w] : 0u; all += wsum[w]; }
        if (i < W) wave_rel[base + i] = (uint32_t)(run + before + inc - v);
        run += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) chunk_words[c] = run;
}

// Side-band decode (drx_decode_with_wave_words): the caller hands over the n_i table an encode left behind (SURVEY section 7:
// "reuse its offset table as a side-band"), so no header chain is walked.  Per chunk: header positions by a prefix sum over
// 1 + n_i, each checked against the stream itself (the word at that position must BE n_i, n_i within the bounds of its
// waveform, the chain must end exactly at the chunk's end, the chunk header must be the sample count) -- a table that does
// not belong to the stream is DRX_ERR_CORRUPT, never a wild read.
__global__ __launch_bounds__(256) void k_sideband_tables(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                         const uint64_t *__restrict__ chunk_word_off,
                                                         const uint32_t *__restrict__ n_in, uint64_t *__restrict__ wave_off,
                                                         uint32_t *__restrict__ wave_words, DevStatus *st) {
    __shared__ uint32_t wsum[4];
    const uint64_t c = blockIdx.x;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint64_t base;
    uint32_t W, L, N;
    if (G.uniform) { base = c * G.u_n_waves; W = G.u_n_waves; L = G.u_wave_len; N = G.u_n_samples; }
    else { const ChunkDesc d = G.chunks[c]; base = d.wave_base; W = d.n_waves; L = d.wave_len; N = d.n_samples; }
    const uint64_t off0 = chunk_word_off[c], off1 = chunk_word_off[c + 1];
    bool bad = off1 > in_words || off0 >= off1;
    uint64_t run = 1;  // the chunk header word
    for (uint32_t i0 = 0; i0 < W; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t n = (i < W) ? n_in[base + i] : 0u;
        const uint32_t v = (i < W) ? n + 1u : 0u;
        const uint32_t inc = wave_incl_scan_u32(v, lane);
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { before += (w < wv) ? wsum[w] : 0u; all += wsum[w]; }
        if (i < W) {
            const uint64_t at = off0 + run + before + inc - v;
            const uint32_t len = (i + 1u == W) ? N - i * L : L;
            if (n > max_payload_words(len) || n < min_payload_words(len, G.k) || at + 1u + n > off1 || bad) bad = true;
            else if (in[at] != n) bad = true;
            wave_off[base + i] = bad ? off0 : at;  // (a rejected table is never dereferenced: the launch behind this is skipped on error)
            wave_words[base + i] = bad ? 0u : n;
        }
        run += all;
        __syncthreads();
    }
    if (threadIdx.x == 0 && !bad && (off0 + run != off1 || in[off0] != N)) bad = true;
    if (bad) atomicOr(&st->err, kErrCorrupt);
}

// Exclusive prefix over chunk totals -> chunk_word_off[0..n_chunks]; one workgroup.
__global__ __launch_bounds__(1024) void k_chunk_offsets(uint64_t n_chunks, const uint64_t *__restrict__ chunk_words,
                                                        uint64_t *__restrict__ chunk_word_off,
                                                        uint64_t out_cap, DevStatus *st) {
    __shared__ uint64_t wsum[16];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint64_t run = 0;
    for (uint64_t i0 = 0; i0 < n_chunks; i0 += 1024) {
        const uint64_t i = i0 + threadIdx.x;
        const uint64_t v = (i < n_chunks) ? chunk_words[i] : 0;
        uint64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint64_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { before += (w < wv) ? wsum[w] : 0; all += wsum[w]; }
        if (i < n_chunks) chunk_word_off[i] = run + before + inc - v;
        run += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        chunk_word_off[n_chunks] = run;
        st->total_words = run;
        if (run > out_cap) atomicOr(&st->err, kErrCapacity);
    }
}

// Pass B: encode and write every waveform at its final position.
__global__ __launch_bounds__(256) void k_encode_pack(Geom G, const int16_t *__restrict__ in,
                                                     const uint32_t *__restrict__ wave_words,
                                                     const uint32_t *__restrict__ wave_rel,
                                                     const uint64_t *__restrict__ chunk_word_off,
                                                     uint32_t *__restrict__ out, uint64_t out_cap) {
    __shared__ uint32_t stage_all[4][kStageWords];
    const int lane = lane_id();
    uint32_t *stage = stage_all[threadIdx.x >> 6];
    const uint64_t g = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= G.total_waves) return;
    for (int i = lane; i < kStageWords; i += 64) stage[i] = 0;
    const WaveRef r = locate(G, g);
    const uint64_t pos = chunk_word_off[r.chunk] + wave_rel[g];  // this waveform's header word
    const uint32_t n = wave_words[g];
    if (pos + 1u + n > out_cap) return;  // k_chunk_offsets has raised kErrCapacity
    if (lane == 0) {
        out[pos] = n;                                // :379
        if (r.idx == 0) out[pos - 1] = r.n_samples;  // chunk header, :415
    }
    uint32_t *__restrict__ outp = out + pos + 1;
    const int16_t *x = in + r.sample_off;
    const bool vec_ok = ((uintptr_t)x & 15u) == 0;
    const uint32_t k = G.k;
    uint64_t P = 0;  // bits emitted so far (wave uniform)
    int32_t carry = 0;
    wave_sync();
    for (uint32_t t0 = 0; t0 < r.len; t0 += kTile) {
        int32_t v[8];
        const int nv = load8(x, r.len, t0, lane, vec_ok, v);
        int32_t prev = __shfl_up(v[7], 1);
        if (lane == 0) prev = carry;
        carry = __shfl(v[7], 63);
        uint32_t code[8], nb[8];
        uint32_t lane_bits = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int32_t d = (int32_t)(int16_t)(v[j] - prev);
            prev = v[j];
            if (G.n_taps) d = (j < nv) ? fir_residual(x, t0 + 8u * (uint32_t)lane + (uint32_t)j, G) : 0;
            rice_code(d, k, code[j], nb[j]);
            if (j >= nv) nb[j] = 0;
            lane_bits += nb[j];
        }
        const uint32_t incl = wave_incl_scan_u32(lane_bits, lane);
        const uint32_t tile_bits = __shfl(incl, 63);
        const uint64_t w0 = P >> 5;                                   // first staged word
        uint32_t p = (uint32_t)(P & 31u) + (incl - lane_bits);        // bit position relative to word w0
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (nb[j]) {
                const uint32_t rw = p >> 5, b = p & 31u;
                const int s = 32 - (int)b - (int)nb[j];
                if (s >= 0) {
                    atomicOr(&stage[rw], code[j] << s);
                } else {
                    const uint32_t hi = code[j] >> (-s);
                    if (hi) atomicOr(&stage[rw], hi);
                    atomicOr(&stage[rw + 1], code[j] << (32 + s));
                }
                p += nb[j];
            }
        }
        P += tile_bits;
        wave_sync();
        const uint32_t nfull = (uint32_t)((P >> 5) - w0);
        for (uint32_t i = lane; i < nfull; i += 64) {
            outp[w0 + i] = stage[i];
            stage[i] = 0;
        }
        wave_sync();
        if (nfull && lane == 0) {  // the partly filled word becomes word 0 of the next tile
            const uint32_t c = stage[nfull];
            stage[nfull] = 0;
            stage[0] = c;
        }
        wave_sync();
    }
    if ((P & 31u) && lane == 0) outp[P >> 5] = stage[0];  // last word left aligned, zero padded (:237-241)
}

// ---------------------------------------------------------------------------
// encode, single pass
// ---------------------------------------------------------------------------
// One wavefront per waveform, one read of the input, one write of the output:
//   tile loop   8 consecutive samples per lane (one 16-byte load), residuals / zig-zag /
//               code lengths in packed 16-bit math (v_pk_*: two samples per instruction),
//               DPP prefix scan of the lanes' bit counts -> bit offset of every code,
//               codes OR-ed (ds_or_b32) into the waveform's own LDS buffer;
//   look-back   the waveform's size n_i is known only now; its position in the packed
//               output is the prefix sum over all earlier waveforms (src/deltaRice.c:427-432
//               does this with a serial memcpy loop).  Decoupled look-back over 8-byte
//               {status, value} words, one per waveform, written and polled with agent-scope
//               relaxed atomics (the word is its own flag).  Waveform indices are handed
//               out by an atomic ticket, so every predecessor a wave may wait for is already
//               running: no dependence on dispatch order or placement;
//   copy out    LDS -> HBM, 256 contiguous bytes per store instruction.
// A waveform whose code does not fit the LDS buffer (incompressible data, very long
// waveforms) finishes the size count without emitting, does the same look-back, and is then
// re-encoded tile by tile straight to its final position (second read of its samples).
// (the packed tile code: drx_encode.h)

constexpr int kEncWaves = 8;  // waveforms (wavefronts) per workgroup = per ticket
#ifndef DRX_ENC_LB_WIN
#define DRX_ENC_LB_WIN 2
#endif
#ifndef DRX_ENC_GATE_SLEEP
#define DRX_ENC_GATE_SLEEP 8
#endif
constexpr int kLbWin = DRX_ENC_LB_WIN;  // look-back window of k_encode_fused in units of 64 entries

#ifndef DRX_ENC_WAVES_PER_EU
#define DRX_ENC_WAVES_PER_EU 1
#endif
template <bool GEN>
__global__ __launch_bounds__(64 * kEncWaves, DRX_ENC_WAVES_PER_EU) void k_encode_fused(Geom G, const int16_t *__restrict__ in,
                                                      uint32_t *__restrict__ out, uint64_t out_cap,
                                                      uint64_t *__restrict__ chunk_word_off,
                                                      uint32_t *__restrict__ wave_words,
                                                      uint64_t *__restrict__ scan_state,
                                                      uint32_t *__restrict__ ticket, DevStatus *st) {
    // per waveform: 4 pad words (emit_tile_concat ORs zeros below a lane's first word), the code, 4 slack words
    __shared__ __attribute__((aligned(16))) uint32_t buf_all[kEncWaves][kEncCapWords + 8];
    __shared__ uint32_t s_ticket;
    __shared__ uint64_t s_mine[kEncWaves];
    __shared__ uint64_t s_excl;
    const int lane = lane_id();
    uint32_t *row = buf_all[threadIdx.x >> 6];
    uint32_t *buf = row + 4;
    const uint32_t buf_bits = lds_addr(buf) * 8u;  // LDS is 160 KB: bit addresses fit easily

    // Waveform indices by ticket: every lower index is already owned by a running (or finished)
    // wave.  One ticket per workgroup of kEncWaves waveforms: a single global counter serves
    // about 88 atomics per microsecond (a ticket per waveform made the whole kernel run at
    // exactly that rate: 1M waveforms in 11.9 ms).
    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint64_t g = (uint64_t)s_ticket * kEncWaves + (threadIdx.x >> 6);
    const bool live = g < G.total_waves;  // the last workgroup may be partial; its idle waves still join the barriers

    for (int i = lane; i < (int)(kEncCapWords + 8) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
    WaveRef r = locate(G, live ? g : 0);
    if (!live) { r.len = 0; r.idx = 1; }  // an idle wave of the last workgroup: nothing to encode, nothing to add
    const int16_t *x = in + r.sample_off;
    // 16-byte loads at any int16 alignment (unaligned access is on for HSA queues): a WaveformLength like 3500 puts
    // every other waveform 8 bytes off a 16-byte boundary, an odd one 2 bytes off a dword, and the per-sample
    // fallback is 2x slower
    const bool vec_ok = true;
    const uint32_t k = G.k;
    wave_sync();

    // ---- pass over the samples: emit into LDS while it fits, count bits always ----
    // The next tile's load is issued before the current tile is processed, so that the HBM
    // round trip (PMC: 76 % of the wave cycles were s_waitcnt without this) overlaps the packing.
    uint64_t P = 0;        // bits so far (wave uniform)
    bool fits = true;      // everything so far is in buf (wave uniform)
    uint32_t carry = 0;    // dword whose high half is the sample before the tile (x[-1] := 0, :53-54)
    uint32_t carry2 = 0;   // GEN: the dword before that one (samples -4, -3)
    const u16x2 tp[4] = {splat(GEN ? G.enc_t[0] : 1u), splat(GEN ? G.enc_t[1] : 0xffffu), splat(GEN ? G.enc_t[2] : 0u),
                         splat(GEN ? G.enc_t[3] : 0u)};
    // Full tiles run in a loop without any masking, with the next tile's 16-byte load in flight
    // while the current one is packed; the trailing partial tile (if any) takes the masked path once.
    auto process_tile = [&](const uint32_t (&w)[4], int nv, auto full_tag) {
        constexpr bool FULLT = decltype(full_tag)::value;
        uint32_t xprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], 0x138, 0xf, 0xf, false);  // wave_shr:1
        if (lane == 0) xprev = carry;
        carry = (uint32_t)__builtin_amdgcn_readlane((int)w[3], 63);
        uint32_t xprev2 = 0;
        if (GEN) {
            xprev2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[2], 0x138, 0xf, 0xf, false);  // wave_shr:1
            if (lane == 0) xprev2 = carry2;
            carry2 = (uint32_t)__builtin_amdgcn_readlane((int)w[2], 63);
        }
        PackedCodes c;
        packed_codes<GEN>(w, xprev, xprev2, tp, k, c);
        if (!FULLT) mask_tail(c, nv);
        const uint32_t lane_bits = lane_tile_bits(c);
        uint32_t cw[4];
        if (FULLT) concat_codes(c, cw);  // independent of the scan: fills its DPP wait states
        const uint32_t incl = wave_incl_scan_dpp(lane_bits);
        const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (kAblate && (G.dbg & 32u)) {  // ablation: no emission
        } else if (fits && ((P + tile_bits + 31u) >> 5) < (uint64_t)kEncCapWords) {
            if (FULLT && !(kAblate && (G.dbg & 16u)) && !__any(lane_bits > 128u))
                place_words(cw, buf_bits + (uint32_t)P + incl);
            else
                emit_tile<FULLT>(c, buf_bits + (uint32_t)P + incl - lane_bits);
        } else {
            fits = false;
        }
        P += tile_bits;
    };
    // r.len is the same in every lane (one waveform per wave): say so, or the tile loop is compiled with
    // per-lane predicates, register copies and a full vmcnt(0) in front of every tile
    const uint32_t wlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.len);
    const uint32_t n_full = wlen / kTile;
    {
        // kDepth tiles of loads in flight, in kDepth fixed register sets (the loop is unrolled by
        // kDepth so that no loaded-but-not-yet-arrived register is ever copied): with 4 waves per
        // SIMD a tile takes ~2.6 K cycles of wall time, less than one HBM round trip under load.
        constexpr int kDepth = 3;
        const uint4 *xv = reinterpret_cast<const uint4 *>(x) + lane;  // tile t: xv[64 * t]
        uint4 q[kDepth];
        uint32_t t = 0;
        if (vec_ok) {
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                q[u] = make_uint4(0, 0, 0, 0);
                if ((uint32_t)u < n_full) q[u] = xv[64 * (size_t)u];
            }
            // while every register set has a successor tile: consume a set, then refill it -- no predicate
            // on the load, so no copy of a set and a plain vmcnt(kDepth - 1) in front of each tile
#pragma unroll 1
            for (; t + 2u * kDepth <= n_full; t += kDepth) {
#pragma unroll
                for (int u = 0; u < kDepth; ++u) {
                    const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                    process_tile(w, 8, std::true_type{});
                    q[u] = xv[64 * (size_t)(t + u + kDepth)];
                }
            }
            // drain: the last kDepth..2 kDepth - 1 full tiles
#pragma unroll 1
            for (; t < n_full; t += kDepth) {
#pragma unroll
                for (int u = 0; u < kDepth; ++u) {
                    if (t + (uint32_t)u < n_full) {
                        const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                        process_tile(w, 8, std::true_type{});
                        if (t + (uint32_t)u + kDepth < n_full) q[u] = xv[64 * (size_t)(t + u + kDepth)];
                    }
                }
            }
            t = n_full;
        }
        // unaligned waveforms, and the trailing partial tile
        for (uint32_t t0 = t * kTile; t0 < wlen; t0 += kTile) {
            uint32_t w[4];
            const int nv = load8_dwords(x, wlen, t0, lane, vec_ok, w);
            process_tile(w, nv, std::false_type{});
        }
    }
    const uint32_t n = (uint32_t)((P + 31u) >> 5);  // payload words n_i
    wave_sync();

    // ---- position of this waveform ----
    // Prefix sum over v_j = 1 + n_j (+1 for a chunk's first waveform).  The kEncWaves waveforms of
    // this workgroup are summed through LDS; wave 0 then runs a decoupled look-back over ONE entry
    // per workgroup, 128 entries per poll.  (One entry per waveform and 64 per poll capped the whole
    // encoder at ~64 waveforms per memory round trip, i.e. ~9 ms for 1M waveforms: a predecessor's
    // prefix is published one round trip after its aggregate, so the frontier of known prefixes
    // advances by at most one window per round trip.)
    const uint64_t mine = live ? 1ull + n + (r.idx == 0 ? 1ull : 0ull) : 0ull;
    const int wv = threadIdx.x >> 6;
    if (lane == 0) s_mine[wv] = mine;
    __syncthreads();
    if (wv == 0) {
        uint64_t block_sum = 0;
#pragma unroll
        for (int i = 0; i < kEncWaves; ++i) block_sum += s_mine[i];
        const uint64_t T = s_ticket;
        uint64_t excl_blk = 0;
        if (T == 0 || (kAblate && (G.dbg & 128u))) {  // dbg 128: ablation, no look-back (positions are wrong)
            if (lane == 0) __hip_atomic_store(scan_state + T, kScanPrefix | block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (T) excl_blk = T * 2048ull * kEncWaves;
        } else {
            if (lane == 0) __hip_atomic_store(scan_state + T, kScanAgg | block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t base = (int64_t)T - 1;
            uint32_t spins = 0;
#ifndef DRX_ENC_NO_GATE
            // Wait for the NEAREST predecessor alone first: one 8-byte load per poll instead of a whole window from every
            // waiting workgroup.  Workgroups finish roughly in ticket order, so when T - 1 has published, the window behind it
            // has too; and ~500 workgroups polling 128 entries each were a fabric load of their own beside the encoder's
            // streaming reads (agent-scope loads are served by the memory side, not by L2).
            for (;;) {
                uint64_t v = 0;
                if (lane == 0) v = __hip_atomic_load(scan_state + base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 62)) != 0) break;
                __builtin_amdgcn_s_sleep(DRX_ENC_GATE_SLEEP);
                if (++spins > (1u << 22)) break;  // (the window loop below reports it)
            }
#endif
            // kLbWin x 64 entries per poll.  The frontier of known prefixes advances one window per hop (a hop = an
            // agent-scope store becoming visible + an agent-scope load, 3-5 us under the encoder's own streaming loads), so
            // the window bounds the rate of the whole kernel: 128 entries carried ~25 workgroups per microsecond, just what
            // 1M waveforms in 6 ms need (round 3: without the look-back the kernel took 4.5 ms instead of 6.0)
            for (;;) {
                // lane l looks at predecessors base - 64 j - l, j = 0 (nearest) .. kLbWin - 1
                uint64_t sv[kLbWin];
                int fp = 64 * kLbWin;          // position of the nearest prefix in the window (0 = nearest predecessor)
                bool hole = false;             // an entry nearer than that prefix has not been published yet
#pragma unroll
                for (int j = 0; j < kLbWin; ++j) {
                    const int64_t i = base - 64 * j - lane;
                    sv[j] = kScanPrefix;  // before the first workgroup: an empty prefix
                    if (i >= 0) sv[j] = __hip_atomic_load(scan_state + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < kLbWin; ++j) {
                    const uint32_t stj = (uint32_t)(sv[j] >> 62);
                    const uint64_t pj = __ballot(stj == 2u), zj = __ballot(stj == 0u);
                    if (fp == 64 * kLbWin) {  // no prefix found in the nearer groups
                        const int f = pj ? __builtin_ctzll(pj) : 64;
                        const uint64_t nearer = f >= 64 ? ~0ull : ((1ull << f) - 1ull);
                        hole = hole || (zj & nearer) != 0;
                        if (pj) fp = 64 * j + f;
                    }
                }
                if (hole) {  // a nearer predecessor has not published yet
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) {  // cannot happen with a zeroed scan_state; never hang the GPU
                        if (lane == 0) atomicOr(&st->err, kErrInternal);
                        break;
                    }
                    continue;
                }
                uint64_t c = 0;
#pragma unroll
                for (int j = 0; j < kLbWin; ++j) c += (64 * j + lane <= fp) ? (sv[j] & kScanValMask) : 0ull;
                excl_blk += wave_sum_u64(c);
                if (fp < 64 * kLbWin) break;
                base -= 64 * kLbWin;
            }
            if (lane == 0)
                __hip_atomic_store(scan_state + T, kScanPrefix | (excl_blk + block_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (kAblate && (G.dbg & 1024u)) excl_blk = T * 2048ull * kEncWaves;  // ablation: look-back done, sparse placement all the same
        if (lane == 0) s_excl = excl_blk;
    }
    __syncthreads();
    uint64_t excl = s_excl;
#pragma unroll
    for (int i = 0; i < kEncWaves; ++i) excl += (i < wv) ? s_mine[i] : 0ull;
    if (!live) return;
    const uint64_t pos = excl + (r.idx == 0 ? 1ull : 0ull);  // this waveform's header word
    if (lane == 0) {
        wave_words[g] = n;
        if (r.idx == 0) chunk_word_off[r.chunk] = excl;
        if (g + 1 == G.total_waves) {
            chunk_word_off[G.n_chunks] = excl + mine;
            st->total_words = excl + mine;
            if (excl + mine > out_cap) atomicOr(&st->err, kErrCapacity);
        }
    }
    if (pos + 1u + n > out_cap) return;  // the last waveform raises kErrCapacity
    if (lane == 0) {
        out[pos] = n;                               // :379
        if (r.idx == 0) out[pos - 1] = r.n_samples;  // chunk header, :415
    }
    uint32_t *__restrict__ outp = out + pos + 1;
    if (fits) {
        if (!(kAblate && (G.dbg & 64u))) {
            // 16 bytes per lane: 1 KB per store instruction instead of 256 bytes.  The waveform's place in the stream is
            // only word aligned; unaligned vector stores are on for HSA queues, and a wavefront's 64 pieces are contiguous
            // whatever their alignment (round 3: 22 store instructions per waveform became 6)
            typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
            typedef u32x4s __attribute__((address_space(1), aligned(4))) g_u32x4_a4;
            const uint32_t n4 = n & ~3u;
            for (uint32_t i = 4u * (uint32_t)lane; i < n4; i += 256u) {
                const uint4 v = *reinterpret_cast<const uint4 *>(buf + i);
                *(g_u32x4_a4 *)(outp + i) = (u32x4s){v.x, v.y, v.z, v.w};
            }
            if ((uint32_t)lane < n - n4) outp[n4 + (uint32_t)lane] = buf[n4 + (uint32_t)lane];
        }
        return;
    }

    // ---- the code did not fit the LDS buffer: stream it tile by tile to its final position ----
    for (int i = lane; i < (int)(kEncCapWords + 8) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
    wave_sync();
    P = 0;
    carry = 0;
    carry2 = 0;
    for (uint32_t t0 = 0; t0 < r.len; t0 += kTile) {
        uint32_t w[4];
        const int nv = load8_dwords(x, r.len, t0, lane, vec_ok, w);
        uint32_t xprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], 0x138, 0xf, 0xf, false);
        if (lane == 0) xprev = carry;
        carry = (uint32_t)__shfl((int)w[3], 63);
        uint32_t xprev2 = 0;
        if (GEN) {
            xprev2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[2], 0x138, 0xf, 0xf, false);
            if (lane == 0) xprev2 = carry2;
            carry2 = (uint32_t)__shfl((int)w[2], 63);
        }
        PackedCodes c;
        packed_codes<GEN>(w, xprev, xprev2, tp, k, c);
        mask_tail(c, nv);
        const uint32_t lane_bits = lane_tile_bits(c);
        const uint32_t incl = wave_incl_scan_dpp(lane_bits);
        const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint64_t w0 = P >> 5;  // first staged word
        emit_tile<false>(c, buf_bits + (uint32_t)(P & 31u) + incl - lane_bits);
        P += tile_bits;
        wave_sync();
        const uint32_t nfull = (uint32_t)((P >> 5) - w0);
        for (uint32_t i = lane; i < nfull; i += 64) { outp[w0 + i] = buf[i]; buf[i] = 0; }
        wave_sync();
        if (nfull && lane == 0) { const uint32_t cw = buf[nfull]; buf[nfull] = 0; buf[0] = cw; }
        wave_sync();
    }
    if ((P & 31u) && lane == 0) outp[P >> 5] = buf[0];
}

// ---------------------------------------------------------------------------
// encode, few long waveforms: a wavefront per SEGMENT of a waveform
// ---------------------------------------------------------------------------
// The single-pass encoder gives a waveform to one wavefront; with WaveformLength = -1 (the reference's
// default) a chunk is one waveform of millions of samples and 25 chunks keep 25 wavefronts busy.  For
// batches that long_waveform_batch() selects, a waveform is cut into segments of kSegSamples samples:
//   k_seg_sizes    bits of every segment (the packed tile code of the single-pass encoder, no emission);
//   k_seg_scan     per waveform: bit position of every segment, n_i;  then k_chunk_scan / k_chunk_offsets;
//   k_seg_zero     zero the words that two segments share;
//   k_seg_pack     every segment encoded again, streamed tile by tile through a small LDS stage to its final
//                  BIT position; words shared with a neighbour are merged with an atomic OR.
constexpr uint32_t kSegSamples = 16u * kTile;  // 8192 samples per wavefront

struct SegRef {
    WaveRef r;
    uint64_t g;       // waveform
    uint32_t s;       // segment of the waveform
    uint32_t start;   // first sample of the segment inside the waveform
    uint32_t count;   // samples in the segment (0: this unit does not exist)
    bool last;        // last segment of its waveform
};

__device__ __forceinline__ SegRef locate_seg(const Geom &G, uint64_t u, uint32_t segs_per_wave) {
    SegRef q;
    if (G.uniform) {
        q.g = u / segs_per_wave;
        q.s = (uint32_t)(u - q.g * segs_per_wave);
        q.r = locate(G, q.g);
    } else {
        uint64_t lo = 0, hi = G.n_chunks;  // invariant: seg_unit_base[lo] <= u < seg_unit_base[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (G.seg_unit_base[mid] <= u) lo = mid; else hi = mid;
        }
        const ChunkDesc d = G.chunks[lo];
        const uint32_t S = (d.wave_len + kSegSamples - 1u) / kSegSamples;
        const uint64_t local = u - G.seg_unit_base[lo];
        const uint32_t idx = (uint32_t)(local / S);
        q.s = (uint32_t)(local - (uint64_t)idx * S);
        q.g = d.wave_base + idx;
        q.r.chunk = lo;
        q.r.idx = idx;
        q.r.n_samples = d.n_samples;
        q.r.sample_off = d.sample_off + (uint64_t)idx * d.wave_len;
        q.r.len = (idx + 1 == d.n_waves) ? (d.n_samples - idx * d.wave_len) : d.wave_len;
    }
    q.start = q.s * kSegSamples;
    q.count = q.start < q.r.len ? ((q.r.len - q.start) < kSegSamples ? (q.r.len - q.start) : kSegSamples) : 0u;
    q.last = q.start + q.count == q.r.len;
    return q;
}

// the tiles of one segment: calls tile(c, lane_bits, incl, tile_bits, full) for each
template <typename F>
__device__ __forceinline__ void for_segment_tiles(const int16_t *__restrict__ xw, const SegRef &q, uint32_t k, int lane, F &&tile) {
    const int16_t *x = xw + q.start;
    const bool vec_ok = true;  // any int16 alignment (see k_encode_fused)
    // dword whose high half is the sample before the segment (x[-1] := 0 at the start of the waveform, :53-54)
    // (a unit past the end of a shorter last waveform has count == 0: nothing of it may be touched)
    uint32_t carry = (q.start && q.count) ? ((uint32_t)(uint16_t)xw[q.start - 1u] << 16) : 0u;
    const u16x2 tp[4] = {splat(1u), splat(0xffffu), splat(0u), splat(0u)};
    for (uint32_t t0 = 0; t0 < q.count; t0 += kTile) {
        uint32_t w[4];
        const int nv = load8_dwords(x, q.count, t0, lane, vec_ok, w);
        uint32_t xprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], 0x138, 0xf, 0xf, false);  // wave_shr:1
        if (lane == 0) xprev = carry;
        carry = (uint32_t)__builtin_amdgcn_readlane((int)w[3], 63);
        PackedCodes c;
        packed_codes<false>(w, xprev, 0u, tp, k, c);
        const bool full = t0 + kTile <= q.count;
        if (!full) mask_tail(c, nv);
        const uint32_t lane_bits = lane_tile_bits(c);
        const uint32_t incl = wave_incl_scan_dpp(lane_bits);
        const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        tile(c, lane_bits, incl, tile_bits, full);
    }
}

__global__ __launch_bounds__(256) void k_seg_sizes(Geom G, const int16_t *__restrict__ in, uint32_t segs_per_wave,
                                                   uint64_t n_units, uint32_t upw, uint32_t *__restrict__ seg_bits) {
    const int lane = lane_id();
    // upw consecutive units per wavefront: with one-tile waveforms the launch of a wavefront costs as much as its work
    const uint64_t u0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * upw;
    for (uint32_t rep = 0; rep < upw; ++rep) {
        const uint64_t u = u0 + rep;
        if (u >= n_units) return;
        const SegRef q = locate_seg(G, u, segs_per_wave);
        uint32_t bits = 0;  // <= 8192 * 25
        for_segment_tiles(in + q.r.sample_off, q, G.k, lane,
                          [&](const PackedCodes &, uint32_t, uint32_t, uint32_t tile_bits, bool) { bits += tile_bits; });
        if (lane == 0) seg_bits[u] = bits;
    }
}

// one wavefront per waveform: exclusive prefix of its segments' bits (a waveform has < 2^31 * 25 / 2^32 ... bits
// fit 64, positions inside one waveform are kept in 64 bits), n_i
__global__ __launch_bounds__(64) void k_seg_scan(Geom G, uint32_t segs_uniform, const uint32_t *__restrict__ seg_bits,
                                                 uint64_t *__restrict__ seg_pos, uint32_t *__restrict__ wave_words) {
    const int lane = lane_id();
    const uint64_t g = blockIdx.x;
    if (g >= G.total_waves) return;
    uint32_t segs_per_wave = segs_uniform;
    uint64_t first = g * segs_uniform;  // the waveform's first unit
    if (!G.uniform) {
        const WaveRef r = locate(G, g);
        segs_per_wave = (G.chunks[r.chunk].wave_len + kSegSamples - 1u) / kSegSamples;
        first = G.seg_unit_base[r.chunk] + (uint64_t)r.idx * segs_per_wave;
    }
    uint64_t run = 0;
    for (uint32_t s0 = 0; s0 < segs_per_wave; s0 += 64) {
        const uint32_t sidx = s0 + (uint32_t)lane;
        const uint32_t v = sidx < segs_per_wave ? seg_bits[first + sidx] : 0u;
        const uint32_t inc = wave_incl_scan_dpp(v);  // 64 * 204 800 bits fit 32
        if (sidx < segs_per_wave) seg_pos[first + sidx] = run + inc - v;
        run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (lane == 0) wave_words[g] = (uint32_t)((run + 31u) >> 5);
}

__global__ __launch_bounds__(256) void k_seg_zero(Geom G, uint32_t segs_per_wave, uint64_t n_units,
                                                  const uint64_t *__restrict__ seg_pos, const uint32_t *__restrict__ wave_rel,
                                                  const uint64_t *__restrict__ chunk_word_off, uint32_t *__restrict__ out,
                                                  uint64_t out_cap) {
    const uint64_t u = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (u >= n_units) return;
    const SegRef q = locate_seg(G, u, segs_per_wave);
    if (q.count == 0 || q.s == 0) return;
    const uint64_t B = seg_pos[u];
    if ((B & 31u) == 0) return;  // the segment starts on a word boundary: nothing is shared
    const uint64_t w = chunk_word_off[q.r.chunk] + wave_rel[q.g] + 1u + (B >> 5);
    if (w < out_cap) out[w] = 0u;
}

__global__ __launch_bounds__(256) void k_seg_pack(Geom G, const int16_t *__restrict__ in, uint32_t segs_per_wave,
                                                  uint64_t n_units, const uint64_t *__restrict__ seg_pos,
                                                  const uint32_t *__restrict__ wave_words, const uint32_t *__restrict__ wave_rel,
                                                  const uint64_t *__restrict__ chunk_word_off, uint32_t *__restrict__ out,
                                                  uint64_t out_cap, uint32_t upw) {
    // per wave: 4 pad words (place_words ORs zeros below a lane's first word), the stage, slack
    __shared__ __attribute__((aligned(16))) uint32_t stage_all[4][4 + kStageWords + 12];
    const int lane = lane_id();
    uint32_t *row = stage_all[threadIdx.x >> 6];
    uint32_t *stage = row + 4;
    const uint64_t u0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * upw;
    for (uint32_t rep = 0; rep < upw; ++rep) {
        const uint64_t u = u0 + rep;
        if (u >= n_units) return;
        const SegRef q = locate_seg(G, u, segs_per_wave);
        if (q.count == 0) continue;
        for (int i = lane; i < 4 + (int)kStageWords + 12; i += 64) row[i] = 0;
        const uint64_t pos = chunk_word_off[q.r.chunk] + wave_rel[q.g];  // the waveform's header word
        const uint32_t n = wave_words[q.g];
        if (pos + 1u + n > out_cap) continue;  // k_chunk_offsets has raised kErrCapacity
        if (q.s == 0 && lane == 0) {
            out[pos] = n;                                    // :379
            if (q.r.idx == 0) out[pos - 1] = q.r.n_samples;  // chunk header, :415
        }
        const uint64_t B = seg_pos[u];
        uint32_t *__restrict__ outp = out + pos + 1 + (B >> 5);  // the word that holds the segment's first bit
        const uint32_t stage_bits = lds_addr(stage) * 8u;
        uint32_t P = (uint32_t)(B & 31u);  // bits in the stage, counted from the start of outp[wdone]
        uint32_t wdone = 0;                // words of outp already written
        bool shared_first = (B & 31u) != 0;  // outp[0] also holds the end of the previous segment
        wave_sync();
        for_segment_tiles(in + q.r.sample_off, q, G.k, lane,
                          [&](const PackedCodes &c, uint32_t lane_bits, uint32_t incl, uint32_t tile_bits, bool full) {
            if (full && !__any(lane_bits > 128u)) {
                uint32_t cw[4];
                concat_codes(c, cw);
                place_words(cw, stage_bits + P + incl);
            } else {
                emit_tile<false>(c, stage_bits + P + incl - lane_bits);
            }
            P += tile_bits;
            wave_sync();
            const uint32_t nfull = P >> 5;
            for (uint32_t i = lane; i < nfull; i += 64) {
                const uint32_t v = stage[i];
                stage[i] = 0;
                if (i == 0 && shared_first) atomicOr(outp + wdone, v); else outp[wdone + i] = v;
            }
            wave_sync();
            if (nfull) {
                if (lane == 0) { const uint32_t cwd = stage[nfull]; stage[nfull] = 0; stage[0] = cwd; }
                shared_first = false;
                wdone += nfull;
                P &= 31u;
            }
            wave_sync();
        });
        if (P && lane == 0) {
            // the last, partly filled word: the next segment continues in it, unless the waveform ends here
            // (then it is left aligned and zero padded, :237-241)
            if (q.last && !shared_first) outp[wdone] = stage[0]; else atomicOr(outp + wdone, stage[0]);
        }

        wave_sync();
    }
}

// ---------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------

// Header chain walk (:320-325) with validation; one lane per chunk.
// granules (optional): one 8-byte word per waveform, {valid:1 | n_i:31 | header position relative to
// the chunk start:32}, stored with one agent-scope relaxed atomic each -- the word is its own flag
// (it is zero until written), so a decoder wave in the SAME launch can consume waveform w of a
// chunk while the walk of that chunk is still at waveform w+1.
constexpr uint64_t kGranValid = 1ull << 63;

__device__ __forceinline__ void walk_chunk(const Geom &G, uint64_t c, const uint32_t *__restrict__ in,
                                           uint64_t in_words, const uint64_t *__restrict__ chunk_word_off,
                                           uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                           uint64_t *__restrict__ granules, DevStatus *st) {
    uint64_t base;
    uint32_t W, L, N;
    if (G.uniform) { base = c * G.u_n_waves; W = G.u_n_waves; L = G.u_wave_len; N = G.u_n_samples; }
    else { const ChunkDesc d = G.chunks[c]; base = d.wave_base; W = d.n_waves; L = d.wave_len; N = d.n_samples; }
    const uint64_t begin = chunk_word_off[c];
    uint64_t end = chunk_word_off[c + 1];
    bool bad = false;
    if (end > in_words || begin + 2 > end || end - begin > 0xffffffffull) { bad = true; end = begin; }
    if (!bad && in[begin] != N) bad = true;  // :306 totalNumberPoints
    uint64_t at = begin + 1;
    for (uint32_t w = 0; w < W; ++w) {
        uint32_t n = 0;
        uint64_t here = at;
        if (!bad && at < end) {
            n = in[at];
            const uint32_t len = (w + 1 == W) ? (N - w * L) : L;
            if (n > max_payload_words(len) || n < min_payload_words(len, G.k) || at + 1u + n > end) { bad = true; n = 0; }
            else at += (uint64_t)n + 1u;
        } else {
            bad = true;
            here = begin;  // keeps later loads in bounds; decoded as zero words
        }
        wave_off[base + w] = here;
        wave_words[base + w] = n;
        if (granules)
            __hip_atomic_store(granules + base + w, kGranValid | ((uint64_t)n << 32) | (uint64_t)(uint32_t)(here - begin),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!bad && at != end) bad = true;
    if (bad) atomicOr(&st->err, kErrCorrupt);
}

// The same walk for kWalkChains chunks of a uniform batch per wavefront (lanes 0..kWalkChains-1), with the
// header loads issued as SCALAR loads (s_load_dword through the scalar cache): under the decode's
// ~3.6 TB/s of vector traffic a dependent vector load takes ~2.9 us per hop (it queues behind the other
// waves' 64-line gathers in the CU's vector memory pipeline) and the chain becomes the critical path of
// the fused launch; the scalar path does not share that queue.
constexpr int kWalkChains = 8;

__device__ __forceinline__ void walk_chunks_scalar(const Geom &G, uint64_t c0, const uint32_t *__restrict__ list,
                                                   uint64_t n_list, const uint32_t *__restrict__ in,
                                                   uint64_t in_words, const uint64_t *__restrict__ chunk_word_off,
                                                   uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                   uint64_t *__restrict__ granules, DevStatus *st,
                                                   const uint32_t *__restrict__ only = nullptr) {
    // lanes 0..kWalkChains-1 take entries c0.. of the chunk list (list == nullptr: chunk index = entry);
    // only != nullptr: just the chunks it flags (the ones k_walk_parallel gave up on)
    const int lane = lane_id();
    const uint64_t e = c0 + (uint64_t)lane;
    bool mine = lane < kWalkChains && e < n_list;
    const uint64_t c = mine ? (list ? (uint64_t)list[e] : e) : 0;
    if (only && mine && !only[c]) mine = false;
    if (only && !__any(mine)) return;
    uint32_t W = 0, L = 1, N = 0;
    uint64_t base = 0;
    if (mine) {
        if (G.uniform) { W = G.u_n_waves; L = G.u_wave_len; N = G.u_n_samples; base = c * W; }
        else { const ChunkDesc d = G.chunks[c]; W = d.n_waves; L = d.wave_len; N = d.n_samples; base = d.wave_base; }
    }
    const uint32_t W_max = wave_max_u32(W);
    uint64_t begin = 0, end = 0;
    bool bad = false;
    if (mine) {
        begin = chunk_word_off[c];
        end = chunk_word_off[c + 1];
        if (end > in_words || begin + 2 > end || end - begin > 0xffffffffull) { bad = true; end = begin; }
    }
    if (in_words == 0) {  // nothing to load from (every chunk is bad)
        if (mine) {
            for (uint32_t w = 0; w < W; ++w) {
                wave_off[base + w] = begin;
                wave_words[base + w] = 0;
                if (granules) __hip_atomic_store(granules + base + w, kGranValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            atomicOr(&st->err, kErrCorrupt);
        }
        return;
    }
    // word `a` of the stream for lanes 0..kWalkChains-1 (a < in_words), one scalar load per chain
    auto sload = [&](uint64_t a) __attribute__((always_inline)) -> uint32_t {
        static_assert(kWalkChains == 8, "the asm block below issues eight loads");
        uint64_t p[kWalkChains];
#pragma unroll
        for (int i = 0; i < kWalkChains; ++i) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)a, i);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(a >> 32), i);
            p[i] = (uint64_t)(uintptr_t)(in + (((uint64_t)hi << 32) | lo));
        }
        uint32_t v0, v1, v2, v3, v4, v5, v6, v7;
        // one block: all eight loads in flight before the wait (left to itself the compiler waits after seven)
        asm volatile(
            "s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %9, 0x0\n\ts_load_dword %2, %10, 0x0\n\t"
            "s_load_dword %3, %11, 0x0\n\ts_load_dword %4, %12, 0x0\n\ts_load_dword %5, %13, 0x0\n\t"
            "s_load_dword %6, %14, 0x0\n\ts_load_dword %7, %15, 0x0\n\ts_waitcnt lgkmcnt(0)"
            : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3), "=&s"(v4), "=&s"(v5), "=&s"(v6), "=&s"(v7)
            : "s"(p[0]), "s"(p[1]), "s"(p[2]), "s"(p[3]), "s"(p[4]), "s"(p[5]), "s"(p[6]), "s"(p[7])
            : "memory");
        const uint32_t v[kWalkChains] = {v0, v1, v2, v3, v4, v5, v6, v7};
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < kWalkChains; ++i) r = (lane == i) ? v[i] : r;
        return r;
    };
    const uint32_t head = sload((mine && !bad) ? begin : 0ull);
    if (mine && !bad && head != N) bad = true;  // :306 totalNumberPoints
    uint64_t at = begin + 1;
    for (uint32_t w = 0; w < W_max; ++w) {
        const bool live = mine && w < W;  // chunks of a ragged batch differ in their number of waveforms
        const bool can = live && !bad && at < end;
        const uint32_t nn = sload(can ? at : 0ull);
        uint32_t n = 0;
        uint64_t here = at;
        if (can) {
            n = nn;
            const uint32_t len = (w + 1 == W) ? (N - w * L) : L;
            if (n > max_payload_words(len) || n < min_payload_words(len, G.k) || at + 1u + n > end) { bad = true; n = 0; }
            else at += (uint64_t)n + 1u;
        } else if (live) {
            bad = true;
            here = begin;
        }
        if (live) {
            wave_off[base + w] = here;
            wave_words[base + w] = n;
            if (granules)
                __hip_atomic_store(granules + base + w, kGranValid | ((uint64_t)n << 32) | (uint64_t)(uint32_t)(here - begin),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (mine && !bad && at != end) bad = true;
    if (mine && bad) atomicOr(&st->err, kErrCorrupt);
}

__global__ __launch_bounds__(64) void k_walk_scalar(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                    const uint64_t *__restrict__ chunk_word_off,
                                                    uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                    DevStatus *st) {
    walk_chunks_scalar(G, (uint64_t)blockIdx.x * kWalkChains, nullptr, G.n_chunks, in, in_words, chunk_word_off, wave_off,
                       wave_words, nullptr, st);
}

__global__ __launch_bounds__(64) void k_walk_scalar_only(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                         const uint64_t *__restrict__ chunk_word_off,
                                                         uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                         DevStatus *st, const uint32_t *__restrict__ only) {
    walk_chunks_scalar(G, (uint64_t)blockIdx.x * kWalkChains, nullptr, G.n_chunks, in, in_words, chunk_word_off, wave_off,
                       wave_words, nullptr, st, only);
}

// The header chain WITHOUT its 2000 dependent round trips, for batches of a handful of chunks (where nothing hides
// them: 1.7 ms of a 2.2 ms decode).  A length header is a small number (n_i <= 25 bits per sample: 5469 for
// L = 7000) and payload words are Rice-coded bits, which practically never start with 19 zero bits.  So:
//   1. the whole chunk is read once (k_pw_scan, 16 workgroups per chunk) and every word <= that bound becomes a
//      CANDIDATE header (the ~2000 real ones plus a few impostors); one workgroup per chunk sorts them by position;
//   2. candidate i links to the candidate at position pos_i + n_i + 1 (binary search), to END if that is the chunk
//      end, to INVALID if no candidate sits there;
//   3. binary lifting over those links (up[k][i] = 2^k links ahead), then waveform w's header is w links from the
//      candidate at word 1: eleven steps, every waveform in parallel.  Impostors are simply never reached.
// Anything unexpected (too many candidates, a broken link, a chain that does not end at the chunk end) flags the
// chunk, and the scalar-load walker walks -- and judges -- the flagged chunks afterwards.
constexpr int kPwThreads = 1024;
constexpr uint32_t kPwCap = 4096;      // candidates per chunk
constexpr uint32_t kPwMaxParts = 128;  // slices of a chunk (pw_parts())
constexpr uint32_t kPwStride = kPwCap + kPwMaxParts;  // a chunk's scratch: its candidates, then {first, count} of every slice
constexpr int kPwLevels = 12;
// (kPwMaxWaves, kPwMaxChunks: drx_internal.h)

// workgroups that scan one chunk: enough of them to fill the chip when the chunks are few (one chunk of 2000 x 7000, what an
// H5Z call brings: 128 instead of 16 took the walk from 0.113 to 0.070 ms, k_pw_scan itself 9 us)
__host__ inline uint32_t pw_parts(uint32_t n_chunks) { return n_chunks <= 4u ? 128u : (n_chunks <= 32u ? 32u : 16u); }

// 1. candidates of one slice of a chunk -> the chunk's list in global memory (cand: kPwCap x {pos, val} per chunk,
//    cand_count: one counter per chunk, zeroed before the launch)
__global__ __launch_bounds__(256) void k_pw_scan(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                 const uint64_t *__restrict__ chunk_word_off, const uint32_t *__restrict__ list,
                                                 uint2 *__restrict__ cand, uint32_t *__restrict__ cand_count, uint32_t parts) {
    const uint64_t c = list ? (uint64_t)list[blockIdx.x / parts] : blockIdx.x / parts;  // scratch is indexed by chunk
    const uint32_t part = blockIdx.x % parts, tid = threadIdx.x;
    const uint64_t begin = chunk_word_off[c], end = chunk_word_off[c + 1];
    if (end > in_words || begin + 2 > end || end - begin > 0x7fffffffull) return;  // k_walk_parallel flags the chunk
    const uint32_t len_w = (uint32_t)(end - begin);
    const uint32_t wl = G.uniform ? G.u_wave_len : G.chunks[c].wave_len;
    const uint32_t max_full = (uint32_t)(((uint64_t)wl * 25u + 31u) >> 5);
    uint2 *clist = cand + c * kPwStride;
    // the slice's candidates are collected in LDS and appended with ONE global atomic (2000 atomics on one counter
    // cost 0.2 ms: same-address atomics serialise in the L2)
    __shared__ uint2 s_list[kPwCap / 4];
    __shared__ uint32_t s_n, s_base;
    if (tid == 0) s_n = 0;
    __syncthreads();
    auto consider = [&](uint32_t i, uint32_t v) __attribute__((always_inline)) {
        if (i >= 1u && i < len_w && v <= max_full) {
            const uint32_t k = atomicAdd(&s_n, 1u);
            if (k < kPwCap / 4) s_list[k] = make_uint2(i, v);
        }
    };
    // 16-byte loads, four in flight per thread (a dependent 4-byte load per word made this pass take as long as
    // the serial walk it replaces); quads are aligned, the first one may start below the chunk
    const uint32_t mis = (uint32_t)((((uintptr_t)in >> 2) + begin) & 3u);
    const uint32_t *q0 = in + begin - mis;  // words before `begin` are ignored by consider()
    const uint32_t n_quads = (len_w + mis + 3u) >> 2;
    const bool vec_ok = begin >= mis;
    const uint32_t per = (n_quads + parts - 1u) / parts;
    const uint32_t q_lo = part * per, q_hi = (q_lo + per < n_quads) ? q_lo + per : n_quads;
    constexpr uint32_t U = 4;
    for (uint32_t qb = q_lo + tid; qb < q_hi; qb += 256u * U) {
        uint4 v[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t qi = qb + u * 256u;
            v[u] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            if (qi < q_hi) {
                const uint64_t w0 = begin - mis + 4ull * qi;  // absolute word index of the quad
                if (vec_ok && w0 + 4u <= in_words) {
                    v[u] = *reinterpret_cast<const uint4 *>(q0 + 4ull * qi);
                } else {
                    if (w0 + 0u < in_words && w0 + 0u >= begin) v[u].x = in[w0 + 0u];
                    if (w0 + 1u < in_words && w0 + 1u >= begin) v[u].y = in[w0 + 1u];
                    if (w0 + 2u < in_words && w0 + 2u >= begin) v[u].z = in[w0 + 2u];
                    if (w0 + 3u < in_words && w0 + 3u >= begin) v[u].w = in[w0 + 3u];
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t i0 = 4u * (qb + u * 256u) - mis;  // may wrap below zero for the first quad: consider() rejects
            consider(i0 + 0u, v[u].x);
            consider(i0 + 1u, v[u].y);
            consider(i0 + 2u, v[u].z);
            consider(i0 + 3u, v[u].w);
        }
    }
    __syncthreads();
    const uint32_t n_loc = s_n;
    if (n_loc > kPwCap / 4) {  // more candidates in one slice than a sane chunk has in four: let the serial walker judge
        if (tid == 0) atomicAdd(cand_count + c, kPwCap);
        return;
    }
    if (tid == 0) {
        s_base = atomicAdd(cand_count + c, n_loc);
        clist[kPwCap + part] = make_uint2(s_base, n_loc);  // k_walk_parallel puts the slices in order
    }
    __syncthreads();
    // written in position order inside the slice (rank by counting: a slice holds tens of candidates), so that
    // k_walk_parallel needs no sort
    const uint32_t b0 = s_base;
    for (uint32_t i = tid; i < n_loc; i += 256u) {
        const uint2 e = s_list[i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < n_loc; ++j) r += s_list[j].x < e.x ? 1u : 0u;
        if (b0 + r < kPwCap) clist[b0 + r] = e;
    }
}

__global__ __launch_bounds__(kPwThreads) void k_walk_parallel(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                              const uint64_t *__restrict__ chunk_word_off,
                                                              uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                              uint32_t *__restrict__ fail, const uint32_t *__restrict__ list,
                                                              const uint2 *__restrict__ cand, const uint32_t *__restrict__ cand_count,
                                                              uint32_t parts) {
    __shared__ uint32_t pos[kPwCap];   // candidate positions relative to the chunk start; padding entries sort last
    __shared__ uint32_t val[kPwCap];
    __shared__ uint16_t up[kPwLevels][kPwCap];
    __shared__ uint32_t s_bad, s_start, s_first[kPwMaxParts], s_pre[kPwMaxParts + 1];
    const uint32_t tid = threadIdx.x;
    const uint64_t c = list ? (uint64_t)list[blockIdx.x] : blockIdx.x;
    uint32_t W, L, N;
    uint64_t base;
    if (G.uniform) { W = G.u_n_waves; L = G.u_wave_len; N = G.u_n_samples; base = c * W; }
    else { const ChunkDesc d = G.chunks[c]; W = d.n_waves; L = d.wave_len; N = d.n_samples; base = d.wave_base; }
    const uint64_t begin = chunk_word_off[c];
    const uint64_t end = chunk_word_off[c + 1];
    if (tid == 0) { s_bad = 0; s_start = 0xffffffffu; }
    __syncthreads();
    bool ok = !(end > in_words || begin + 2 > end || end - begin > 0x7fffffffull);
    if (ok && in[begin] != N) ok = false;
    if (!ok) { if (tid == 0) fail[c] = 1u; return; }  // (uniform across the workgroup)
    const uint32_t len_w = (uint32_t)(end - begin);  // words in the chunk
    const uint32_t max_full = (uint32_t)(((uint64_t)L * 25u + 31u) >> 5);
    const uint32_t max_last = (uint32_t)(((uint64_t)(N - (W - 1) * L) * 25u + 31u) >> 5);
    const uint32_t min_full = min_payload_words(L, G.k), min_last = min_payload_words(N - (W - 1) * L, G.k);
    const uint32_t nc = cand_count[c];
    if (nc > kPwCap - 2u || nc < W) { if (tid == 0) fail[c] = 1u; return; }
    // the candidates in position order: k_pw_scan's slices cover the chunk in order and each wrote its own in order, so the
    // slices only have to be put one behind the other (a bitonic sort of 4096 did this before: 78 barrier-separated stages)
    if (tid < 64u) {
        uint32_t run = 0;
        for (uint32_t s0 = 0; s0 < parts; s0 += 64u) {
            const uint32_t sl = s0 + tid;
            uint2 e = make_uint2(0u, 0u);
            if (sl < parts) e = cand[c * kPwStride + kPwCap + sl];
            const uint32_t incl = wave_incl_scan_dpp(e.y);
            if (sl < parts) { s_first[sl] = e.x; s_pre[sl + 1u] = run + incl; }
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (tid == 0) s_pre[0] = 0;
    }
    __syncthreads();
    if (s_pre[parts] != nc) { if (tid == 0) fail[c] = 1u; return; }  // (cannot happen: the slices' counts add up to it)
    uint32_t n_pad = 64u;  // the candidates and the two sentinel nodes
    while (n_pad < nc + 2u) n_pad <<= 1;
    for (uint32_t i = tid; i < n_pad; i += kPwThreads) {
        uint2 e = make_uint2(0xffffffffu, 0u);
        if (i < nc) {
            uint32_t lo = 0, hi = parts;  // invariant: s_pre[lo] <= i < s_pre[hi]
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (s_pre[mid] <= i) lo = mid; else hi = mid; }
            e = cand[c * kPwStride + s_first[lo] + (i - s_pre[lo])];
        }
        pos[i] = e.x;
        val[i] = e.y;
    }
    __syncthreads();
    // 2. links.  Nodes nc (END) and nc + 1 (INVALID) point to themselves.
    const uint32_t END = nc, INV = nc + 1u;
    for (uint32_t i = tid; i < n_pad; i += kPwThreads) {
        uint32_t to = i;  // padding and the two sentinels: self loops
        if (i < nc) {
            const uint64_t target = (uint64_t)pos[i] + val[i] + 1u;
            if (target == len_w) {
                to = END;
            } else if (target > len_w) {
                to = INV;
            } else {
                uint32_t lo = 0, hi = nc;  // first candidate with pos >= target
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (pos[mid] < (uint32_t)target) lo = mid + 1u; else hi = mid; }
                to = (lo < nc && pos[lo] == (uint32_t)target) ? lo : INV;
            }
            if (pos[i] == 1u) s_start = i;  // the first waveform's header follows the chunk header
        }
        up[0][i] = (uint16_t)to;
    }
    __syncthreads();
    // 3. binary lifting
    for (int k = 1; k < kPwLevels; ++k) {
        for (uint32_t i = tid; i < n_pad; i += kPwThreads) up[k][i] = up[k - 1][up[k - 1][i]];
        __syncthreads();
    }
    const uint32_t start = s_start;
    if (start == 0xffffffffu) { if (tid == 0) fail[c] = 1u; return; }
    bool bad = false;
    for (uint32_t w = tid; w < W; w += kPwThreads) {
        uint32_t node = start;
#pragma unroll
        for (int k = 0; k < kPwLevels; ++k)
            if ((w >> k) & 1u) node = up[k][node];
        if (node >= nc) { bad = true; continue; }
        const uint32_t n = val[node];
        if (n > ((w + 1u == W) ? max_last : max_full) || n < ((w + 1u == W) ? min_last : min_full)) { bad = true; continue; }
        if (w + 1u == W && up[0][node] != END) { bad = true; continue; }
        wave_off[base + w] = begin + pos[node];
        wave_words[base + w] = n;
    }
    if (bad) atomicOr(&s_bad, 1u);
    __syncthreads();
    if (tid == 0 && s_bad) fail[c] = 1u;
}

__global__ __launch_bounds__(64) void k_walk_list(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                  const uint64_t *__restrict__ chunk_word_off,
                                                  const uint32_t *__restrict__ chunk_list, uint32_t n_list,
                                                  uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                  DevStatus *st) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n_list) return;
    walk_chunk(G, chunk_list[i], in, in_words, chunk_word_off, wave_off, wave_words, nullptr, st);
}

// Header-chain walk for chunks of SHORT waveforms (one wavefront per chunk).  With n_i of a few
// hundred words a chunk holds tens of thousands of waveforms and the per-hop HBM round trip of
// walk_chunk() adds up to tens of milliseconds (27 343 hops for 14 M samples at L = 512).  Here the
// wave streams the chunk through a 16 KB LDS block with coalesced 16-byte loads and lane 0 chases the
// chain inside LDS (~0.07 us per hop); the price is one extra read of the chunk's stream.
constexpr uint32_t kWalkBlockWords = 4096;
constexpr uint32_t kWalkShortLen = 2048;  // WaveformLength up to which a chunk is walked through LDS

constexpr uint32_t kWalkHopCap = 1024;   // hops buffered in LDS between coalesced flushes

// blk: kWalkBlockWords words (16-byte aligned), hop: kWalkHopCap entries, both in LDS and private to the wave.
__device__ __forceinline__ void walk_chunk_block(const Geom &G, uint64_t c, const uint32_t *__restrict__ in,
                                                 uint64_t in_words, const uint64_t *__restrict__ chunk_word_off,
                                                 uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                 uint64_t *__restrict__ granules, DevStatus *st, uint32_t *blk, uint2 *hop) {
    constexpr uint32_t B = kWalkBlockWords;
    constexpr int NV = B / 256;  // 16-byte loads per lane and block
    const int lane = lane_id();
    uint64_t base;
    uint32_t W, L, N;
    if (G.uniform) { base = c * G.u_n_waves; W = G.u_n_waves; L = G.u_wave_len; N = G.u_n_samples; }
    else { const ChunkDesc d = G.chunks[c]; base = d.wave_base; W = d.n_waves; L = d.wave_len; N = d.n_samples; }
    const uint64_t begin = chunk_word_off[c];
    uint64_t end = chunk_word_off[c + 1];
    bool bad = false;
    if (end > in_words || begin + 2 > end || end - begin > 0xffffffffull) { bad = true; end = begin; }
    if (!bad && in[begin] != N) bad = true;
    const uint32_t max_full = (uint32_t)(((uint64_t)L * 25u + 31u) >> 5);
    const uint32_t max_last = W ? (uint32_t)(((uint64_t)(N - (W - 1) * L) * 25u + 31u) >> 5) : 0u;
    const uint32_t min_full = min_payload_words(L, G.k), min_last = W ? min_payload_words(N - (W - 1) * L, G.k) : 0u;
    // blocks on a fixed grid from g0 (16-byte aligned when the stream is), so that block k + 1 can be
    // requested before the chase through block k starts
    const bool vec_ok = ((uintptr_t)in & 15u) == 0;
    const uint64_t g0 = begin & ~3ull;
    uint64_t at = begin + 1;  // header of waveform w (wave uniform)
    uint32_t w = 0;
    uint4 pre[NV];
    uint64_t pre_b0 = ~0ull;  // block the registers hold
    auto request = [&](uint64_t b0) __attribute__((always_inline)) {
        pre_b0 = b0;
        if (vec_ok && b0 + B <= end) {
#pragma unroll
            for (int j = 0; j < NV; ++j) pre[j] = *reinterpret_cast<const uint4 *>(in + b0 + (uint32_t)(j * 64 + lane) * 4u);
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const uint64_t i = b0 + (uint32_t)(j * 64 + lane) * 4u;
                uint4 v;
                v.x = (i + 0u < end) ? in[i + 0u] : 0u;
                v.y = (i + 1u < end) ? in[i + 1u] : 0u;
                v.z = (i + 2u < end) ? in[i + 2u] : 0u;
                v.w = (i + 3u < end) ? in[i + 3u] : 0u;
                pre[j] = v;
            }
        }
    };
    if (!bad) request(g0);
    while (w < W && !bad && at < end) {
        const uint64_t b0 = g0 + (at - g0) / B * B;
        if (pre_b0 != b0) request(b0);  // a hop longer than a block skipped the requested one
#pragma unroll
        for (int j = 0; j < NV; ++j) *reinterpret_cast<uint4 *>(blk + (uint32_t)(j * 64 + lane) * 4u) = pre[j];
        wave_sync();
        if (b0 + B < end) request(b0 + B);
        const uint32_t blk_len = (end - b0 < B) ? (uint32_t)(end - b0) : B;
        const uint32_t end_rel = (end - b0 < 0xffffffffull) ? (uint32_t)(end - b0) : 0xffffffffu;
        uint32_t rel = (uint32_t)(at - b0);
        while (w < W && rel < blk_len && !bad) {
            // chase up to kWalkHopCap hops inside the block; every value here is wave uniform (SGPRs)
            const uint32_t w0 = w;
            uint32_t hops = 0;
            while (w < W && rel < blk_len && hops < kWalkHopCap) {
                const uint32_t n = __builtin_amdgcn_readfirstlane(blk[rel]);
                const uint32_t lim = (w + 1 == W) ? max_last : max_full, lim_lo = (w + 1 == W) ? min_last : min_full;
                if (n > lim || n < lim_lo || rel + 1u + n > end_rel) { bad = true; break; }
                hop[hops] = make_uint2(rel, n);
                rel += n + 1u;
                ++w;
                ++hops;
            }
            wave_sync();
            for (uint32_t i = lane; i < hops; i += 64) {
                const uint2 h = hop[i];
                wave_off[base + w0 + i] = b0 + h.x;
                wave_words[base + w0 + i] = h.y;
                if (granules)
                    __hip_atomic_store(granules + base + w0 + i,
                                       kGranValid | ((uint64_t)h.y << 32) | (uint64_t)(uint32_t)(b0 + h.x - begin),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wave_sync();
        }
        at = b0 + rel;
    }
    if (w < W) bad = true;
    if (!bad && at != end) bad = true;
    if (bad) {
        for (uint32_t i = w + lane; i < W; i += 64) {
            wave_off[base + i] = begin;
            wave_words[base + i] = 0;
            if (granules) __hip_atomic_store(granules + base + i, kGranValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) atomicOr(&st->err, kErrCorrupt);
    }
}

__global__ __launch_bounds__(64) void k_walk_block(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                   const uint64_t *__restrict__ chunk_word_off,
                                                   const uint32_t *__restrict__ chunk_list, uint32_t n_list,
                                                   uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                   DevStatus *st) {
    __shared__ __attribute__((aligned(16))) uint32_t blk[kWalkBlockWords];
    __shared__ __attribute__((aligned(8))) uint2 hop[kWalkHopCap];  // {position in the block, n}
    if (blockIdx.x >= n_list) return;
    const uint64_t c = chunk_list ? chunk_list[blockIdx.x] : blockIdx.x;
    walk_chunk_block(G, c, in, in_words, chunk_word_off, wave_off, wave_words, nullptr, st, blk, hop);
}

// The same idea for chunks of SHORT waveforms (tens of thousands of headers per chunk: too many for one
// workgroup's LDS, and the chain chase through LDS still costs 0.13 us per hop, 3.6 ms for 14 M samples at
// L = 512).  A header is at most max_words = 25 L / 32 (400 for L = 512) and every B-word block of the stream with
// B > max_words holds at least one, so the BLOCKS become independent: a wavefront loads its block, takes the first word in
// [1, max_words] as the block's entry header, chases the chain through LDS to the block's end (if the chain breaks,
// the entry was an impostor: try the next small word), and reports {entry, headers, exit}.  k_bw_scan checks per
// chunk that every block's exit is the next block's entry (and word 1 / the chunk end at the two ends) and
// turns the counts into first-waveform indices; k_bw_blocks then runs again and writes the table.  A chunk
// that does not stitch is flagged and walked by k_walk_block.
// Launch shape.  How many blocks a chunk really has is only known on the device (chunk_word_off), while the host can only
// bound it by 25 bits per sample, four times the usual: a grid with a workgroup per POSSIBLE block spent most of its time
// on empty workgroups, each holding its LDS for a few microseconds.  So the grid is a fixed number of wavefronts (as many as
// the LDS lets the chip hold) that stride over the REAL blocks, numbered through a prefix sum over the chunks' block
// counts that every wavefront computes for itself (at most kBwMaxList chunks).  B is the smallest of 1024 / 2048 / 4096
// that exceeds max_words: the chase is a chain of dependent LDS reads, so what hides it is wavefronts per CU, i.e.
// little LDS per block.  (A one-pass version -- blocks in ticket order, {headers, exit} through a decoupled look-back,
// table written straight from LDS -- was built and measured SLOWER than the two passes, 0.97 against 0.88 ms on config 5:
// the frontier of known prefixes advances one window of entries per memory round trip; profiles/r02_notes.md.)
constexpr uint32_t kBwTries = 6;  // impostors tolerated in front of a block's first real header
constexpr uint32_t kBwMaxList = 256;  // chunks per launch (bw_walk_blocks_max() / the plan admit at most 224)

struct BwBlock { uint32_t entry, count, exit, base; };

// pre[s] = real blocks of the listed chunks in front of chunk s (pre[n_list] = all); a chunk whose extent is unusable has
// none (k_bw_scan flags it), one longer than the host's bound is cut there (ditto)
template <uint32_t B>
__device__ __forceinline__ void bw_block_prefix(const uint64_t *__restrict__ chunk_word_off, uint64_t in_words,
                                                const uint32_t *__restrict__ list, uint32_t n_list, uint32_t blocks_max,
                                                uint32_t *pre, int lane) {
    uint32_t run = 0;
    for (uint32_t s0 = 0; s0 < n_list; s0 += 64u) {
        const uint32_t sl = s0 + (uint32_t)lane;
        uint32_t nb = 0;
        if (sl < n_list) {
            const uint64_t c = list ? (uint64_t)list[sl] : sl;
            const uint64_t begin = chunk_word_off[c], end = chunk_word_off[c + 1];
            const bool bad = end > in_words || begin + 2 > end || end - begin > 0x7fffffffull;
            if (!bad) nb = (uint32_t)((end - begin + B - 1u) / B);
            if (nb > blocks_max) nb = blocks_max;
        }
        const uint32_t incl = wave_incl_scan_dpp(nb);
        if (sl < n_list) pre[sl + 1u] = run + incl;
        run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) pre[0] = 0;
    wave_sync();
}

template <uint32_t B, bool EMIT, bool LIST = false>
__global__ __launch_bounds__(64) void k_bw_blocks(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                  const uint64_t *__restrict__ chunk_word_off, const uint32_t *__restrict__ list,
                                                  uint32_t n_list, uint32_t blocks_max,
                                                  BwBlock *__restrict__ info, const uint32_t *__restrict__ fail,
                                                  uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                  DevStatus *st, uint32_t *__restrict__ hops, uint32_t hop_cap) {
    constexpr int NV = B / 256;
    __shared__ __attribute__((aligned(16))) uint32_t blk[B];
    __shared__ uint16_t hop[EMIT ? B / 2 : 2];  // header positions inside the block (a waveform has at least one payload word)
    __shared__ uint32_t pre[kBwMaxList + 1];
    const int lane = lane_id();
    bw_block_prefix<B>(chunk_word_off, in_words, list, n_list, blocks_max, pre, lane);
    const uint32_t total = pre[n_list];
    for (uint32_t unit = blockIdx.x; unit < total; unit += gridDim.x) {
        wave_sync();  // (the previous block's LDS reads are done)
        uint32_t lo = 0, hi = n_list;  // invariant: pre[lo] <= unit < pre[hi]
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pre[mid] <= unit) lo = mid; else hi = mid;
        }
        const uint32_t slot = lo, b = unit - pre[lo];
        const uint64_t c = list ? (uint64_t)list[slot] : slot;
        if (EMIT && fail[c]) continue;
        uint32_t W, L, n_samples;
        uint64_t wbase;
        if (G.uniform) { W = G.u_n_waves; L = G.u_wave_len; n_samples = G.u_n_samples; wbase = c * W; }
        else { const ChunkDesc d = G.chunks[c]; W = d.n_waves; L = d.wave_len; n_samples = d.n_samples; wbase = d.wave_base; }
        const uint64_t begin = chunk_word_off[c], end = chunk_word_off[c + 1];
        const uint32_t len_w = (uint32_t)(end - begin);  // (a chunk with an unusable extent has no blocks)
        const uint32_t b0 = b * B;  // block = words [b0, b0 + B) of the chunk
        if (b0 >= len_w) continue;   // (only a chunk cut at the host's bound; flagged by k_bw_scan)
        const uint32_t blk_len = len_w - b0 < B ? len_w - b0 : B;
        const uint32_t max_full = (uint32_t)(((uint64_t)L * 25u + 31u) >> 5);
        const uint32_t min_words = min_payload_words(L, G.k);
        // the block's words into LDS (a fixed grid of 16-byte quads relative to the block)
        {
            const uint64_t a0 = begin + b0;
            const bool vec_ok = (((uintptr_t)(in + a0)) & 15u) == 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const uint32_t i = (uint32_t)(j * 64 + lane) * 4u;
                uint4 v;
                if (vec_ok && i + 4u <= blk_len) {
                    v = *reinterpret_cast<const uint4 *>(in + a0 + i);
                } else {
                    v.x = (i + 0u < blk_len) ? in[a0 + i + 0u] : 0xffffffffu;
                    v.y = (i + 1u < blk_len) ? in[a0 + i + 1u] : 0xffffffffu;
                    v.z = (i + 2u < blk_len) ? in[a0 + i + 2u] : 0xffffffffu;
                    v.w = (i + 3u < blk_len) ? in[a0 + i + 3u] : 0xffffffffu;
                }
                *reinterpret_cast<uint4 *>(blk + i) = v;
            }
        }
        wave_sync();
        if (!EMIT) {
            // candidates in position order: the first word in [1, max_words] at or after `from` (word 0 of the chunk is its
            // sample count: block 0 starts at word 1), 256 words per step.  Never 0: a waveform has at least one payload
            // word, while the zero-padded LAST word of a waveform is all zeros whenever its final code ends in zero bits --
            // an impostor that would chain straight into the real header behind it
            uint32_t from = b == 0 ? 1u : 0u;
            uint32_t entry = 0xffffffffu, count = 0, exit_pos = 0;
            bool found = false;
            for (uint32_t t = 0; t < kBwTries && !found; ++t) {
                uint32_t first = 0xffffffffu;
                for (uint32_t base = from & ~255u; base < blk_len; base += 256u) {
                    const uint32_t i = base + 4u * (uint32_t)lane;
                    const uint4 v = *reinterpret_cast<const uint4 *>(blk + i);  // (all B words were written above)
                    uint32_t f = 0xffffffffu;
                    if (i + 3u >= from && i + 3u < blk_len && v.w - 1u < max_full) f = i + 3u;
                    if (i + 2u >= from && i + 2u < blk_len && v.z - 1u < max_full) f = i + 2u;
                    if (i + 1u >= from && i + 1u < blk_len && v.y - 1u < max_full) f = i + 1u;
                    if (i + 0u >= from && i + 0u < blk_len && v.x - 1u < max_full) f = i + 0u;
                    first = ~wave_max_u32(~f);  // minimum over the wave
                    if (first != 0xffffffffu) break;
                }
                if (first == 0xffffffffu) break;
                // chase from `first` to the block's end
                uint32_t rel = first, cnt = 0;
                bool ok = true;
                while (rel < blk_len) {
                    const uint32_t n = __builtin_amdgcn_readfirstlane(blk[rel]);
                    // (n == 0 is no waveform: at least one bit per sample; it also bounds the headers of a block by B / 2)
                    if (n - 1u >= max_full || (uint64_t)b0 + rel + 1u + n > len_w) { ok = false; break; }
                    if (LIST) {
                        // the header list for k_bw_emit: {position in the block, n}.  Its capacity counts on at least 1 + k
                        // bits per sample (min_words) for every waveform but the chunk's last, shorter one
                        if ((n < min_words && b0 + rel + 1u + n != len_w) || cnt >= hop_cap) { ok = false; break; }
                        if (lane == 0) hops[(uint64_t)unit * hop_cap + cnt] = rel | (n << 12);
                    }
                    rel += n + 1u;
                    ++cnt;
                }
                if (ok) { found = true; entry = first; count = cnt; exit_pos = b0 + rel; }
                else from = first + 1u;
            }
            if (lane == 0) {
                BwBlock o;
                o.entry = found ? b0 + entry : 0xffffffffu;
                o.count = count;
                o.exit = exit_pos;
                o.base = 0;
                info[unit] = o;
            }
            continue;
        }
        // EMIT: chase again from the accepted entry, then write the block's part of the table
        const BwBlock me = info[unit];
        if (me.entry == 0xffffffffu) continue;  // a last block without a header (k_bw_scan)
        uint32_t rel = me.entry - b0, hops = 0;
        while (rel < blk_len) {
            const uint32_t n = __builtin_amdgcn_readfirstlane(blk[rel]);
            if (lane == 0) hop[hops] = (uint16_t)rel;
            rel += n + 1u;
            ++hops;
        }
        wave_sync();
        for (uint32_t i = (uint32_t)lane; i < hops; i += 64u) {
            const uint32_t pos = hop[i], n = blk[pos], wi = me.base + i;  // (k_bw_scan accepted the chunk: wi < W)
            wave_off[wbase + wi] = begin + b0 + pos;
            wave_words[wbase + wi] = n;
            // the chunk's last waveform may be shorter than the rest: its header has tighter bounds; and no
            // header may be below the minimum of 1 + k bits per sample (the chase only checked the upper bound)
            if (wi + 1u == W) {
                const uint32_t last_len = n_samples - (W - 1u) * L;
                if (n > max_payload_words(last_len) || n < min_payload_words(last_len, G.k)) atomicOr(&st->err, kErrCorrupt);
            } else if (n < min_payload_words(L, G.k)) {
                atomicOr(&st->err, kErrCorrupt);
            }
        }
    }
}

// one wavefront per chunk: stitch the blocks, first-waveform index of every block, verdict
template <uint32_t B>
__global__ __launch_bounds__(64) void k_bw_scan(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                const uint64_t *__restrict__ chunk_word_off, const uint32_t *__restrict__ list,
                                                uint32_t n_list, uint32_t blocks_max, BwBlock *__restrict__ info,
                                                uint32_t *__restrict__ fail) {
    __shared__ uint32_t pre[kBwMaxList + 1];
    const int lane = lane_id();
    bw_block_prefix<B>(chunk_word_off, in_words, list, n_list, blocks_max, pre, lane);
    const uint64_t slot = blockIdx.x;
    const uint64_t c = list ? (uint64_t)list[slot] : slot;
    uint32_t W, L, N;
    if (G.uniform) { W = G.u_n_waves; L = G.u_wave_len; N = G.u_n_samples; }
    else { const ChunkDesc d = G.chunks[c]; W = d.n_waves; L = d.wave_len; N = d.n_samples; }
    const uint64_t begin = chunk_word_off[c], end = chunk_word_off[c + 1];
    bool bad = end > in_words || begin + 2 > end || end - begin > 0x7fffffffull;
    if (!bad && in[begin] != N) bad = true;
    const uint32_t len_w = bad ? 0u : (uint32_t)(end - begin);
    const uint32_t n_blocks = (len_w + B - 1u) / B;
    if (n_blocks > blocks_max) bad = true;
    BwBlock *my = info + pre[slot];
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < n_blocks && !bad; b0 += 64) {
        const uint32_t b = b0 + (uint32_t)lane;
        BwBlock o{0xffffffffu, 0, 0, 0};
        if (b < n_blocks) o = my[b];
        // every block must have been entered, start where its predecessor left, and the ends must be the chunk's
        uint32_t prev_exit = (uint32_t)__shfl_up((int)o.exit, 1);
        if (lane == 0) prev_exit = b0 ? my[b0 - 1u].exit : 1u;
        bool lane_bad = false;
        if (b < n_blocks) {
            if (b + 1u == n_blocks && b > 0u && prev_exit == len_w) {
                // the chain already ended inside the previous block: the last block is the tail of the last payload and
                // has no header of its own (whatever small word it may hold is not one)
                o.count = 0;
                my[b].entry = 0xffffffffu;
                my[b].count = 0;
            } else {
                lane_bad = o.entry == 0xffffffffu || o.entry != prev_exit;
                if (b + 1u == n_blocks && o.exit != len_w) lane_bad = true;
            }
        }
        if (__any(lane_bad)) { bad = true; break; }
        const uint32_t inc = wave_incl_scan_dpp(b < n_blocks ? o.count : 0u);
        if (b < n_blocks) my[b].base = run + inc - o.count;
        run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (!bad && run != W) bad = true;
    // the last waveform may be shorter: its header has a tighter bound than the blocks checked
    if (lane == 0) fail[c] = bad ? 1u : 0u;
    (void)L;
}

// Second pass where the first one left the header list (bw_hop_cap() != 0): a wavefront per block, striding over the real
// blocks as above, copies {position, n} into the table at the index k_bw_scan gave the block.  No LDS, no second read of
// the stream, no second chase (config 5: 0.185 -> 0.03 ms).
template <uint32_t B>
__global__ __launch_bounds__(256) void k_bw_emit(Geom G, uint64_t in_words, const uint64_t *__restrict__ chunk_word_off,
                                                 const uint32_t *__restrict__ list, uint32_t n_list, uint32_t blocks_max,
                                                 const BwBlock *__restrict__ info, const uint32_t *__restrict__ fail,
                                                 const uint32_t *__restrict__ hops, uint32_t hop_cap,
                                                 uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words, DevStatus *st) {
    __shared__ uint32_t pre[kBwMaxList + 1];
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    if (wv == 0) bw_block_prefix<B>(chunk_word_off, in_words, list, n_list, blocks_max, pre, lane);
    __syncthreads();
    const uint32_t total = pre[n_list];
    for (uint32_t unit = blockIdx.x * 4u + wv; unit < total; unit += gridDim.x * 4u) {
        uint32_t lo = 0, hi = n_list;  // invariant: pre[lo] <= unit < pre[hi]
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pre[mid] <= unit) lo = mid; else hi = mid;
        }
        const uint32_t slot = lo, b = unit - pre[lo];
        const uint64_t c = list ? (uint64_t)list[slot] : slot;
        if (fail[c]) continue;
        const BwBlock me = info[unit];
        if (me.entry == 0xffffffffu) continue;  // a last block without a header (k_bw_scan)
        uint32_t W, L, n_samples;
        uint64_t wbase;
        if (G.uniform) { W = G.u_n_waves; L = G.u_wave_len; n_samples = G.u_n_samples; wbase = c * W; }
        else { const ChunkDesc d = G.chunks[c]; W = d.n_waves; L = d.wave_len; n_samples = d.n_samples; wbase = d.wave_base; }
        const uint64_t at = chunk_word_off[c] + (uint64_t)b * B;
        for (uint32_t i = (uint32_t)lane; i < me.count; i += 64u) {
            const uint32_t h = hops[(uint64_t)unit * hop_cap + i], pos = h & 0xfffu, n = h >> 12, wi = me.base + i;
            wave_off[wbase + wi] = at + pos;
            wave_words[wbase + wi] = n;
            // the chunk's last waveform may be shorter than the rest: its header has tighter bounds (the others were held
            // to [min, max] by the chase)
            if (wi + 1u == W) {
                const uint32_t last_len = n_samples - (W - 1u) * L;
                if (n > max_payload_words(last_len) || n < min_payload_words(last_len, G.k)) atomicOr(&st->err, kErrCorrupt);
            }
        }
    }
}

// capacity of a block's header list (0: the batch keeps the second chase): B-word blocks hold at most B / (min_words + 1)
// headers + the chunk's last.  Waveforms of fewer than 32 words keep the second chase: one lane's store per header costs
// more than it saves there (100 chunks of 14 M samples, walk with lists / with the second chase: L = 64 2.85 / 2.44 ms,
// 128 1.59 / 1.44, 512 0.64 / 0.73, 1024 0.48 / 0.64, 2048 0.45 / 0.68, 3072 0.55 / 0.97)
__host__ inline uint32_t bw_hop_cap(uint32_t B, uint32_t min_len, uint32_t k) {
    const uint32_t mw = min_payload_words(min_len, k);
    if (mw < 32u || B > 4096u) return 0u;
    return B / (mw + 1u) + 2u;
}

__global__ __launch_bounds__(64) void k_walk_block_only(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                        const uint64_t *__restrict__ chunk_word_off,
                                                        uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                        DevStatus *st, const uint32_t *__restrict__ only) {
    __shared__ __attribute__((aligned(16))) uint32_t blk[kWalkBlockWords];
    __shared__ __attribute__((aligned(8))) uint2 hop[kWalkHopCap];
    const uint64_t c = blockIdx.x;
    if (c >= G.n_chunks || !only[c]) return;
    walk_chunk_block(G, c, in, in_words, chunk_word_off, wave_off, wave_words, nullptr, st, blk, hop);
}

// Straightforward lane-per-waveform decoder: global loads and 2-byte stores.
// Kept as the simple cross-check of the staged kernel below (decode_impl = 0).
__global__ __launch_bounds__(64) void k_decode_simple(Geom G, const uint32_t *__restrict__ in,
                                                      const uint64_t *__restrict__ wave_off,
                                                      const uint32_t *__restrict__ wave_words, DevStatus *st,
                                                      int16_t *__restrict__ out, const uint32_t *__restrict__ only) {
    const uint64_t g = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (g >= G.total_waves) return;
    if (only && !only[g]) return;  // (behind the block decoder with a general filter: the waveforms it flagged)
    const WaveRef r = locate(G, g);
    const uint32_t *s = in + wave_off[g] + 1;
    const uint32_t n = wave_words[g];
    int16_t *y = out + r.sample_off;
    const uint32_t k = G.k;
    uint64_t win = 0;
    uint32_t have = 0, wi = 0;
    int32_t acc = 0;
    int16_t hist[64];  // general prediction filters only (lives in scratch; the delta path never touches it)
    for (uint32_t i = 0; i < r.len; ++i) {
        if (have <= 32u) {
            const uint32_t w = wi < n ? s[wi] : 0u;
            ++wi;
            win |= (uint64_t)w << (32u - have);
            have += 32u;
        }
        uint32_t q = (uint32_t)__clzll((long long)win);
        q = q > 8u ? 8u : q;
        const uint32_t pl = (q == 8u) ? 16u : k;
        const uint64_t t = win << (q + 1u);
        const uint32_t rem = pl ? (uint32_t)(t >> (64u - pl)) : 0u;
        const uint32_t z = (q == 8u) ? rem : ((q << k) + rem);
        const int32_t d = (int32_t)(z >> 1) ^ -(int32_t)(z & 1u);  // un-zig-zag (:172-177)
        if (G.n_taps == 0) {
            acc += d;  // running sum (:80-89)
        } else {
            // general inverse (:92-101): y[i] = (int16)((int16)(d[i] - sum_{j>=1} taps[j] y[i-j]) / taps[0]);
            // the last 64 outputs live in a lane-private circular history (taps <= DRX_MAX_TAPS = 64)
            uint32_t a = (uint32_t)(int32_t)(int16_t)d;
            for (uint32_t j = 1; j < G.n_taps && j <= i; ++j) a -= (uint32_t)((int32_t)hist[(i - j) & 63u] * G.taps[j]);
            acc = (int32_t)(int16_t)(uint16_t)a / G.taps[0];
            hist[i & 63u] = (int16_t)acc;
        }
        y[i] = (int16_t)acc;
        const uint32_t used = q + 1u + pl;
        win <<= used;
        have -= used;
    }
    // a valid waveform's codes end in its last payload word: n_i = ceil(bits / 32) (src/deltaRice.c:237-241)
    const uint64_t bits = 32ull * wi - have;
    if (r.len && ((bits + 31u) >> 5) != n) atomicOr(&st->err, kErrCorrupt);
}

// Decoder for FEW, LONG waveforms (WaveformLength = -1, the reference's default, makes every chunk one
// waveform of millions of samples): one WORKGROUP of 8 wavefronts per waveform, its 512 lanes parse 512
// consecutive 512-bit segments of the stream at once.  A lane does not know where the first code of its segment
// starts; it assumes the segment start, parses to the end of its segment and reports where its last
// code ended.  Rice codes re-synchronise within a few codes, so that end is almost always right even
// when the start was not.  Each lane then restarts from its predecessor's end until no start changes
// (lane 0's start is known, so after at most 512 rounds every lane is exact; typically one restart),
// the sample counts and delta sums of the segments are prefix-summed over the workgroup, and a last parse
// writes the samples.  Three parses of every bit instead of one, 512 at a time.  Delta filter only (the
// prefix sum over segment sums is what makes the segments independent).
constexpr int kLongSeg = 16;  // words per lane and block
constexpr int kLongOv = 2;    // words of the next segment kept below a lane's column (a code has <= 25 bits)
constexpr uint32_t kLongSegBits = kLongSeg * 32u;
// launch_decode() takes this path for uniform batches whose lane-per-waveform decode would leave most of the
// 98 304 lane slots empty: at most 16 384 waveforms of at least 65 536 samples, or at most 4 096 of at least
// 16 384 (measured crossover at 350 M samples: L = 32 768 lanes 2.2 ms / long 2.9 ms, L = 65 536 3.6 / 2.4 ms)
__host__ __device__ inline bool long_waveform_batch(uint64_t total_waves, uint32_t wave_len) {
    return (wave_len >= 65536u && total_waves <= 16384u) || (wave_len >= 16384u && total_waves <= 4096u);
}

constexpr int kLongWaves = 8;                 // wavefronts per workgroup
constexpr int kLongThreads = 64 * kLongWaves;  // segments parsed at once
constexpr uint32_t kLongGuessBits = 160;  // bits in front of a segment's end from which the first guess is parsed

// One workgroup per waveform walks its blocks in order (fail != nullptr: only the waveforms it flags).  This is the
// FALLBACK of the block-parallel decoder (drx_blocks.hip), which gives every block a workgroup of its own and is 2-6 x
// faster where a parse falls into step within a few codes; waveforms where that fails (a slope-1 ramp: all codes the same
// length) are flagged and come here, where a block starts exactly where its predecessor ended.  (The workgroup-per-block
// form of THIS kernel, with a tail parse predicting block starts, was round 1's path for a handful of long waveforms; the
// new decoder replaced it: 25 x 14 M samples 1.55 -> 0.83 ms, profiles/r02_notes.md.)
__global__ __launch_bounds__(kLongThreads) void k_decode_long(Geom G, const uint32_t *__restrict__ in,
                                                              const uint64_t *__restrict__ wave_off,
                                                              const uint32_t *__restrict__ wave_words, DevStatus *st,
                                                              int16_t *__restrict__ out, const uint32_t *__restrict__ fail,
                                                              const uint32_t *__restrict__ suspect, uint32_t verdict_only) {
    constexpr uint32_t NT = kLongThreads;
    // [kLongRows - 2 - word of the segment][thread] (a lane's bank is its lane number whatever row it reads), rows
    // in REVERSE word order plus one unused row on top: with the bit position kept negated, Q = -pos, the row
    // pair (Q >> 5, Q >> 5 + 1) is (word w + 1, word w) inside a word and (word w, word w - 1) on a word
    // boundary, and v_alignbit_b32(hi, lo, Q) is the 32-bit window in both cases (as in k_decode_lanes)
    // (+ 2 rows below: a lane that has left its segment keeps reading at its last position, up to 49 bits past it)
    constexpr uint32_t kLongRows = kLongSeg + kLongOv + 3;
    __shared__ uint32_t col[kLongRows * NT];
    __shared__ uint32_t s_end[NT];
    __shared__ uint32_t s_tot[2][kLongWaves];
    const uint32_t tid = threadIdx.x;
    const int wv = (int)(tid >> 6), lane = (int)(tid & 63u);
    const uint64_t g = blockIdx.x;
    if (g >= G.total_waves) return;
    if (fail && !fail[g]) {
        // not flagged: the block-parallel decoder's output stands, and so does its verdict on the stream
        if (suspect && suspect[g] && tid == 0) atomicOr(&st->err, kErrCorrupt);
        return;
    }
    if (verdict_only) return;  // (general filters: flagged waveforms are decoded again by k_decode_simple, which judges them too)
    const WaveRef r = locate(G, g);
    const uint32_t *src = in + wave_off[g] + 1;
    const uint32_t n = wave_words[g];
    int16_t *y = out + r.sample_off;
    const uint32_t len = r.len, k = G.k;
    uint32_t blk_word = 0;  // first word of the block
    uint32_t carry_in = 0;  // bit of thread 0's segment at which the next code starts
    uint32_t done = 0;      // samples written
    uint32_t acc_base = 0;  // running sum before the block (mod 2^16)

    // parses this thread's segment from bit `start`; a code is taken when it STARTS inside the segment and inside
    // the stream.  emit: add to the running sum `acc` and store sample number idx, idx + 1, ...
    auto parse = [&](bool enable, uint32_t start, uint32_t avail_bits, auto emit_tag, uint32_t idx, uint32_t acc,
                     uint32_t &end, uint32_t &cnt, uint32_t &sum) __attribute__((always_inline)) {
        constexpr bool EMIT = decltype(emit_tag)::value;
        uint32_t c = 0, sacc = EMIT ? acc : 0u;
        uint32_t Q = 0u - start;  // minus the bit position
        const uint32_t lim = avail_bits < kLongSegBits ? avail_bits : kLongSegBits;
        const int32_t nlim = enable ? -(int32_t)lim : 1;  // a code is taken while -Q < lim, i.e. Q > -lim
        // EMIT: two samples per store where they fill an aligned dword (2-byte stores run into the L2's request rate)
        const uint32_t par4 = (uint32_t)(((uintptr_t)y >> 1) & 1u);  // sample 0's half of its aligned dword
        uint32_t held = 0, held_i = 0;  // the sample waiting for its partner
        bool holding = false;
        // LDS byte address of this thread's row of word 0
        const uint32_t row0 = lds_addr(col) + ((kLongRows - 2u) * NT + tid) * 4u;
        while (__any((int32_t)Q > nlim)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // one vote per four codes
                const bool act = (int32_t)Q > nlim;
                typedef const uint32_t __attribute__((address_space(3))) lds_cu32;
                const lds_cu32 *wp = (const lds_cu32 *)(uintptr_t)(row0 + (uint32_t)(((int32_t)Q >> 5) * (int32_t)(NT * 4u)));
                const uint32_t lo = wp[0], hi = wp[NT];
                const uint32_t win = __builtin_amdgcn_alignbit(hi, lo, Q);
                const uint32_t q = ffbh(win);  // all-zero window (padding): the hardware's -1 is as good as any
                const uint32_t kk = (win < (1u << 24)) ? 16u : k;
                const uint32_t nu = ~(q + kk);  // minus the code length
                const uint32_t z = (q << kk) + __builtin_amdgcn_ubfe(win, nu, kk);
                const uint32_t d = (z >> 1) ^ (0u - (z & 1u));
                const uint32_t s2 = sacc + d;
                if (EMIT) {
                    if (act && idx + c < len && !(kAblate && (G.dbg & 16384u))) {  // (16384: ablation, no stores)
                        const uint32_t i = idx + c;
                        if (((i + par4) & 1u) == 0u) {  // low half of an aligned dword: wait for the next sample
                            held = s2 & 0xffffu;
                            held_i = i;
                            holding = true;
                        } else if (holding) {
                            *reinterpret_cast<uint32_t *>(y + i - 1u) = held | (s2 << 16);
                            holding = false;
                        } else {
                            y[i] = (int16_t)(uint16_t)s2;  // the lane's first sample sits in a high half
                        }
                    }
                }
                sacc = act ? s2 : sacc;
                Q = act ? Q + nu : Q;
                c += act ? 1u : 0u;
            }
        }
        if (EMIT && holding) y[held_i] = (int16_t)(uint16_t)held;  // the lane's last sample had no partner
        const uint32_t pos = 0u - Q;
        end = pos;
        cnt = c;
        sum = sacc;
    };

    while (done < len && blk_word < n) {
        // the block's words, transposed into per-thread columns; the first kLongOv words of segment s + 1 repeat below column s
#pragma unroll
        for (int rr = 0; rr < kLongSeg; ++rr) {
            const uint32_t j = tid + NT * (uint32_t)rr;
            const uint32_t wvl = (blk_word + j < n) ? src[blk_word + j] : 0u;
            const uint32_t sgm = j / kLongSeg, i = j % kLongSeg;
            col[(kLongRows - 2u - i) * NT + sgm] = wvl;
            if (i < (uint32_t)kLongOv && sgm >= 1u) col[(kLongRows - 2u - ((uint32_t)kLongSeg + i)) * NT + sgm - 1u] = wvl;
        }
        if (tid < (uint32_t)kLongOv) {
            const uint32_t wi = blk_word + NT * kLongSeg + tid;
            col[(kLongRows - 2u - ((uint32_t)kLongSeg + tid)) * NT + NT - 1u] = (wi < n) ? src[wi] : 0u;
        }
        __syncthreads();
        const uint32_t seg_word = blk_word + tid * kLongSeg;
        const uint32_t avail_bits = seg_word < n ? ((n - seg_word) > (1u << 26) ? 0xffffffffu : (n - seg_word) * 32u) : 0u;

        // first guess: only where the segment's last code ends is wanted, and a parse re-synchronises within a few
        // codes, so the guess starts kLongGuessBits before the segment's end (thread 0 knows its start)
        uint32_t start = tid == 0 ? carry_in : kLongSegBits - kLongGuessBits, end, cnt, sum;
        parse(true, start, avail_bits, std::false_type{}, 0u, 0u, end, cnt, sum);
        for (uint32_t it = 0; it < NT; ++it) {
            s_end[tid] = end;
            __syncthreads();
            // where the predecessor's last code ended, as a bit of MY segment (kLongSegBits = "nothing left for me")
            const uint32_t pe = tid ? s_end[tid - 1u] : 0u;
            const uint32_t ns = tid == 0 ? carry_in : (pe >= kLongSegBits ? pe - kLongSegBits : kLongSegBits);
            const bool changed = ns != start;
            if (!__syncthreads_or(changed ? 1 : 0)) break;  // (also orders the reads of s_end before its next writes)
            start = ns;
            uint32_t e2, c2, s2;
            parse(changed, start, avail_bits, std::false_type{}, 0u, 0u, e2, c2, s2);
            if (changed) { end = e2; cnt = c2; sum = s2; }
        }
        // prefix sums over the workgroup: samples before my segment, sum of deltas before my segment
        const uint32_t incl_c = wave_incl_scan_dpp(cnt), incl_s = wave_incl_scan_dpp(sum);
        if (lane == 63) { s_tot[0][wv] = incl_c; s_tot[1][wv] = incl_s; }
        s_end[tid] = end;
        __syncthreads();
        uint32_t pre_c = 0, pre_s = 0, tot_c = 0, tot_s = 0;
#pragma unroll
        for (int i = 0; i < kLongWaves; ++i) {
            const uint32_t tc = s_tot[0][i], ts = s_tot[1][i];
            if (i < wv) { pre_c += tc; pre_s += ts; }
            tot_c += tc;
            tot_s += ts;
        }
        const uint32_t end_last = s_end[NT - 1u];
        uint32_t e3, c3, s3;
        parse(true, start, avail_bits, std::true_type{}, done + pre_c + incl_c - cnt, acc_base + pre_s + incl_s - sum, e3, c3, s3);
        done = (tot_c > len - done) ? len : done + tot_c;
        acc_base += tot_s;
        carry_in = end_last >= kLongSegBits ? end_last - kLongSegBits : 0u;
        blk_word += NT * kLongSeg;
        __syncthreads();
    }
    if (done < len && tid == 0) atomicOr(&st->err, kErrCorrupt);  // the stream ended before the waveform did
}

// Staged lane-per-waveform decoder (the production kernel; generations 1-4 are in the git
// history, what each measurement changed is in DESIGN.md).
//
// Decomposition.  The Rice parse is serial inside a waveform, so the unit of parallelism is the
// waveform: one lane per waveform, 64 waveforms per wavefront.
//   stream in   each lane owns an LDS ring of RW words of its compressed stream, refilled in
//               pieces of LW words by 16-byte loads (every byte of the stream is requested once;
//               rocprofv3: TCC_EA0_RDREQ x 128 B = the stream size).  A piece is loaded ahead of
//               need and written to the ring just before the round's stores are issued (vmcnt is
//               in order: a load issued after a store cannot be waited for without that store).
//   ring layout word-major and reversed, ring[RW - (w mod RW)][lane], plus a mirror row: a lane's
//               bank is its lane number whatever row it reads (no conflicts although the 64
//               streams drift apart), and the pair (w, w+1) is always (row+1, row).
//   per sample  the bit position is kept negated, Q = -P: row = Q[5 +: log2 RW] (v_bfe),
//               (lo, hi) = ds_read2st64_b32, win = v_alignbit(hi, lo, Q): 3 VALU + 1 LDS
//               instruction form the 32-bit window, no refill state.  Escape and ordinary codes
//               share one extraction (payload width kk = esc ? 16 : k; the 8 << 16 an escape leaves
//               above bit 15 never reaches the int16 running sum).  14.5 VALU instructions per
//               sample; the kernel is bound by VALU issue (4 cycles per wave64 instruction).
//   samples out transposed through LDS; a lane-private start delay phi makes step u of every round
//               land u*2 bytes past a T*2-byte boundary, so each round stores whole aligned
//               128-byte lines (T = 64).  tools/ubench_store.hip: 16-byte stores reach 5.5 TB/s only
//               when every contiguous run is a whole line; 64-byte aligned runs give 4.2 TB/s and
//               runs that straddle lines 2.6-3.3 TB/s whatever their length.
//   PAIR        two samples per ring access: three words = a 64-bit window always hold two codes
//               (2 x 25 bits); the second sample's window is v_alignbit(winA, winB, -len1).  One LDS
//               round trip on the dependent chain per two samples.
//   FUSED       the header-chain walk runs inside the same launch: workgroups take a ticket; the first
//               ceil(n_chunks/8) tickets walk (eight chunks per wave through scalar loads,
//               walk_chunks_scalar(), publishing one granule per waveform), every later ticket decodes
//               64 waveforms of one chunk as soon as their granules appear.  Decode tickets are dealt group-major (waveforms 0-63 of every chunk,
//               then 64-127 of every chunk, ...), the order in which the 2000-hop chains release them,
//               so the ~1.7 ms of dependent-load latency of the walk disappears behind the decode.
//               A ticket holder is by construction running, so waiting on a lower ticket's walker
//               cannot deadlock whatever the dispatch order.  (Uniform batches only.)
//   NW > 1      STAGED FLUSH (round 3).  What limits the wavefronts per CU is LDS: 16.9 KB of ring + 9.2 KB of transposition
//               buffer allow six, and at an unchanged instruction stream 6 -> 9 per CU is worth 21 % (profiles/r03_notes.md,
//               occupancy A/B).  The transposition buffer is needed only while a round's samples change hands, so NW
//               wavefronts form one workgroup and SHARE one: a wavefront keeps the round's 64 samples per lane in 32
//               VGPRs (the interior rounds are fully unrolled: static register indices), takes the workgroup's lock
//               (an LDS compare-and-swap by lane 0), dumps its registers with eight ds_write_b128, reads them back
//               transposed with eight ds_read_b128, releases the lock and stores whole lines as before.  Rows are 128
//               bytes without padding; the 16-byte block j of row r lives at block j ^ (r & 7), which makes both the
//               dump and the transposed read conflict-free.  The rare edge rounds (masked stores, per-sample staging)
//               hold the lock for the whole round and use the buffer as the private one was used.  NW = 4: two
//               workgroups = 8 wavefronts per CU; NW = 9: one workgroup = 9 per CU (160 260 of 163 840 bytes).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int RW, int LW, int T, int GS, bool FUSED, bool PAIR = false, bool GEN = false, int NW = 1>
__global__ __launch_bounds__(64 * NW, NW == 1 ? 1 : (NW == 4 ? 2 : 3)) void k_decode_lanes(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                     const uint64_t *__restrict__ chunk_word_off,
                                                     uint64_t *__restrict__ wave_off,
                                                     uint32_t *__restrict__ wave_words,
                                                     uint64_t *__restrict__ granules, uint32_t *__restrict__ ticket,
                                                     DevStatus *st, int16_t *__restrict__ out) {
    static_assert(RW >= 2 * LW && (RW == 32 || RW == 64 || RW == 128) && (LW == 8 || LW == 16 || LW == 32), "ring");
    constexpr int LOG_RW = RW == 32 ? 5 : (RW == 64 ? 6 : 7);
#ifndef DRX_DEC_OPAD
#define DRX_DEC_OPAD 4
#endif
    constexpr bool STG = NW > 1;
    static_assert(!STG || (PAIR && T == 64 && GS == 16), "the staged flush is built for two samples per access and 64-sample rounds");
    constexpr int OSW = T / 2 + (STG ? 0 : DRX_DEC_OPAD);  // output row stride in words (16- or 8-byte aligned rows)
    static_assert(OSW % 2 == 0, "rows are read in 8- or 16-byte pieces");
    constexpr int PPS = T / 8;      // 16-byte pieces per stream per round
    constexpr int SPI = 64 / PPS;   // streams per write-out iteration
    constexpr int NV = LW / 4;      // 16-byte loads per piece
    constexpr uint32_t WMASK = (1u << 27) - 1u;
    constexpr uint32_t NEED_AT = GS + 2;                                   // must refill below this many words
    static_assert(T % GS == 0 && RW - LW >= GS + 2, "round length / ring slack");
    // row r: word w with RW - (w mod RW) == r; row 0 mirrors row RW and row -1 mirrors row RW - 1 (PAIR reads
    // three consecutive words: rows r + 1, r, r - 1)
    __shared__ uint32_t ring_mem[NW][(RW + 2) * 64];
    const int wv = STG ? (int)(threadIdx.x >> 6) : 0;
    uint32_t (&ring_all)[(RW + 2) * 64] = ring_mem[wv];
    uint32_t *const ring = ring_all + 64;
    __shared__ uint32_t fl_lock;  // STG: who holds obuf (0: nobody)
#ifdef DRX_DEC_NOOBUF  // (ablation builds: no transposition buffer, a lane's sample stores go to four words, no write-out)
    constexpr int OBW = 256;
    constexpr bool kNoObuf = true;
#else
    constexpr int OBW = 64 * OSW;
    constexpr bool kNoObuf = false;
#endif
    __shared__ __attribute__((aligned(16))) uint32_t obuf[OBW];  // doubles as the start-up tables
    uint64_t *tab_off = reinterpret_cast<uint64_t *>(obuf);  // [64] sample offset of step 0 of round 0
    uint32_t *tab_lo = obuf + 128, *tab_hi = obuf + 192;      // [64] each
    static_assert(64 * OSW >= 256, "tables fit in obuf");

    const int lane = lane_id();
    const uint32_t k = G.k;
    if constexpr (STG) {  // (before any wavefront can leave: a walker returns early)
        if (threadIdx.x == 0) fl_lock = 0u;
        __syncthreads();
    }
    // the workgroup's lock on obuf: taken by lane 0, held by the wavefront.  LDS operations of a wavefront complete in
    // order, so the holder's reads are done once its s_waitcnt has passed; the holder never waits for another wavefront.
    auto flush_lock = [&]() __attribute__((always_inline)) {
        if constexpr (STG) {
            if (kAblate && (G.dbg & 8u)) return;  // (ablation: no lock -- the samples of neighbouring wavefronts mix)
            for (;;) {
                uint32_t got = 1u;
                if (lane == 0) {
                    uint32_t expect = 0u;
                    got = __hip_atomic_compare_exchange_strong(&fl_lock, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP) ? 0u : 1u;
                }
                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)got) == 0u) break;
                __builtin_amdgcn_s_sleep(1);
            }
            wave_sync();
        }
    };
    auto flush_unlock = [&]() __attribute__((always_inline)) {
        if constexpr (STG) {
            if (kAblate && (G.dbg & 8u)) return;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of the buffer has returned
            if (lane == 0) __hip_atomic_store(&fl_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // Stores go through pointers with an explicit global address space: once `out` has travelled through
    // nested by-reference lambda captures the compiler no longer infers it and emits flat_store, which
    // also ticks lgkmcnt and serialises against the LDS traffic of the write-out (measured: 2x slower).
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    typedef u32x4v __attribute__((address_space(1))) g_uint4;
    typedef int16_t __attribute__((address_space(1))) g_i16;
    g_i16 *const outg = (g_i16 *)out;
    uint64_t g;
    bool active;
    uint32_t len = 0, n = 0;
    uint64_t S = 1, ooff = 0;
    const uint32_t wave_idx = STG ? blockIdx.x * (uint32_t)NW + (uint32_t)wv : blockIdx.x;  // (outside the ticketed launch)
    if (FUSED) {
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(ticket, 1u);
        tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
        // walker tickets first.  Chunks of short waveforms (tens of thousands of hops) are streamed
        // through LDS by a whole wave each (walk_chunk_block, in this wave's ring/transposition LDS),
        // the others are chased through scalar loads, kWalkChains chunks per wave.
        constexpr bool kBlockWalkFits = !STG && sizeof(ring_all) >= kWalkBlockWords * 4u && sizeof(obuf) >= kWalkHopCap * 8u;
        const bool u_short = G.uniform && G.u_wave_len <= kWalkShortLen;
        const uint32_t n_blockwalk = G.uniform ? (u_short ? (uint32_t)G.n_chunks : 0u) : G.n_short;
        const uint64_t n_chain = G.uniform ? (u_short ? 0ull : G.n_chunks) : (uint64_t)G.n_long;
        const uint32_t n_walk = n_blockwalk + (uint32_t)((n_chain + (uint32_t)kWalkChains - 1u) / (uint32_t)kWalkChains);
        if (tk < n_walk) {  // walker role
            __builtin_amdgcn_s_setprio(3);  // the chain is the critical path of the whole launch (A/B: -2.5 %)
            if (tk < n_blockwalk) {
                if constexpr (kBlockWalkFits) {
                    const uint64_t c = G.uniform ? (uint64_t)tk : (uint64_t)G.walk_short[tk];
                    walk_chunk_block(G, c, in, in_words, chunk_word_off, wave_off, wave_words, granules, st, ring_all,
                                     reinterpret_cast<uint2 *>(obuf));
                }
            } else {
                walk_chunks_scalar(G, (uint64_t)(tk - n_blockwalk) * kWalkChains, G.uniform ? nullptr : G.walk_long, n_chain,
                                   in, in_words, chunk_word_off, wave_off, wave_words, granules, st);
            }
            return;
        }
        // decode tickets, group-major: waveforms 0-63 of every chunk, then 64-127 of every chunk, ...
        const uint64_t t2 = tk - n_walk;
        const uint64_t grp = t2 / G.n_chunks, c = t2 - grp * G.n_chunks;
        const uint32_t idx = (uint32_t)grp * 64u + lane;  // waveform index inside chunk c
        uint64_t gr = 0;
        if (G.uniform) {
            if ((uint32_t)grp * 64u >= G.u_n_waves) return;  // (STG: a ticket beyond the last group)
            active = idx < G.u_n_waves;
            g = c * G.u_n_waves + idx;
            if (active) {
                len = (idx + 1 == G.u_n_waves) ? (G.u_n_samples - idx * G.u_wave_len) : G.u_wave_len;
                ooff = c * (uint64_t)G.u_n_samples + (uint64_t)idx * G.u_wave_len;
            }
        } else {
            // ragged: the grid covers max_groups groups of every chunk; a chunk with fewer has idle tickets
            const ChunkDesc d = G.chunks[c];
            if ((uint32_t)grp * 64u >= d.n_waves) return;
            active = idx < d.n_waves;
            g = d.wave_base + idx;
            if (active) {
                len = (idx + 1 == d.n_waves) ? (d.n_samples - idx * d.wave_len) : d.wave_len;
                ooff = d.sample_off + (uint64_t)idx * d.wave_len;
            }
        }
        uint32_t spins = 0;
        for (;;) {  // wait for this wave's granules; the walker that writes them holds a lower ticket
            if (active && !(gr & kGranValid))
                gr = __hip_atomic_load(granules + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__any(active && !(gr & kGranValid))) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 24)) {  // cannot happen; never hang the GPU
                if (lane == 0) atomicOr(&st->err, kErrInternal);
                gr |= kGranValid;
                break;
            }
        }
        if (active) {
            n = (uint32_t)(gr >> 32) & 0x7fffffffu;
            S = chunk_word_off[c] + (uint32_t)gr + 1u;
        }
    } else if (!G.uniform && G.rag_order) {
        // ragged batch: wavefronts in order of decreasing WaveformLength (longest processing time first).  A lane takes
        // ~60 ns per sample whatever else runs, so a wavefront of 16 384-sample waveforms that starts last adds its
        // whole 1 ms to the launch (config 5: 1.9 -> 1.2 ms)
        if (wave_idx >= G.rag_groups) return;  // (STG: the last workgroup may have wavefronts to spare)
        const uint2 e = G.rag_order[wave_idx];  // {chunk, group of 64 waveforms inside it}
        const ChunkDesc d = G.chunks[e.x];
        const uint32_t idx = e.y * 64u + (uint32_t)lane;
        active = idx < d.n_waves;
        g = d.wave_base + idx;
        if (active) {
            len = (idx + 1 == d.n_waves) ? (d.n_samples - idx * d.wave_len) : d.wave_len;
            ooff = d.sample_off + (uint64_t)idx * d.wave_len;
            S = wave_off[g] + 1u;
            n = wave_words[g];
        }
    } else {
        g = (uint64_t)wave_idx * 64u + lane;
        active = g < G.total_waves;
        if (active) {
            const WaveRef r = locate(G, g);
            len = r.len;
            ooff = r.sample_off;
            S = wave_off[g] + 1u;
            n = wave_words[g];
        }
    }
    // start delay: step u of every round sits u*2 bytes past a T*2-byte boundary
    const uint32_t phi = active ? (uint32_t)((((uintptr_t)out >> 1) + ooff) & (uint64_t)(T - 1)) : 0u;
    const uint32_t steps = wave_max_u32(len + phi);
    const uint32_t lo_max = wave_max_u32(phi);
    const uint32_t hi_min = ~wave_max_u32(~(phi + len));
    uint64_t wo_off[PPS];  // write-out constants: piece p of stream st_i, i = 0..PPS-1
    uint32_t wo_lo[PPS], wo_hi[PPS];
    // STG: lines are addressed as a wave-uniform base (the line of the wavefront's first waveform: an SGPR pair) plus a
    // 32-bit offset per lane -- half the registers, and no 64-bit vector adds in front of the stores (the launcher sends
    // batches whose 64 waveforms could lie 2^31 samples apart to the one-wavefront form)
    const uint64_t line0 = ooff - phi;
    const uint64_t line_base = STG ? (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(line0 >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)line0))
                                   : 0ull;
    uint32_t wo_rel[PPS];
    if constexpr (STG) {
#pragma unroll
        for (int i = 0; i < PPS; ++i) {
            const int st = i * SPI + lane / PPS, p = lane % PPS;
            wo_rel[i] = (uint32_t)__shfl((int)(uint32_t)(line0 - line_base), st) + 8u * (uint32_t)p;
            wo_off[i] = 0;
            wo_lo[i] = wo_hi[i] = 0u;
        }
    } else {
        tab_off[lane] = ooff - phi;
        tab_lo[lane] = phi;
        tab_hi[lane] = phi + len;
        wave_sync();
#pragma unroll
        for (int i = 0; i < PPS; ++i) {
            const int st = i * SPI + lane / PPS, p = lane % PPS;
            wo_off[i] = tab_off[st] + 8u * (uint32_t)p;
            wo_lo[i] = tab_lo[st];
            wo_hi[i] = tab_hi[st];
        }
        wave_sync();
    }

    const uint64_t A = (S & ~(uint64_t)(RW - 1)) - (uint64_t)RW;  // s0 in [RW, 2 RW)
    const uint32_t s0 = (uint32_t)(S - A);
    const uint32_t endw = s0 + n;
    uint32_t flw = s0 & ~(uint32_t)(LW - 1);
    const bool in_vec_ok = ((uintptr_t)in & 15u) == 0;
    uint32_t *myring = ring + lane;
    typedef uint16_t __attribute__((may_alias)) u16a;
#ifdef DRX_DEC_NOOBUF
    u16a *myout = reinterpret_cast<u16a *>(obuf + lane * 4);
#define DRX_OIDX(x) ((x) & 6)
#else
    u16a *myout = reinterpret_cast<u16a *>(obuf + lane * OSW);
#define DRX_OIDX(x) (x)
#endif
    // the round's whole-line stores (A/B: -DDRX_DEC_NT_STORE marks them non-temporal)
    auto store16 = [&](g_i16 *dst, const uint4 &v) __attribute__((always_inline)) {
#ifdef DRX_DEC_NT_STORE
        __builtin_nontemporal_store((u32x4v){v.x, v.y, v.z, v.w}, (g_uint4 *)dst);
#else
        *(g_uint4 *)dst = (u32x4v){v.x, v.y, v.z, v.w};
#endif
    };
    // 16 bytes of a stream's row: one ds_read_b128 where the rows are 16-byte aligned, else two ds_read_b64
    auto orow16 = [&](int st, int p) __attribute__((always_inline)) -> uint4 {
        if constexpr (OSW % 4 == 0) {
            return *reinterpret_cast<const uint4 *>(obuf + st * OSW + 4 * p);
        } else {
            const uint2 a = *reinterpret_cast<const uint2 *>(obuf + st * OSW + 4 * p);
            const uint2 b = *reinterpret_cast<const uint2 *>(obuf + st * OSW + 4 * p + 2);
            return make_uint4(a.x, a.y, b.x, b.y);
        }
    };

    auto load_piece = [&](uint4 (&v)[NV], uint32_t ahead = 0) {
        const uint64_t a = A + flw + ahead;
        if (in_vec_ok && a + (uint32_t)LW <= in_words) {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const uint4 *>(in + a + 4 * j);
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                v[j].x = (a + 4 * j + 0 < in_words) ? in[a + 4 * j + 0] : 0u;
                v[j].y = (a + 4 * j + 1 < in_words) ? in[a + 4 * j + 1] : 0u;
                v[j].z = (a + 4 * j + 2 < in_words) ? in[a + 4 * j + 2] : 0u;
                v[j].w = (a + 4 * j + 3 < in_words) ? in[a + 4 * j + 3] : 0u;
            }
        }
    };
    auto store_piece = [&](const uint4 (&v)[NV]) {
        const uint32_t r0 = (uint32_t)RW - (flw & (uint32_t)(RW - 1));
        uint32_t *dst = myring + r0 * 64u;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            dst[-(4 * j + 0) * 64] = v[j].x; dst[-(4 * j + 1) * 64] = v[j].y;
            dst[-(4 * j + 2) * 64] = v[j].z; dst[-(4 * j + 3) * 64] = v[j].w;
        }
        if (r0 == (uint32_t)RW) {
            myring[0] = v[0].x;
            if (PAIR) myring[-64] = v[0].y;
        }
        flw += (uint32_t)LW;
    };

    uint32_t Q = 0u - 32u * s0;  // minus the bit position (relative to A)
    // Q_need: position (in Q units, Q decreases) at which this lane must have its next piece;
    // 0x7fffffff away from Q means "never" (stream exhausted)
    uint32_t Q_need;
    auto set_limits = [&]() __attribute__((always_inline)) {
        // avail = flw - cw < X  <=>  cw > flw - X  <=>  Q <= ~(32 * (flw - X + 1) - 1) ... kept simple:
        // cw = (~Q) >> 5, so cw >= c  <=>  ~Q >= 32 c  <=>  Q <= ~(32 c)
        const bool more = flw < endw;
        Q_need = more ? ~(32u * (flw - NEED_AT + 1u)) : Q - 0x7fffffffu;
    };
    auto sync_refill = [&]() __attribute__((always_inline)) {  // serve every lane that is (nearly) dry, waiting for the data
        for (;;) {
            const uint32_t avail = (flw - ((~Q) >> 5)) & WMASK;
            const bool more = flw < endw;
            if (!__any(more && avail < NEED_AT)) break;
            if (more && avail <= (uint32_t)(RW - LW) && !(kAblate && (G.dbg & 2u))) {
                uint4 v[NV];
                load_piece(v);
                store_piece(v);
            } else if (more && avail <= (uint32_t)(RW - LW)) {
                flw += (uint32_t)LW;
            }
            wave_sync();
        }
    };

    {   // start-up: the piece that holds word s0 and as many more as fit
        uint4 v[NV];
        for (int i = 0; i < RW / LW; ++i) {
            if (__ballot(flw < endw && flw + (uint32_t)LW <= s0 + (uint32_t)RW) == 0) break;
            if (flw < endw && flw + (uint32_t)LW <= s0 + (uint32_t)RW) {
                if (!(kAblate && (G.dbg & 2u))) { load_piece(v); store_piece(v); } else flw += (uint32_t)LW;
            }
        }
        wave_sync();
    }
    int32_t acc = 0;
    // GEN: inverse of a general prediction filter with taps[0] = +-1 and at most 4 taps (src/deltaRice.c:92-101):
    // y[i] = +-(d[i] - sum_{j=1..3} taps[j] y[i-j]) in int16, i.e. modulo 2^16 (only the low 16 bits of the
    // products count, so 24-bit multiplies of whatever the registers hold above bit 15 are exact).
    // acc is y[i-1]; y2, y3 the two before it; all zero before the waveform (:96 `if ((i - j) >= 0)`).
    int32_t y2 = 0, y3 = 0;
    const uint32_t nt1 = GEN ? G.fast_nt[0] & 0xffffu : 0u, nt2 = GEN ? G.fast_nt[1] & 0xffffu : 0u,
                   nt3 = GEN ? G.fast_nt[2] & 0xffffu : 0u;
    const bool t0neg = GEN && G.fast_t0neg;
    auto advance = [&](int32_t d) __attribute__((always_inline)) {
        if constexpr (GEN) {
            uint32_t a = (uint32_t)d + __umul24(nt1, (uint32_t)acc) + __umul24(nt2, (uint32_t)y2) + __umul24(nt3, (uint32_t)y3);
            if (t0neg) a = 0u - a;
            y3 = y2;
            y2 = acc;
            acc = (int32_t)a;
        } else {
            acc += d;
        }
    };

    // Where the lane's last code ended (as Q): a valid waveform's codes end inside its last payload word, i.e.
    // n_i = ceil(bits / 32) (src/deltaRice.c:237-241) -- checked after the last round, so that flipped payload bits
    // that change a code length are reported (DRX_ERR_CORRUPT) instead of decoding to garbage silently.  The last
    // step of a lane lies in an edge round or is the last step of an interior round: only those capture.
    const uint32_t hi_step = phi + len;  // this lane decodes its last sample in step hi_step - 1
    uint32_t Q_end = 0;
    uint32_t stg[STG ? T / 2 : 1];  // STG: the round's samples of this lane, two per dword (interior rounds)
    auto decode_group = [&](auto first_tag, auto edge_tag, auto stg_tag, int tg, uint32_t tcur) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool EDGE = decltype(edge_tag)::value;  // tcur + u is the step index; capture Q_end
        constexpr int TGC = decltype(stg_tag)::value;     // >= 0: tg at compile time, samples go to stg[] (interior rounds of STG)
        if (PAIR && !FIRST) {
            // two samples per ring access: a 64-bit window (three words) always holds two codes (2 x 25 bits),
            // so the second sample's window is one v_alignbit away from the first one's length -- one LDS
            // round trip on the dependent chain per two samples instead of one per sample
#pragma unroll
            for (int u = 0; u < GS; u += 2) {
                const uint32_t row = __builtin_amdgcn_ubfe(Q, 5u, (uint32_t)LOG_RW);
                const uint32_t *wp = myring + row * 64u;
                const uint32_t lo = wp[0], hi = wp[64], lo2 = wp[-64];
                const uint32_t winA = __builtin_amdgcn_alignbit(hi, lo, Q);
                const uint32_t winB = __builtin_amdgcn_alignbit(lo, lo2, Q);
                const uint32_t q1 = ffbh(winA);
                const uint32_t kk1 = (winA < (1u << 24)) ? 16u : k;
                const uint32_t nu1 = ~(q1 + kk1);  // minus the code length
                const uint32_t win2 = __builtin_amdgcn_alignbit(winA, winB, nu1);
                const uint32_t q2 = ffbh(win2);
                const uint32_t kk2 = (win2 < (1u << 24)) ? 16u : k;
                const uint32_t nu2 = ~(q2 + kk2);
                // v_bfe_u32 and v_alignbit_b32 read 5 bits of their offset / shift: ~t == 31 - t (mod 32) serves both
                if constexpr (EDGE) {
                    Q_end = (tcur + (uint32_t)u + 1u == hi_step) ? Q + nu1 : Q_end;
                    Q_end = (tcur + (uint32_t)u + 2u == hi_step) ? Q + nu1 + nu2 : Q_end;
                }
                asm("v_add3_u32 %0, %1, %2, %3" : "=v"(Q) : "v"(Q), "v"(nu1), "v"(nu2));
                const uint32_t z1 = (q1 << kk1) + __builtin_amdgcn_ubfe(winA, nu1, kk1);
                const uint32_t z2 = (q2 << kk2) + __builtin_amdgcn_ubfe(win2, nu2, kk2);
                advance((int32_t)(z1 >> 1) ^ -(int32_t)(z1 & 1u));
                const uint32_t a1 = (uint32_t)acc;
                advance((int32_t)(z2 >> 1) ^ -(int32_t)(z2 & 1u));
                // low halves of the two running sums in one v_perm_b32
                if constexpr (TGC >= 0)
                    stg[(TGC + u) / 2] = __builtin_amdgcn_perm((uint32_t)acc, a1, 0x05040100u);
                else
                    *reinterpret_cast<uint32_t *>(myout + DRX_OIDX(tg + u)) = __builtin_amdgcn_perm((uint32_t)acc, a1, 0x05040100u);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            const uint32_t row = __builtin_amdgcn_ubfe(Q, 5u, (uint32_t)LOG_RW);
            const uint32_t *wp = myring + row * 64u;
            const uint32_t lo = wp[0], hi = wp[64];
            const uint32_t win = __builtin_amdgcn_alignbit(hi, lo, Q);
            const uint32_t q = ffbh(win);  // win == 0 only past the end of a corrupt stream
            const bool esc = win < (1u << 24);
            const uint32_t kk = esc ? 16u : k;
            const uint32_t used = q + kk + 1u;
            const uint32_t rem = __builtin_amdgcn_ubfe(win, 32u - used, kk);
            const uint32_t z = (q << kk) + rem;  // escape: 8 << 16 stays above bit 15
            const int32_t d = (int32_t)(z >> 1) ^ -(int32_t)(z & 1u);
            if (FIRST) {
                const bool act = (uint32_t)(tg + u) >= phi;
                const int32_t o1 = acc, o2 = y2, o3 = y3;
                advance(d);
                acc = act ? acc : o1;
                y2 = act ? y2 : o2;
                y3 = act ? y3 : o3;
                Q = act ? Q - used : Q;
            } else {
                advance(d);
                Q -= used;
            }
            if constexpr (EDGE) Q_end = (tcur + (uint32_t)u + 1u == hi_step) ? Q : Q_end;
            if constexpr (TGC >= 0) {  // (two steps fill a staging dword)
                if ((u & 1) == 0) stg[(TGC + u) / 2] = (uint32_t)acc & 0xffffu;
                else stg[(TGC + u) / 2] |= (uint32_t)acc << 16;
            } else {
                myout[DRX_OIDX(tg + u)] = (uint16_t)acc;
            }
        }
    };

    // piece i of an EDGE round (first / last rounds of a waveform): only the samples that belong to the stream
    auto emit_masked = [&](g_i16 *dst, uint32_t lo, uint32_t hi, uint32_t t0, const uint4 &v) __attribute__((always_inline)) {
        const int p = lane % PPS;
        const uint32_t tpos = t0 + 8u * (uint32_t)p;  // step index of the piece's first sample
        if (tpos + 8u > lo && tpos < hi) {
            if (tpos >= lo && tpos + 8u <= hi) {
                *(g_uint4 *)dst = (u32x4v){v.x, v.y, v.z, v.w};
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (tpos + (uint32_t)j >= lo && tpos + (uint32_t)j < hi)
                        dst[j] = (int16_t)(w[j >> 1] >> (16 * (j & 1)));
            }
        }
    };
    auto write_out = [&](uint32_t t0) __attribute__((always_inline)) {
        if (kNoObuf || (kAblate && (G.dbg & 1u))) return;
        if (t0 >= lo_max && t0 + T <= hi_min) {  // interior round: whole aligned lines only
#pragma unroll
            for (int i = 0; i < PPS; ++i) {
                const int st = i * SPI + lane / PPS, p = lane % PPS;
                const uint4 v = orow16(st, p);
                store16(outg + wo_off[i] + t0, v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPS; ++i) {
                const int st = i * SPI + lane / PPS, p = lane % PPS;
                const uint32_t tpos = t0 + 8u * (uint32_t)p;
                if (tpos + 8u > wo_lo[i] && tpos < wo_hi[i]) emit_masked(outg + wo_off[i] + t0, wo_lo[i], wo_hi[i], t0, orow16(st, p));
            }
        }
    };
    // STG: the round's samples change hands through the workgroup's buffer: rows of 128 bytes, block j of row r at block
    // j ^ (r & 7) (conflict-free both ways); nothing but these sixteen LDS operations happens under the lock
    auto flush_exchange = [&](uint4 (&ov)[PPS]) __attribute__((always_inline)) {
        if constexpr (STG) {
            const uint32_t rowx = (uint32_t)lane * 32u + (uint32_t)(lane & 7) * 4u;
            const uint32_t rd = (uint32_t)(lane >> 3) * 32u + 4u * (uint32_t)((lane & 7) ^ (lane >> 3));
            flush_lock();
#pragma unroll
            for (int j = 0; j < T / 8; ++j)
                *reinterpret_cast<uint4 *>(obuf + (rowx ^ (4u * (uint32_t)j))) = make_uint4(stg[4 * j], stg[4 * j + 1], stg[4 * j + 2], stg[4 * j + 3]);
            wave_sync();
#pragma unroll
            for (int i = 0; i < PPS; ++i) ov[i] = *reinterpret_cast<const uint4 *>(obuf + (uint32_t)i * 256u + rd);
            flush_unlock();
        }
    };
    // the samples of the group that ends here are final (see the interior loop)
    auto pin_group = [&](auto gi) __attribute__((always_inline)) {
        if constexpr (STG) {
            constexpr int TG = decltype(gi)::value * GS;
            int32_t &a_ = acc;  // (named here so that the generic lambda captures them: asm operands alone do not)
            uint32_t (&s_)[STG ? T / 2 : 1] = stg;
            asm volatile("" : "+v"(a_), "+v"(s_[TG / 2]), "+v"(s_[TG / 2 + 1]), "+v"(s_[TG / 2 + 2]), "+v"(s_[TG / 2 + 3]),
                         "+v"(s_[TG / 2 + 4]), "+v"(s_[TG / 2 + 5]), "+v"(s_[TG / 2 + 6]), "+v"(s_[TG / 2 + 7]));
        }
    };
    auto stg_line = [&](int i, uint32_t t0) __attribute__((always_inline)) -> g_i16 * {  // piece i of this lane in round t0
        return (outg + line_base + t0) + wo_rel[i];
    };
    auto stg_edge_out = [&](uint32_t t0) __attribute__((always_inline)) {  // STG: an edge round's masked write-out
        uint4 ov[PPS];
        flush_exchange(ov);
        if (kAblate && (G.dbg & 1u)) return;
        if (t0 >= lo_max && t0 + T <= hi_min) {
#pragma unroll
            for (int i = 0; i < PPS; ++i) store16(stg_line(i, t0), ov[i]);
        } else {
#pragma unroll
            for (int i = 0; i < PPS; ++i) {  // (the stream's limits come by shuffle: edge rounds are 2-3 of ~110)
                const int st = i * SPI + lane / PPS;
                emit_masked(stg_line(i, t0), (uint32_t)__shfl((int)phi, st), (uint32_t)__shfl((int)(phi + len), st), t0, ov[i]);
            }
        }
    };

    // round 0 (start delays; synchronous refills; runs once)
    if (steps > 0) {
        if constexpr (STG) {
            static_for<0, T / GS>([&](auto gi) __attribute__((always_inline)) {
                constexpr int TG = decltype(gi)::value * GS;
                sync_refill();
                decode_group(std::true_type{}, std::true_type{}, std::integral_constant<int, TG>{}, TG, (uint32_t)TG);
                pin_group(gi);
            });
            wave_sync();
            stg_edge_out(0);
        } else {
#pragma unroll 1
            for (int tg = 0; tg < T; tg += GS) {
                sync_refill();
                decode_group(std::true_type{}, std::true_type{}, std::integral_constant<int, -1>{}, tg, (uint32_t)tg);
            }
            wave_sync();
            write_out(0);
        }
        wave_sync();
    }

    // steady state.  Stream pieces are requested at the END of a round, just BEFORE the round's stores
    // are issued, and written to the ring at the end of the next round.  vmcnt retires in issue order
    // and counts loads and stores together, so a wait for loads that are OLDER than the PPS stores of
    // their own round is `s_waitcnt vmcnt(PPS)` and never waits for those stores (ablation: loads alone
    // +0.04 ms, stores alone +0.11 ms, both +1.0 ms when the commit had to drain the stores too).
    // The interior rounds (every stream fully inside its waveform: whole-line stores only) run in a
    // loop of their own whose only vector-memory operations are those loads and those PPS stores, so
    // that the compiler's waitcnt insertion can prove the count.  Up to two pieces per lane and round
    // (2 LW words = 16 bits per sample at LW = 16); hungrier streams fall back to sync_refill().
    auto edge_round = [&](uint32_t t0) __attribute__((always_inline)) {  // first / last rounds: masked stores, synchronous refills
        if constexpr (STG) {
            static_for<0, T / GS>([&](auto gi) __attribute__((always_inline)) {
                constexpr int TG = decltype(gi)::value * GS;
                sync_refill();
                decode_group(std::false_type{}, std::true_type{}, std::integral_constant<int, TG>{}, TG, t0 + (uint32_t)TG);
                pin_group(gi);
            });
            wave_sync();
            stg_edge_out(t0);
        } else {
#pragma unroll 1
            for (int tg = 0; tg < T; tg += GS) {
                sync_refill();
                decode_group(std::false_type{}, std::true_type{}, std::integral_constant<int, -1>{}, tg, t0 + (uint32_t)tg);
            }
            wave_sync();
            write_out(t0);
        }
        wave_sync();
    };
    uint32_t t0 = T;
    for (; t0 < steps && !(t0 >= lo_max && t0 + T <= hi_min); t0 += T) edge_round(t0);

    const uint32_t min_words = (kAblate && (G.dbg & 4u)) ? 0u : ((uint32_t)T * (k + 1u)) >> 5;
    uint4 pv0[NV], pv1[NV];               // pieces in flight
    bool pneed0 = false, pneed1 = false;  // this lane has them in flight
    set_limits();
    auto group_refill = [&]() __attribute__((always_inline)) {
        if (__any((int32_t)(Q - Q_need) <= 0)) {  // one signed compare per test (positions are mod 2^32)
            // a piece in flight was requested counting on the words this round consumes at least
            // (min_words): before the round is over it may only be committed where it already fits
            uint32_t avail = (flw - ((~Q) >> 5)) & WMASK;
            if (pneed0 && avail <= (uint32_t)(RW - LW)) {
                store_piece(pv0);
                pneed0 = false;
                avail += (uint32_t)LW;
                if (pneed1 && avail <= (uint32_t)(RW - LW)) { store_piece(pv1); pneed1 = false; }
            }
            if (!pneed0 && pneed1) {
#pragma unroll
                for (int j = 0; j < NV; ++j) pv0[j] = pv1[j];
                pneed0 = true;
                pneed1 = false;
            }
            wave_sync();
            sync_refill();
            set_limits();
        }
    };
    for (; t0 + T <= hi_min && t0 < steps; t0 += T) {  // interior rounds
        if constexpr (STG) {  // fully unrolled: the staging registers are indexed at compile time
            static_for<0, T / GS>([&](auto gi) __attribute__((always_inline)) {
                constexpr int TG = decltype(gi)::value * GS;
                group_refill();
                decode_group(std::false_type{}, std::false_type{}, std::integral_constant<int, TG>{}, TG, 0u);
                // the group's values are final HERE: without this the compiler runs the bit-position chain of the whole
                // unrolled round ahead (the next group's refill test needs only that) and parks every window, quotient and
                // width of sixteen samples in registers until it gets round to the values: 415 VGPRs and 350 spills
                pin_group(gi);
            });
        } else {
#pragma unroll 1
            for (int tg = 0; tg < T; tg += GS) {
                group_refill();
                decode_group(std::false_type{}, std::false_type{}, std::integral_constant<int, -1>{}, tg, 0u);
            }
        }
        Q_end = (t0 + (uint32_t)T == hi_step) ? Q : Q_end;  // a lane whose last step closes an interior round
        wave_sync();
        if (pneed0) store_piece(pv0);  // loads of the previous round end: older than that round's PPS stores
        if (pneed1) store_piece(pv1);
        wave_sync();
        set_limits();
        {
            // a piece is committed one interior round after its request: by then every lane has decoded T more
            // samples of at least k + 1 bits each, i.e. consumed min_words more words
            const uint32_t mc = (t0 + 2u * T <= hi_min && t0 + T < steps) ? min_words : 0u;
            const uint32_t avail = (flw - ((~Q) >> 5)) & WMASK;
            pneed0 = (flw < endw) && avail + (uint32_t)LW <= (uint32_t)RW + mc && !(kAblate && (G.dbg & 2u));
            pneed1 = pneed0 && (flw + (uint32_t)LW < endw) && avail + 2u * (uint32_t)LW <= (uint32_t)RW + mc;
            if (pneed0) load_piece(pv0);
            if (pneed1) load_piece(pv1, (uint32_t)LW);
        }
        if constexpr (STG) {
            uint4 ov[PPS];
            flush_exchange(ov);
            if (!(kAblate && (G.dbg & 1u))) {
#pragma unroll
                for (int i = 0; i < PPS; ++i) store16(stg_line(i, t0), ov[i]);
            }
        } else if (!(kNoObuf || (kAblate && (G.dbg & 1u)))) {
#pragma unroll
            for (int i = 0; i < PPS; ++i) {  // whole aligned lines only
                const int st = i * SPI + lane / PPS, p = lane % PPS;
                const uint4 v = orow16(st, p);
                store16(outg + wo_off[i] + t0, v);
            }
        }
        wave_sync();
    }
    if (pneed0) store_piece(pv0);
    if (pneed1) store_piece(pv1);
    wave_sync();
    for (; t0 < steps; t0 += T) edge_round(t0);
    // bits of the waveform = -Q_end - 32 s0 (Q counts from A); n_i words hold them exactly
    if (active && len && (((0u - Q_end) - 32u * s0 + 31u) >> 5) != n) atomicOr(&st->err, kErrCorrupt);
}

// ---------------------------------------------------------------------------
// launchers (host side, same translation unit so that <<<>>> stays in HIP code)
// ---------------------------------------------------------------------------
static inline unsigned blocks_for(uint64_t items, unsigned per_block) {
    return (unsigned)((items + per_block - 1) / per_block);
}

static inline void mark(hipEvent_t *ev, int i, hipStream_t s) {
    if (ev) (void)hipEventRecord(ev[i], s);
}

hipError_t launch_sideband_tables(const Geom &G, const uint32_t *d_in, uint64_t in_words, const uint64_t *d_chunk_word_off,
                                  const uint32_t *d_n, uint64_t *d_wave_off, uint32_t *d_wave_words, DevStatus *d_status, hipStream_t s) {
    if (G.total_waves == 0) return hipSuccess;
    k_sideband_tables<<<(unsigned)G.n_chunks, 256, 0, s>>>(G, d_in, in_words, d_chunk_word_off, d_n, d_wave_off, d_wave_words, d_status);
    return hipGetLastError();
}

hipError_t launch_estimate_words(const Geom &G, const int16_t *d_in, unsigned long long *d_words16, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_words16, 0, 16 * sizeof(unsigned long long), s);
    if (e != hipSuccess || G.total_waves == 0) return e;
    k_estimate_words<<<blocks_for(G.total_waves, 4), 256, 0, s>>>(G, d_in, d_words16);
    return hipGetLastError();
}

// Single-pass encode (k_encode_fused).  d_scan: uint64[total_waves] + one uint32 ticket
// word after it, zeroed here on the stream before every launch.
hipError_t launch_encode_fused(const Geom &G, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                               uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint64_t *d_scan,
                               DevStatus *d_status, hipEvent_t *ev, hipStream_t s) {
    if (G.total_waves == 0) return hipSuccess;
    mark(ev, 0, s);
    hipError_t e = hipMemsetAsync(d_scan, 0, (G.total_waves + 2) * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    mark(ev, 1, s);
    mark(ev, 2, s);
    uint32_t *ticket = reinterpret_cast<uint32_t *>(d_scan + G.total_waves);
    if (G.n_taps)
        k_encode_fused<true><<<blocks_for(G.total_waves, kEncWaves), 64 * kEncWaves, 0, s>>>(G, d_in, d_out, out_cap, d_chunk_word_off,
                                                                                      d_wave_words, d_scan, ticket, d_status);
    else
        k_encode_fused<false><<<blocks_for(G.total_waves, kEncWaves), 64 * kEncWaves, 0, s>>>(G, d_in, d_out, out_cap, d_chunk_word_off,
                                                                                       d_wave_words, d_scan, ticket, d_status);
    mark(ev, 3, s);
    return hipGetLastError();
}

// ev: optional 4 events recorded before / between / after the kernels (profiling).
hipError_t launch_encode(const Geom &G, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                         uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint32_t *d_wave_rel,
                         uint64_t *d_chunk_words, DevStatus *d_status, hipEvent_t *ev, hipStream_t s) {
    if (G.total_waves == 0) return hipSuccess;
    mark(ev, 0, s);
    k_encode_sizes<<<blocks_for(G.total_waves, 4), 256, 0, s>>>(G, d_in, d_wave_words);
    mark(ev, 1, s);
    k_chunk_scan<<<(unsigned)G.n_chunks, 256, 0, s>>>(G, d_wave_words, d_wave_rel, d_chunk_words);
    k_chunk_offsets<<<1, 1024, 0, s>>>(G.n_chunks, d_chunk_words, d_chunk_word_off, out_cap, d_status);
    mark(ev, 2, s);
    k_encode_pack<<<blocks_for(G.total_waves, 4), 256, 0, s>>>(G, d_in, d_wave_words, d_wave_rel,
                                                               d_chunk_word_off, d_out, out_cap);
    mark(ev, 3, s);
    return hipGetLastError();
}

// Batches the segment encoder takes: few long waveforms, and SHORT waveforms (one segment each), where the
// single-pass encoder pays a workgroup barrier, a look-back and an 8 KB LDS clear per 512-2048 samples
// (200 chunks of 14 M samples: L = 512 0.57 -> 0.95 TB/s, 1024 0.97 -> 1.33, 2048 1.51 -> 1.68), and waveforms
// long enough to outgrow the single pass's LDS buffer; in between the single pass is better
// block-parallel walk of short waveforms: blocks per chunk at 25 bits per sample (0: the batch does not take it)
uint32_t bw_walk_blocks_max(const Geom &G) {
    // against the walk inside the decode launch (serial chase through LDS, 0.13 us per waveform of a chunk, all chunks at
    // once, so that large batches hide most of it).  Measured (chunks of 14 M samples, decode GB/s of the two paths at
    // 150 / 220 chunks): L = 512 1602 / 1627 against 875 / 1265; L = 1024 1869 / 1927 against 1521 / 1954; L = 2048
    // 1906 / 2012 against 1673 / 2169: about one chunk per 35 waveforms of a chunk.  Above WaveformLength 2048 the
    // alternative is the scalar chain at 0.85 us per hop (L = 3072, 100 / 220 chunks: 1731 / 1459+ against 596 / 1123): W / 18
    const uint64_t per = G.u_wave_len <= kWalkShortLen ? 35u : 18u, cap = kPwMaxChunks;
    const uint64_t limit = G.u_n_waves / per < cap ? G.u_n_waves / per : cap;
    // every 4096-word block must hold a header: n_i <= 25 L / 32 < 4096, i.e. L <= 5000; chunks of longer
    // waveforms within the chunk-wide walk's capacity take that one
    const bool chunk_wide = G.u_wave_len > kWalkShortLen && G.u_n_waves <= kPwMaxWaves && G.u_n_waves >= 64u;
    if (!(G.uniform && G.n_chunks <= limit && G.u_wave_len <= 5000u && G.u_wave_len >= 16u && !chunk_wide)) return 0;
    const uint64_t max_words = 1u + G.u_n_waves + (((uint64_t)G.u_n_samples * 25u + 31u) >> 5) + G.u_n_waves;
    const uint64_t nb = (max_words + kWalkBlockWords - 1u) / kWalkBlockWords;
    return nb > 0xfffffu ? 0u : (uint32_t)nb;
}

// bytes of header list per 4096 words of stream for the block size the launch will choose (0: none kept)
static uint64_t bw_hop_bytes_per_block4096(const Geom &G) {
    const uint32_t max_len = G.uniform ? G.u_wave_len : kWalkShortLen, min_len = G.uniform ? G.u_wave_len : G.rag_bw_min_len;
    const uint32_t max_full = (uint32_t)(((uint64_t)max_len * 25u + 31u) >> 5);
    const uint32_t B = max_full + 2u <= 1024u ? 1024u : (max_full + 2u <= 2048u ? 2048u : 4096u);
    return (uint64_t)bw_hop_cap(B, min_len, G.k) * sizeof(uint32_t) * (kWalkBlockWords / B);
}

// scratch of the parallel header walks (0: the batch takes neither); layout in launch_decode()
uint64_t par_walk_scratch_bytes(const Geom &G) {
    bool pw, bw;
    uint64_t bw_units;
    if (G.uniform) {
        const uint32_t nb = bw_walk_blocks_max(G);
        bw = nb != 0;
        bw_units = G.n_chunks * nb;
        pw = !bw && G.n_chunks <= kPwMaxChunks && G.u_n_waves <= kPwMaxWaves && G.u_n_waves >= 64u && G.u_wave_len > kWalkShortLen;
    } else {
        if (!G.rag_par) return 0;
        pw = G.n_long != 0;
        bw = G.n_short != 0;
        bw_units = (uint64_t)G.n_short * G.rag_bw_blocks_max;
    }
    if (!pw && !bw) return 0;
    return (pw ? G.n_chunks * kPwStride * sizeof(uint2) : 0) + (3u * G.n_chunks + 2u) * sizeof(uint32_t) +
           bw_units * (kWalkBlockWords / 1024u) * sizeof(BwBlock) +  // (blocks of 1024 words at the smallest)
           bw_units * bw_hop_bytes_per_block4096(G);                 // header lists of the first block pass
}

bool long_batch(const Geom &G) {
    if (G.n_taps) return false;
    if (!G.uniform) return G.seg_unit_base != nullptr;  // decided when the plan was made (some chunk is short or long)
    // measured at 100 chunks of 14 M samples (single pass / segments, TB/s): L = 2049 0.95 / 1.06, 3000 1.57 / 1.64,
    // 4096 1.93 / 1.85, 7000 2.16 / 1.87, 12000 1.33 / 1.63 (the single pass outgrows its 8 KB LDS buffer at
    // ~6.5 bits per sample and encodes such waveforms twice)
    return long_waveform_batch(G.total_waves, G.u_wave_len) || G.u_wave_len <= kSegShortLenHost || G.u_wave_len >= kSegLongLenHost ||
           (G.dbg & 8192u);
}
static uint32_t uniform_segments(const Geom &G) { return (G.u_wave_len + kSegSamples - 1u) / kSegSamples; }
uint64_t long_batch_units(const Geom &G) { return G.uniform ? G.total_waves * uniform_segments(G) : G.seg_units; }

// Encoder for few long waveforms.  d_seg_bits: uint32[total_waves * segments], d_seg_pos: uint64[same].
hipError_t launch_encode_long(const Geom &G, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                              uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint32_t *d_wave_rel,
                              uint64_t *d_chunk_words, uint32_t *d_seg_bits, uint64_t *d_seg_pos, DevStatus *d_status,
                              hipEvent_t *ev, hipStream_t s) {
    if (G.total_waves == 0) return hipSuccess;
    const uint32_t S = G.uniform ? uniform_segments(G) : 0u;
    const uint64_t units = long_batch_units(G);
    mark(ev, 0, s);
    const uint32_t upw = (G.uniform && S == 1u && G.u_wave_len <= 1024u) ? 8u : 1u;  // short waveforms: eight per wavefront
    k_seg_sizes<<<blocks_for(units, 4 * upw), 256, 0, s>>>(G, d_in, S, units, upw, d_seg_bits);
    mark(ev, 1, s);
    k_seg_scan<<<(unsigned)G.total_waves, 64, 0, s>>>(G, S, d_seg_bits, d_seg_pos, d_wave_words);
    k_chunk_scan<<<(unsigned)G.n_chunks, 256, 0, s>>>(G, d_wave_words, d_wave_rel, d_chunk_words);
    k_chunk_offsets<<<1, 1024, 0, s>>>(G.n_chunks, d_chunk_words, d_chunk_word_off, out_cap, d_status);
    k_seg_zero<<<blocks_for(units, 256), 256, 0, s>>>(G, S, units, d_seg_pos, d_wave_rel, d_chunk_word_off, d_out, out_cap);
    mark(ev, 2, s);
    k_seg_pack<<<blocks_for(units, 4 * upw), 256, 0, s>>>(G, d_in, S, units, d_seg_pos, d_wave_words, d_wave_rel, d_chunk_word_off,
                                                          d_out, out_cap, upw);
    mark(ev, 3, s);
    return hipGetLastError();
}

// Ablation builds only: DRX_DEC_LDS_PAD = bytes of dynamic LDS added to every k_decode_lanes launch (occupancy A/B at an
// unchanged instruction stream: 26 KB + pad per wavefront decides how many of them a CU holds).
static unsigned dec_lds_pad() {
#ifdef DRX_ABLATION
    static const unsigned pad = [] { const char *e = getenv("DRX_DEC_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
    return pad;
#else
    return 0u;
#endif
}

#ifndef DRX_DEC_T
#define DRX_DEC_T 64
#endif
#ifndef DRX_DEC_LW
#define DRX_DEC_LW 16
#endif
hipError_t launch_decode(const Geom &G, const uint32_t *d_in, uint64_t in_words,
                         const uint64_t *d_chunk_word_off, int16_t *d_out, uint64_t *d_wave_off,
                         uint32_t *d_wave_words, uint64_t *d_granules, DevStatus *d_status, int impl,
                         void *d_pw, void *d_blk, const SideStream *side, hipEvent_t *ev, hipStream_t s, uint32_t *path_out) {
    uint32_t path_dummy = 0;
    uint32_t &path = path_out ? *path_out : path_dummy;
    path = 0;
    if (G.total_waves == 0) return hipSuccess;
    const unsigned lpad = dec_lds_pad();
    mark(ev, 0, s);
    // impl >= 100: wave_off / wave_words are already filled in (the one-chunk host path walks the header
    // chain on the CPU while the chunk is in flight to the device): decode with variant impl - 100, no walk
    const bool tables_ready = impl >= 100;
    if (tables_ready) impl -= 100;
    // general prediction filters: the staged kernel where it has no division to do, else the simple kernel
    const bool gen = G.n_taps != 0;
    if (gen && !G.fast_taps) impl = 0;
    if (gen && G.fast_taps) impl = (impl == 0) ? 0 : ((impl == 1 || impl == 7) ? 7 : 8);
    if (tables_ready && impl == 5) impl = 1;
    if (tables_ready && impl == 8) impl = 7;
    // ragged: the group-major grid has max_groups tickets per chunk; not when most of them would be idle
    // few long waveforms (delta filter): a wavefront per waveform instead of a lane per waveform
    // (general filters the fast kernels take: the block decoder leaves residuals, the inverse filter runs in place behind it)
    const bool blocks_path = impl != 0 && !(G.dbg & (256u | 512u)) && d_blk && blocks_batch(G) && (!gen || (G.iir_tab && G.iir_state));
    const bool long_path = !blocks_path && !gen && impl != 0 && !(G.dbg & 256u) && G.uniform && long_waveform_batch(G.total_waves, G.u_wave_len);
    // a handful of chunks of long-enough waveforms: the parallel walk, then a plain decode launch
    const bool par_walk = d_pw && !tables_ready && !(G.dbg & 2048u) && G.uniform && G.n_chunks <= kPwMaxChunks &&
                          G.u_n_waves <= kPwMaxWaves && G.u_n_waves >= 64u && G.u_wave_len > kWalkShortLen;
    // ... and of short waveforms: the block-parallel walk
    const uint32_t bw_blocks_max = bw_walk_blocks_max(G);
    const bool bw_walk = d_pw && bw_blocks_max && !tables_ready && !(G.dbg & 2048u);
    const bool rag_par = d_pw && !G.uniform && G.rag_par && !tables_ready && !(G.dbg & 2048u);
    const bool sparse = !G.uniform && (uint64_t)G.n_chunks * G.max_groups > 8ull * ((G.total_waves + 63u) / 64u) + 4096ull;
    const bool fused = (impl == 5 || impl == 8) && !sparse && !long_path && !blocks_path && !par_walk && !bw_walk && !rag_par;
    if (fused) {
        // granules + ticket word, zeroed before every launch (a granule is its own ready flag)
        hipError_t e = hipMemsetAsync(d_granules, 0, (G.total_waves + 2) * sizeof(uint64_t), s);
        if (e != hipSuccess) return e;
        mark(ev, 1, s);
        uint32_t *ticket = reinterpret_cast<uint32_t *>(d_granules + G.total_waves);
        unsigned n_walk, groups;
        if (G.uniform) {
            n_walk = (G.u_wave_len <= kWalkShortLen) ? (unsigned)G.n_chunks : blocks_for(G.n_chunks, kWalkChains);
            groups = (G.u_n_waves + 63u) / 64u;
        } else {
            n_walk = G.n_short + blocks_for(G.n_long, kWalkChains);
            groups = G.max_groups;
        }
        const unsigned nb = n_walk + (unsigned)(G.n_chunks * groups);
        path |= 1u;  // DRX_PATH_LANES_FUSED
#ifndef DRX_DEC_NW
#define DRX_DEC_NW 1
#endif
        // staged flush (NW wavefronts share one transposition buffer: 8 or 9 wavefronts per CU instead of 6); its walker
        // role has no LDS to stream short-waveform chunks through, those batches keep the one-wavefront form
        constexpr int NWs = DRX_DEC_NW;
        const bool short_walk = G.uniform ? G.u_wave_len <= kWalkShortLen : G.n_short != 0;
        const bool lines_near = G.uniform ? (uint64_t)G.u_wave_len * 64u < (1ull << 31) : G.max_wave_len64 < (1ull << 31);
        if (NWs > 1 && impl == 8 && !short_walk && lines_near && !(G.dbg & 1048576u)) {
            const unsigned nwg = (nb + NWs - 1u) / NWs;
            if (gen)
                k_decode_lanes<64, DRX_DEC_LW, 64, 16, true, true, true, NWs><<<nwg, 64 * NWs, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
            else
                k_decode_lanes<64, DRX_DEC_LW, 64, 16, true, true, false, NWs><<<nwg, 64 * NWs, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
        } else if (impl == 8 && gen)
            k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, true, true, true><<<nb, 64, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
        else if (impl == 8)
            k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, true, true><<<nb, 64, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
        else
            k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, true><<<nb, 64, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
    } else {
        // the lane-per-waveform launch outside the fused form (tables in wave_off / wave_words): `nb` wavefronts of view Gv
        auto launch_lanes = [&](const Geom &Gv, unsigned nb, int im, hipStream_t st_) {
            path |= 2u;  // DRX_PATH_LANES
            if (im == 7 && gen)
                k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, false, true, true><<<nb, 64, lpad, st_>>>(Gv, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, nullptr, nullptr, d_status, d_out);
            else if (im == 7)
                k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, false, true><<<nb, 64, lpad, st_>>>(Gv, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, nullptr, nullptr, d_status, d_out);
            else
                k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, false><<<nb, 64, lpad, st_>>>(Gv, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, nullptr, nullptr, d_status, d_out);
        };
        bool lanes_done = false;
        // chunks of short waveforms: stream the chunk through LDS; long waveforms: one dependent load per hop
        if (tables_ready) {
        } else if (par_walk || bw_walk || rag_par) {
            // scratch: uint2 cand[n_chunks * kPwCap] | uint32 count[n_chunks] | pw_fail[n_chunks] | bw_fail[n_chunks] |
            //          BwBlock info[n_bw * bw_blocks]   (cand .. pw_fail only where the chunk-wide walk is used)
            const bool use_pw = par_walk || (rag_par && G.n_long), use_bw = bw_walk || (rag_par && G.n_short);
            const uint32_t *pw_list = G.uniform ? nullptr : G.walk_long, *bw_list = G.uniform ? nullptr : G.walk_short;
            const uint32_t n_pw = G.uniform ? (uint32_t)G.n_chunks : G.n_long, n_bw = G.uniform ? (uint32_t)G.n_chunks : G.n_short;
            const uint32_t bwb = G.uniform ? bw_blocks_max : G.rag_bw_blocks_max;
            uint2 *cand = reinterpret_cast<uint2 *>(d_pw);
            uint32_t *cnt = reinterpret_cast<uint32_t *>(cand + (use_pw ? G.n_chunks * kPwStride : 0));
            uint32_t *pw_fail = cnt + G.n_chunks, *bw_fail = pw_fail + G.n_chunks;
            BwBlock *info = reinterpret_cast<BwBlock *>(bw_fail + G.n_chunks + (G.n_chunks & 1u));
            hipError_t e = hipMemsetAsync(cnt, 0, 3u * G.n_chunks * sizeof(uint32_t), s);
            if (e != hipSuccess) return e;
            // a ragged batch has both kinds of chunk and the two walks touch different chunks: the chunk-wide walk goes to the
            // context's side stream while the block walk runs here (config 5: 0.18 ms of 0.6 off the critical path)
            const bool forked = use_pw && use_bw && side && side->s;
            hipStream_t spw = forked ? side->s : s;
            if (forked) {
                if ((e = hipEventRecord(side->fork, s)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(side->s, side->fork, 0)) != hipSuccess) return e;
            }
            if (use_pw) {
                k_pw_scan<<<(unsigned)(n_pw * pw_parts(n_pw)), 256, 0, spw>>>(G, d_in, in_words, d_chunk_word_off, pw_list, cand, cnt, pw_parts(n_pw));
                k_walk_parallel<<<n_pw, kPwThreads, 0, spw>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words,
                                                              pw_fail, pw_list, cand, cnt, pw_parts(n_pw));
                k_walk_scalar_only<<<blocks_for(G.n_chunks, kWalkChains), 64, 0, spw>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off,
                                                                                        d_wave_words, d_status, pw_fail);
            }
            // ... and so does the decoding of the long-waveform chunks, whose tables are complete long before the block walk is
            // through: their wavefronts (the first rag_groups_long of the longest-first order) are launched behind the
            // chunk-wide walk on the side stream, the rest here behind the block walk
            const int im_split = (impl == 0) ? 0 : ((impl == 5 || impl == 1) ? 1 : 7);
            const bool split = forked && G.rag_order && im_split != 0 && G.rag_groups_long && G.rag_groups_long < G.rag_groups &&
                               !(G.dbg & 131072u) && !blocks_path;
            if (split) {
                Geom Gl = G;
                Gl.rag_groups = G.rag_groups_long;
                launch_lanes(Gl, Gl.rag_groups, im_split, spw);
            }
            if (forked && (e = hipEventRecord(side->join, side->s)) != hipSuccess) {
                // the side stream's kernels write this call's tables and output: never return with them unordered
                (void)hipStreamSynchronize(side->s);
                return e;
            }
            if (use_bw) {
                // block size: the smallest that exceeds every listed chunk's max_words; wavefronts: what the LDS lets the chip hold
                const uint32_t max_len = G.uniform ? G.u_wave_len : kWalkShortLen;
                const uint32_t max_full = (uint32_t)(((uint64_t)max_len * 25u + 31u) >> 5);
                const uint32_t min_len = G.uniform ? G.u_wave_len : G.rag_bw_min_len;
                auto run_bw = [&](auto btag, unsigned waves_per_cu) {
                    constexpr uint32_t B = decltype(btag)::value;
                    const uint32_t bmax = bwb * (kWalkBlockWords / B);
                    const unsigned grid = 256u * waves_per_cu;
                    // header lists behind info[] (sized for the smallest block: par_walk_scratch_bytes())
                    const uint32_t hop_cap = bw_hop_cap(B, min_len, G.k);
                    uint32_t *hops = hop_cap ? reinterpret_cast<uint32_t *>(info + (uint64_t)n_bw * bwb * (kWalkBlockWords / 1024u)) : nullptr;
                    if (hops)
                        k_bw_blocks<B, false, true><<<grid, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, nullptr, nullptr, nullptr, d_status, hops, hop_cap);
                    else
                        k_bw_blocks<B, false><<<grid, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, nullptr, nullptr, nullptr, d_status, nullptr, 0u);
                    k_bw_scan<B><<<n_bw, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, bw_fail);
                    if (hops)
                        k_bw_emit<B><<<256u * 8u, 256, 0, s>>>(G, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, bw_fail, hops, hop_cap, d_wave_off, d_wave_words, d_status);
                    else
                        k_bw_blocks<B, true><<<grid, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, bw_fail, d_wave_off, d_wave_words, d_status, nullptr, 0u);
                };
                if (max_full + 2u <= 1024u) run_bw(std::integral_constant<uint32_t, 1024>{}, 24u);
                else if (max_full + 2u <= 2048u) run_bw(std::integral_constant<uint32_t, 2048>{}, 13u);
                else run_bw(std::integral_constant<uint32_t, 4096>{}, 7u);
                k_walk_block_only<<<(unsigned)G.n_chunks, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_status, bw_fail);
            }
            if (split) {
                mark(ev, 1, s);  // (the block walk's end; the other stream is decoding already)
                Geom Gs = G;
                Gs.rag_order = G.rag_order + G.rag_groups_long;
                Gs.rag_groups = G.rag_groups - G.rag_groups_long;
                launch_lanes(Gs, Gs.rag_groups, im_split, s);
                lanes_done = true;
            }
            if (forked && (e = hipStreamWaitEvent(s, side->join, 0)) != hipSuccess) {
                (void)hipStreamSynchronize(side->s);
                return e;
            }
            if (impl == 5) impl = 1;
            if (impl == 8) impl = 7;
        } else if (G.uniform) {
            if (G.u_wave_len <= kWalkShortLen)
                k_walk_block<<<(unsigned)G.n_chunks, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, nullptr,
                                                                 (uint32_t)G.n_chunks, d_wave_off, d_wave_words, d_status);
            else
                k_walk_scalar<<<blocks_for(G.n_chunks, kWalkChains), 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off,
                                                                                 d_wave_off, d_wave_words, d_status);
        } else {
            if (G.n_short) k_walk_block<<<G.n_short, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, G.walk_short, G.n_short,
                                                                 d_wave_off, d_wave_words, d_status);
            if (G.n_long) k_walk_list<<<blocks_for(G.n_long, 64), 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, G.walk_long,
                                                                             G.n_long, d_wave_off, d_wave_words, d_status);
        }
        if (lanes_done) {
            mark(ev, 2, s);
            mark(ev, 3, s);
            return hipGetLastError();
        }
        mark(ev, 1, s);
        const unsigned nb_plain = blocks_for(G.total_waves, 64);
        if (blocks_path) {
            // a workgroup per block of every waveform (drx_blocks.hip); waveforms it flags are decoded again, one
            // workgroup each, by the kernel that also judges them
            const uint32_t *fail = nullptr, *suspect = nullptr;
            path |= 4u | (gen ? 32u : 0u);  // DRX_PATH_BLOCKS (| DRX_PATH_IIR)
            hipError_t e = launch_decode_blocks(G, d_in, in_words, d_wave_off, d_wave_words, d_blk, d_status, d_out, &fail, &suspect, gen, s);
            if (e != hipSuccess) return e;
            k_decode_long<<<(unsigned)G.total_waves, kLongThreads, 0, s>>>(G, d_in, d_wave_off, d_wave_words, d_status, d_out, fail, suspect, gen ? 1u : 0u);
            if (gen) {
                // residuals -> samples, in place; then the waveforms the block decoder flagged, serially (a slope-1 ramp)
                if ((e = launch_iir(G, G.iir_chunk_tile_base, G.iir_n_tiles, G.iir_tab, G.iir_state, fail, d_status, d_out, s)) != hipSuccess) return e;
                k_decode_simple<<<nb_plain, 64, 0, s>>>(G, d_in, d_wave_off, d_wave_words, d_status, d_out, fail);
            }
            mark(ev, 2, s);
            mark(ev, 3, s);
            return hipGetLastError();
        }
        if (long_path) {  // (flag 512, or a long-waveform batch the block decoder does not take: one workgroup per waveform)
            path |= 8u;  // DRX_PATH_LONG
            k_decode_long<<<(unsigned)G.total_waves, kLongThreads, 0, s>>>(G, d_in, d_wave_off, d_wave_words, d_status, d_out, nullptr, nullptr, 0u);
            mark(ev, 2, s);
            mark(ev, 3, s);
            return hipGetLastError();
        }
        if (impl == 5) impl = 1;  // (a batch that cannot take the in-launch walk)
        if (impl == 8) impl = 7;
        const unsigned nb = (!G.uniform && G.rag_order) ? G.rag_groups : nb_plain;  // (groups per chunk round up)
        if (impl == 0) path |= 16u;  // DRX_PATH_SIMPLE
        if (impl == 0)
            k_decode_simple<<<nb_plain, 64, 0, s>>>(G, d_in, d_wave_off, d_wave_words, d_status, d_out, nullptr);
        else
            launch_lanes(G, nb, impl, s);
    }
    mark(ev, 2, s);
    mark(ev, 3, s);
    return hipGetLastError();
}

}  // namespace drx
