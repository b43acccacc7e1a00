// drx_encode_kernels.hip -- gfx950 (MI355X, CDNA4) ENCODE kernels of the Delta-Rice codec and their launchers
// (round 3 split drx_kernels.hip by role: this file, drx_walk.h, drx_decode_kernels.hip).
//
// Format contract (bit-exact with /root/reference/src/deltaRice.c):
//   chunk   := u32 N | { u32 n_i | u32 payload_i[n_i] }            (:415,379,427-433)
//   payload := MSB-first concatenation of one code per sample      (:229-241)
//   code(z) := (z>>k) zeros, '1', k bits      if (z>>k) < 8        (:215-222)
//              8 zeros, '1', 16 bits of z     otherwise            (:223-228)
//   z = zigzag(d), d_0 = x_0, d_j = x_j - x_{j-1} mod 2^16         (:51-63,207-211)
//
//
// Work decomposition (64-wide wavefronts, no MFMA: this is integer bit packing bounded by HBM bandwidth): one wavefront per
// waveform: 16-byte coalesced loads (8 samples/lane), wave prefix scan of code lengths -> bit offsets, codes OR-ed into an
// LDS staging buffer, whole words copied out coalesced; the waveform's place in the dense stream by a decoupled look-back.
// (The packed tile code: drx_encode.h; short and very long waveforms: drx_pieces.hip.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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

// Exclusive prefix over chunk totals -> chunk_word_off[0..n_chunks]; one workgroup.
__global__ __launch_bounds__(1024) void k_chunk_offsets(uint64_t n_chunks, const uint64_t *__restrict__ chunk_words,
                                                        uint64_t *__restrict__ chunk_word_off,
                                                        uint64_t out_cap, DevStatus *st, uint64_t *host_words) {
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
        if (host_words) *host_words = run;  // (Geom::host_words)
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

#ifndef DRX_ENC_WAVES
#define DRX_ENC_WAVES 8
#endif
constexpr int kEncWaves = DRX_ENC_WAVES;  // waveforms (wavefronts) per workgroup = per ticket
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
// WV x CAPW: waveforms per workgroup x LDS words per waveform.  8 x 2048 (9.3 bits per sample at WaveformLength 7000) is the
// measured geometry of the headline workload; noisier data with the RiceParameter that suits it needs k + 3.5 bits per
// sample, and a waveform whose code outgrows its buffer is coded twice -- 8 x 2496 (two workgroups per CU), 4 x 3072 (three)
// and 4 x 4096 words (two) keep the single pass for it (fused_wide(), drx_internal.h).  Six waveforms per workgroup were
// measured a third slower whatever the buffer (three of them on two SIMDs, one each on the others).
// Diagnostic build (-DDRX_ENC_STAMPS, never shipped): lane 0 of every workgroup's wave 0 adds the 100 MHz ticks between the
// phases of k_encode_fused into eight counters behind the look-back state; launch_encode_fused prints the shares.
//   0 encode (start -> all eight waveforms coded)   1 gate (nearest predecessor has published ANYTHING)
//   2 window (a prefix found behind it)   3 copy-out of wave 0
#ifdef DRX_ENC_STAMPS
// (a slot per workgroup in the upper half of the look-back state, which k_encode_fused does not use: atomics on eight shared
// counters ran at the rate of one counter -- ~88 per microsecond -- and turned a 5.5 ms kernel into an 8 ms one)
#define ENC_STAMP(i)                                                                        \
    do {                                                                                    \
        if (threadIdx.x == 0) {                                                             \
            const uint64_t t_now = __builtin_amdgcn_s_memrealtime();                        \
            scan_state[G.total_waves + 16 + (uint64_t)s_ticket * 4u + (i)] = t_now - t_prev; \
            t_prev = t_now;                                                                 \
        }                                                                                   \
    } while (0)
#else
#define ENC_STAMP(i) do { } while (0)
#endif

template <bool GEN, int WV = DRX_ENC_WAVES, uint32_t CAPW = kEncCapWords>
__global__ __launch_bounds__(64 * WV, DRX_ENC_WAVES_PER_EU) void k_encode_fused(Geom G, const int16_t *__restrict__ in,
                                                      uint32_t *__restrict__ out, uint64_t out_cap,
                                                      uint64_t *__restrict__ chunk_word_off,
                                                      uint32_t *__restrict__ wave_words,
                                                      uint64_t *__restrict__ scan_state,
                                                      uint32_t *__restrict__ ticket, DevStatus *st) {
    constexpr int kEncWaves = WV;             // (shadow the namespace-scope defaults)
    constexpr uint32_t kEncCapWords = CAPW;
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
#ifdef DRX_ENC_STAMPS
    uint64_t t_prev = __builtin_amdgcn_s_memrealtime();
#endif
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
    ENC_STAMP(0);
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
            ENC_STAMP(1);
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
        ENC_STAMP(2);
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
            if (G.host_words) *G.host_words = excl + mine;
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
#ifdef DRX_ENC_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ENC_STAMP(3);
#endif
        return;
    }

    // ---- the code did not fit the LDS buffer: stream it tile by tile to its final position ----
    for (int i = lane; i < (int)(kEncCapWords + 8) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
    wave_sync();
    P = 0;
    carry = 0;
    carry2 = 0;
    uint32_t wn[4];
    int nvn = r.len ? load8_dwords(x, r.len, 0u, lane, vec_ok, wn) : 0;
    for (uint32_t t0 = 0; t0 < r.len; t0 += kTile) {
        uint32_t w[4] = {wn[0], wn[1], wn[2], wn[3]};
        const int nv = nvn;
        if (t0 + kTile < r.len) nvn = load8_dwords(x, r.len, t0 + kTile, lane, vec_ok, wn);  // (travels while this tile is coded)
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
// launchers (host side, same translation unit so that <<<>>> stays in HIP code)
// ---------------------------------------------------------------------------
hipError_t launch_estimate_words(const Geom &G, const int16_t *d_in, unsigned long long *d_words16, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_words16, 0, 16 * sizeof(unsigned long long), s);
    if (e != hipSuccess || G.total_waves == 0) return e;
    k_estimate_words<<<blocks_for(G.total_waves, 4), 256, 0, s>>>(G, d_in, d_words16);
    return hipGetLastError();
}

// Ablation builds only: DRX_ENC_LDS_PAD = bytes of dynamic LDS added to every k_encode_fused launch (occupancy A/B at an
// unchanged instruction stream).
static unsigned enc_lds_pad() {
#ifdef DRX_ABLATION
    static const unsigned pad = [] { const char *e = getenv("DRX_ENC_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
    return pad;
#else
    return 0u;
#endif
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
    auto go = [&](auto gen_tag, auto wv_tag, auto cap_tag) {
        constexpr bool GEN = decltype(gen_tag)::value;
        constexpr int WV = decltype(wv_tag)::value;
        constexpr uint32_t CAPW = decltype(cap_tag)::value;
        k_encode_fused<GEN, WV, CAPW><<<blocks_for(G.total_waves, WV), 64 * WV, enc_lds_pad(), s>>>(G, d_in, d_out, out_cap, d_chunk_word_off, d_wave_words,
                                                                                    d_scan, ticket, d_status);
    };
    using std::integral_constant;
    const std::true_type T;
    const std::false_type F;
    auto pick = [&](auto wv_tag, auto cap_tag) { if (G.n_taps) go(T, wv_tag, cap_tag); else go(F, wv_tag, cap_tag); };
    switch (fused_wide(G)) {
        case 1: pick(integral_constant<int, 8>{}, integral_constant<uint32_t, kEncWide1Words>{}); break;
        case 2: pick(integral_constant<int, 4>{}, integral_constant<uint32_t, kEncWide2Words>{}); break;
        case 3: pick(integral_constant<int, 4>{}, integral_constant<uint32_t, kEncWide3Words>{}); break;
        default: pick(integral_constant<int, kEncWaves>{}, integral_constant<uint32_t, kEncCapWords>{}); break;
    }
    mark(ev, 3, s);
#ifdef DRX_ENC_STAMPS
    if (G.total_waves >= 64) {
        const uint64_t wgs = blocks_for(G.total_waves, kEncWaves);
        unsigned long long *h = (unsigned long long *)malloc(wgs * 4 * sizeof(unsigned long long));
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, d_scan + G.total_waves + 16, wgs * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum[4] = {0, 0, 0, 0};
        for (uint64_t i = 0; i < wgs; ++i)
            for (int j = 0; j < 4; ++j) sum[j] += (double)h[i * 4 + j];
        free(h);
        fprintf(stderr, "enc stamps (us per workgroup, 100 MHz ticks): encode %.2f  gate %.2f  window %.2f  copy-out %.2f  | %llu workgroups\n",
                sum[0] / wgs / 100.0, sum[1] / wgs / 100.0, sum[2] / wgs / 100.0, sum[3] / wgs / 100.0, (unsigned long long)wgs);
    }
#endif
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
    k_chunk_offsets<<<1, 1024, 0, s>>>(G.n_chunks, d_chunk_words, d_chunk_word_off, out_cap, d_status, G.host_words);
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
    k_chunk_offsets<<<1, 1024, 0, s>>>(G.n_chunks, d_chunk_words, d_chunk_word_off, out_cap, d_status, G.host_words);
    k_seg_zero<<<blocks_for(units, 256), 256, 0, s>>>(G, S, units, d_seg_pos, d_wave_rel, d_chunk_word_off, d_out, out_cap);
    mark(ev, 2, s);
    k_seg_pack<<<blocks_for(units, 4 * upw), 256, 0, s>>>(G, d_in, S, units, d_seg_pos, d_wave_words, d_wave_rel, d_chunk_word_off,
                                                          d_out, out_cap, upw);
    mark(ev, 3, s);
    return hipGetLastError();
}

}  // namespace drx
