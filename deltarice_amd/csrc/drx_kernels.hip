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

#include "drx_internal.h"

namespace drx {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ void wave_sync() {
    // All lanes of a wave run in lock step and its LDS operations complete in order;
    // this only stops the compiler from moving LDS accesses across a phase boundary.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        uint32_t t = __shfl_xor(v, d);
        v = t > v ? t : v;
    }
    return v;
}

struct WaveRef {
    uint64_t chunk;       // chunk index
    uint64_t sample_off;  // first sample of this waveform in the raw batch
    uint32_t idx;         // waveform index inside its chunk
    uint32_t len;         // samples in this waveform
    uint32_t n_samples;   // samples in the chunk
};

// waveform g -> chunk and extent.  Uniform batches are pure arithmetic; ragged
// batches bisect the chunk table (once per waveform, i.e. once per ~L samples).
__device__ __forceinline__ WaveRef locate(const Geom &G, uint64_t g) {
    WaveRef r;
    uint32_t L, W;
    uint64_t c, soff;
    if (G.uniform) {
        c = g / G.u_n_waves;
        r.idx = (uint32_t)(g - c * G.u_n_waves);
        L = G.u_wave_len;
        W = G.u_n_waves;
        r.n_samples = G.u_n_samples;
        soff = c * (uint64_t)G.u_n_samples;
    } else {
        uint64_t lo = 0, hi = G.n_chunks;  // invariant: wave_base[lo] <= g < wave_base[hi]
        while (hi - lo > 1) {
            uint64_t mid = (lo + hi) >> 1;
            if (G.chunks[mid].wave_base <= g) lo = mid; else hi = mid;
        }
        c = lo;
        const ChunkDesc d = G.chunks[c];
        r.idx = (uint32_t)(g - d.wave_base);
        L = d.wave_len;
        W = d.n_waves;
        r.n_samples = d.n_samples;
        soff = d.sample_off;
    }
    r.chunk = c;
    r.sample_off = soff + (uint64_t)r.idx * L;
    r.len = (r.idx + 1 == W) ? (r.n_samples - r.idx * L) : L;  // trailing partial waveform (:420-425)
    return r;
}

// ---------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------
constexpr int kTile = 512;        // samples per wave tile: 64 lanes x 8 samples (16 B per lane)
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
            const int32_t d = (int32_t)(int16_t)(v[j] - prev);
            prev = v[j];
            uint32_t code, nb;
            rice_code(d, k, code, nb);
            bits += (j < nv) ? nb : 0u;
        }
    }
    const uint64_t total = wave_sum_u64(bits);
    if (lane == 0) wave_words[g] = (uint32_t)((total + 31u) >> 5);
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
            const int32_t d = (int32_t)(int16_t)(v[j] - prev);
            prev = v[j];
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
// decode
// ---------------------------------------------------------------------------

// Header chain walk (:320-325) with validation; one lane per chunk.
__global__ __launch_bounds__(64) void k_walk(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                             const uint64_t *__restrict__ chunk_word_off,
                                             uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                             DevStatus *st) {
    const uint64_t c = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (c >= G.n_chunks) return;
    uint64_t base;
    uint32_t W, L, N;
    if (G.uniform) { base = c * G.u_n_waves; W = G.u_n_waves; L = G.u_wave_len; N = G.u_n_samples; }
    else { const ChunkDesc d = G.chunks[c]; base = d.wave_base; W = d.n_waves; L = d.wave_len; N = d.n_samples; }
    const uint64_t begin = chunk_word_off[c];
    uint64_t end = chunk_word_off[c + 1];
    bool bad = false;
    if (end > in_words || begin + 2 > end) { bad = true; end = begin; }
    if (!bad && in[begin] != N) bad = true;  // :306 totalNumberPoints
    uint64_t at = begin + 1;
    for (uint32_t w = 0; w < W; ++w) {
        uint32_t n = 0;
        uint64_t here = at;
        if (!bad && at < end) {
            n = in[at];
            const uint32_t len = (w + 1 == W) ? (N - w * L) : L;
            const uint64_t max_words = ((uint64_t)len * 25u + 31u) >> 5;
            if (n > max_words || at + 1u + n > end) { bad = true; n = 0; }
            else at += (uint64_t)n + 1u;
        } else {
            bad = true;
            here = begin;  // keeps later loads in bounds; decoded as zero words
        }
        wave_off[base + w] = here;
        wave_words[base + w] = n;
    }
    if (!bad && at != end) bad = true;
    if (bad) atomicOr(&st->err, kErrCorrupt);
}

// Straightforward lane-per-waveform decoder: global loads and 2-byte stores.
// Kept as the simple cross-check of the staged kernel below (decode_impl = 0).
__global__ __launch_bounds__(64) void k_decode_simple(Geom G, const uint32_t *__restrict__ in,
                                                      const uint64_t *__restrict__ wave_off,
                                                      const uint32_t *__restrict__ wave_words,
                                                      int16_t *__restrict__ out) {
    const uint64_t g = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (g >= G.total_waves) return;
    const WaveRef r = locate(G, g);
    const uint32_t *s = in + wave_off[g] + 1;
    const uint32_t n = wave_words[g];
    int16_t *y = out + r.sample_off;
    const uint32_t k = G.k;
    uint64_t win = 0;
    uint32_t have = 0, wi = 0;
    int32_t acc = 0;
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
        acc += (int32_t)(z >> 1) ^ -(int32_t)(z & 1u);  // un-zig-zag (:172-177), running sum (:80-89)
        y[i] = (int16_t)acc;
        const uint32_t used = q + 1u + pl;
        win <<= used;
        have -= used;
    }
}

// Staged lane-per-waveform decoder.
//   RW   ring words per stream (power of two >= 64): compressed words live in LDS at
//        ring[lane][word_index mod RW]; whole 128-byte lines (32 words) are loaded by
//        8 lanes x 16 B each, 8 streams per wave instruction.
//   T    samples decoded per round; after a round the 64 x T tile is written out with
//        16-byte stores, 8 lanes covering T=64 samples (128 B) of one waveform.
template <int RW, int T>
__global__ __launch_bounds__(64) void k_decode_lanes(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                     const uint64_t *__restrict__ wave_off,
                                                     const uint32_t *__restrict__ wave_words,
                                                     int16_t *__restrict__ out) {
    constexpr int RS = RW + 1;     // ring row stride in words (odd: conflict-free for equal indices)
    constexpr int OSW = T / 2 + 1; // output row stride in words (odd)
    constexpr int PPS = T / 8;     // 16-byte pieces per stream per round
    constexpr int SPI = 64 / PPS;  // streams per write-out iteration
    __shared__ uint32_t ring[64 * RS];
    __shared__ uint32_t obuf[64 * OSW];
    __shared__ uint64_t tab_ooff[64];
    __shared__ uint32_t tab_len[64];

    const int lane = lane_id();
    const uint64_t g = (uint64_t)blockIdx.x * 64u + lane;
    const bool active = g < G.total_waves;
    const uint32_t k = G.k;

    uint32_t len = 0, n = 0;
    uint64_t S = 0, ooff = 0;
    if (active) {
        const WaveRef r = locate(G, g);
        len = r.len;
        ooff = r.sample_off;
        S = wave_off[g] + 1u;
        n = wave_words[g];
    }
    tab_ooff[lane] = ooff;
    tab_len[lane] = len;
    const uint64_t A = S & ~(uint64_t)(RW - 1);   // ring-aligned base (word index): ring slot = (w - A) mod RW
    const uint32_t s0 = (uint32_t)(S - A);        // first payload word, relative to A
    const uint32_t endw = s0 + n;                  // one past the last payload word, relative
    uint32_t flw = s0 & ~31u;                      // words loaded so far (whole 128-byte lines), relative
    uint32_t rdw = s0;                             // next word to move into the window
    const bool in_vec_ok = ((uintptr_t)in & 15u) == 0;
    const uint32_t maxlen = wave_max_u32(len);
    uint32_t *myring = ring + lane * RS;
    typedef uint16_t __attribute__((may_alias)) u16a;
    u16a *myout = reinterpret_cast<u16a *>(obuf + lane * OSW);

    // Loads one more line for every stream that has room for it and data left.
    auto top_up = [&]() {
        for (;;) {
            // room: the line would overwrite words [flw-RW, flw-RW+32), all of which must be consumed
            const bool need = (flw < endw) && (flw <= rdw || flw - rdw + 32u <= (uint32_t)RW);
            const uint64_t mask = __ballot(need);
            if (mask == 0) break;
            const uint64_t nx = A + flw;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int s = i * 8 + (lane >> 3), p = lane & 7;
                const uint64_t nxs = __shfl(nx, s);
                if ((mask >> s) & 1u) {
                    const uint64_t a = nxs + 4u * (uint32_t)p;
                    uint4 v;
                    if (in_vec_ok && a + 4u <= in_words) {
                        v = *reinterpret_cast<const uint4 *>(in + a);
                    } else {
                        v.x = (a + 0u < in_words) ? in[a + 0u] : 0u;
                        v.y = (a + 1u < in_words) ? in[a + 1u] : 0u;
                        v.z = (a + 2u < in_words) ? in[a + 2u] : 0u;
                        v.w = (a + 3u < in_words) ? in[a + 3u] : 0u;
                    }
                    uint32_t *dst = ring + s * RS + ((uint32_t)nxs & (uint32_t)(RW - 1)) + 4 * p;
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                }
            }
            if (need) flw += 32u;
            wave_sync();
        }
    };

    wave_sync();
    top_up();
    // window: bits [o', ...) of the pair (hi, lo) with o in [1, 32]; o = 32 means "lo starts now"
    uint32_t hi = 0, lo = myring[rdw & (RW - 1)];
    ++rdw;
    uint32_t o = 32;
    int32_t acc = 0;

    for (uint32_t t0 = 0; t0 < maxlen; t0 += T) {
#pragma unroll 1
        for (int tg = 0; tg < T; tg += 4) {
            // a group of 4 samples consumes at most 4 words
            if (__any((flw - rdw < 5u) && (flw < endw))) top_up();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t win = __builtin_amdgcn_alignbit(hi, lo, 32u - o);
                uint32_t q = (uint32_t)__clz((int)win);
                const bool esc = q >= 8u;  // valid streams have q <= 8
                q = esc ? 8u : q;
                const uint32_t pl = esc ? 16u : k;
                const uint32_t used = q + 1u + pl;
                const uint32_t rem = __builtin_amdgcn_ubfe(win, 32u - used, pl);
                const uint32_t z = esc ? rem : ((q << k) + rem);
                acc += (int32_t)(z >> 1) ^ -(int32_t)(z & 1u);
                myout[tg + u] = (uint16_t)acc;
                o += used;
                if (o > 32u) {
                    o -= 32u;
                    hi = lo;
                    lo = myring[rdw & (RW - 1)];
                    ++rdw;
                }
            }
        }
        wave_sync();
        // write the 64 x T tile: PPS lanes cover one waveform's T samples
#pragma unroll
        for (int i = 0; i < PPS; ++i) {
            const int s = i * SPI + lane / PPS, p = lane % PPS;
            const uint32_t slen = tab_len[s];
            const uint32_t tpos = t0 + 8u * (uint32_t)p;
            if (tpos < slen) {
                const uint32_t *src = obuf + s * OSW + 4 * p;
                uint4 v;
                v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
                int16_t *dst = out + tab_ooff[s] + tpos;
                if (tpos + 8u <= slen && ((uintptr_t)dst & 15u) == 0) {
                    *reinterpret_cast<uint4 *>(dst) = v;
                } else {
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (tpos + (uint32_t)j < slen) dst[j] = (int16_t)(w[j >> 1] >> (16 * (j & 1)));
                }
            }
        }
        wave_sync();
    }
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

hipError_t launch_decode(const Geom &G, const uint32_t *d_in, uint64_t in_words,
                         const uint64_t *d_chunk_word_off, int16_t *d_out, uint64_t *d_wave_off,
                         uint32_t *d_wave_words, DevStatus *d_status, int impl, hipEvent_t *ev, hipStream_t s) {
    if (G.total_waves == 0) return hipSuccess;
    mark(ev, 0, s);
    k_walk<<<blocks_for(G.n_chunks, 64), 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off,
                                                     d_wave_words, d_status);
    mark(ev, 1, s);
    const unsigned nb = blocks_for(G.total_waves, 64);
    switch (impl) {
        case 0: k_decode_simple<<<nb, 64, 0, s>>>(G, d_in, d_wave_off, d_wave_words, d_out); break;
        case 2: k_decode_lanes<64, 32><<<nb, 64, 0, s>>>(G, d_in, in_words, d_wave_off, d_wave_words, d_out); break;
        case 3: k_decode_lanes<128, 64><<<nb, 64, 0, s>>>(G, d_in, in_words, d_wave_off, d_wave_words, d_out); break;
        default: k_decode_lanes<64, 64><<<nb, 64, 0, s>>>(G, d_in, in_words, d_wave_off, d_wave_words, d_out); break;
    }
    mark(ev, 2, s);
    mark(ev, 3, s);
    return hipGetLastError();
}

}  // namespace drx
