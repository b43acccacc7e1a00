// drx_iir.hip -- the inverse of a general prediction filter, IN PLACE over decoded residuals, parallel inside a waveform.
//
// The reference's decodeWaveform (src/deltaRice.c:91-102) is a serial recurrence per waveform,
//     y[i] = (d[i] - sum_{j=1..n-1} taps[j] y[i-j]) / taps[0]            (every partial result truncated to int16),
// and its docs recommend such filters for exactly the batches that have no waveforms to spare: NOPTREX, 32 waveforms of
// 500 000 samples per chunk, taps [1,-1,1,-1] (docs/Optimization.md:21, docs/Performance.md:38).  The lane-per-waveform
// decoder takes ~60 ns per sample and lane whatever else runs: 30 ms for such a waveform, whatever the batch.  For filters
// the fast kernels take (at most four taps, taps[0] = +-1: no division) the recurrence is LINEAR over Z / 2^16,
//     s_i = A s_{i-1} + e_1 d'_i,      s_i = (y[i], y[i-1], y[i-2]),      A = [[c1 c2 c3] [1 0 0] [0 1 0]],
// (c_j = -+taps[j], d' = +-d), so a run of samples maps its incoming state to its outgoing state by an AFFINE map whose
// matrix depends only on the run's LENGTH.  The block-parallel decoder (drx_blocks.hip, RESID) therefore leaves the residuals
// themselves in the output buffer, and this kernel turns them into samples:
//   tile      a workgroup of 512 lanes x 64 consecutive samples = 32 768 samples of one waveform, tiles in ticket order (one
//             global ticket counter serves ~88 workgroups per microsecond: 8192- and 16 384-sample tiles ran at exactly that
//             rate); two workgroups per CU, so that one's memory phases run under the other's arithmetic (1024 lanes x 32
//             samples, one workgroup per CU: 0.9 ms for 671 M samples; 512 x 32: 0.72 ms; this form: see profiles/r03_notes.md);
//   pass 1    every lane runs the recurrence over its M = 64 residuals from a ZERO state: the zero-state response's final state;
//   scan      Hillis-Steele over the lanes of a wavefront with the matrices A^(M o), o = 1, 2, 4 ... 32 (all lanes of a
//             step use the same matrix: equal run lengths), then over the sixteen wavefronts: the state in front of every lane
//             given a zero state in front of the tile, and the tile's own zero-state response B;
//   look-back the state in front of the tile: x = B_{t-1} + P B_{t-2} + P^2 B_{t-3} + ... (P = A^32768) down to the nearest
//             tile that has published its full state -- a decoupled look-back over 8-byte {status | 3 x 16 bit} entries, the
//             powers of P from a table (equal run lengths again: no matrix travels between workgroups);
//   pass 2    every lane runs the recurrence again from its true state and stores samples over its residuals.
// Two passes of ~5 VALU instructions per sample, 2 + 2 bytes of HBM traffic per sample.  All arithmetic is modulo 2^16 (32-bit
// registers whose upper halves are never looked at; 24-bit multiplies: the low 16 bits of a product depend on the low 16
// bits of its factors alone), exactly the int16 truncations of the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "drx_device.h"
#include "drx_internal.h"
#include "drx_iir_math.h"

namespace drx {

// table layout (uint32 each): PL[7][9] = A^(M * 2^d), d = 0..6 | PLANE[64][9] = A^(M l) | PTP[kIirWin + 1][9] = P^j | c1 c2 c3 sgn
constexpr uint32_t kIirPL = 0, kIirPLANE = 7 * 9, kIirPTP = kIirPLANE + 64 * 9, kIirC = kIirPTP + (kIirWin + 1) * 9;
static_assert(kIirC + 4 == kIirTabWords, "table layout");

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    return v;
}

// tile u -> waveform and tile inside it.  Tiles are numbered waveform-major; a chunk's waveforms all have tiles_full tiles but
// possibly its last one (src/deltaRice.c:420-425).
struct TileRef { uint64_t g; uint32_t t, first_dist; };
__device__ __forceinline__ TileRef locate_tile(const Geom &G, const uint64_t *__restrict__ chunk_tile_base, uint64_t u) {
    uint64_t c, r, wave_base;
    uint32_t L, W, N;
    if (G.uniform) {
        L = G.u_wave_len; W = G.u_n_waves; N = G.u_n_samples;
        const uint64_t per_chunk = (uint64_t)(W - 1u) * ((L + kIirTile - 1u) / kIirTile) + ((N - (W - 1u) * L + kIirTile - 1u) / kIirTile);
        c = u / per_chunk;
        r = u - c * per_chunk;
        wave_base = c * W;
    } else {
        uint64_t lo = 0, hi = G.n_chunks;  // invariant: chunk_tile_base[lo] <= u < chunk_tile_base[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (chunk_tile_base[mid] <= u) lo = mid; else hi = mid;
        }
        c = lo;
        r = u - chunk_tile_base[c];
        const ChunkDesc d = G.chunks[c];
        L = d.wave_len; W = d.n_waves; N = d.n_samples;
        wave_base = d.wave_base;
    }
    const uint32_t tf = (L + kIirTile - 1u) / kIirTile;
    TileRef q;
    const uint64_t full = (uint64_t)(W - 1u) * tf;
    const uint32_t idx = r < full ? (uint32_t)(r / tf) : W - 1u;
    q.t = (uint32_t)(r - (uint64_t)idx * tf);
    q.g = wave_base + idx;
    q.first_dist = q.t;  // entries u - 1 .. u - t belong to this waveform
    return q;
}

__global__ __launch_bounds__(kIirThreads) void k_iir_tiles(Geom G, const uint64_t *__restrict__ chunk_tile_base, uint64_t n_tiles,
                                                           const uint32_t *__restrict__ tab, uint64_t *__restrict__ state,
                                                           uint32_t *__restrict__ ticket, const uint32_t *__restrict__ skip,
                                                           DevStatus *st, int16_t *__restrict__ out) {
    constexpr int M = kIirRun, NWV = kIirThreads / 64;
    // A lane owns M consecutive samples (2 M bytes), so lane-private loads would touch 64 different lines per instruction.
    // A tile that lies wholly inside its waveform therefore moves through LDS: every wavefront loads its 128 M contiguous
    // bytes in 1 KB instructions (16-byte piece p to lane p mod 64), writes the pieces to rows of kRowB bytes (a lane's
    // 2 M bytes + 16 of padding: with a row stride of 36 / 68 banks both the linear writes and the row-wise 16-byte reads
    // are free of bank conflicts), and every lane reads its own row; the samples go back the same way.
    constexpr int NP = M / 8;                      // 16-byte pieces per lane
    constexpr uint32_t kRowB = 2u * M + 16u;       // bytes per row
    __shared__ __attribute__((aligned(16))) uint8_t s_t[NWV][64 * kRowB];
    __shared__ uint32_t s_u;
    __shared__ uint32_t s_F[NWV][3], s_x[3];
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id(), wv = (int)(tid >> 6);
    if (tid == 0) s_u = atomicAdd(ticket, 1u);  // every lower tile is held by a running (or finished) workgroup
    __syncthreads();
    const uint64_t u = s_u;
    if (u >= n_tiles) return;
    const TileRef q = locate_tile(G, chunk_tile_base, u);
    const WaveRef r = locate(G, q.g);
    int16_t *y = out + r.sample_off;
    const uint32_t c1 = tab[kIirC], c2 = tab[kIirC + 1], c3 = tab[kIirC + 2], sg = tab[kIirC + 3];

    // ---- my M residuals, two per dword ----
    const uint32_t i0 = q.t * kIirTile + tid * (uint32_t)M;
    const uint32_t nv = i0 >= r.len ? 0u : (r.len - i0 < (uint32_t)M ? r.len - i0 : (uint32_t)M);
    uint32_t d[M / 2];
    const bool whole = (uint64_t)(q.t + 1u) * kIirTile <= (uint64_t)r.len;  // (uniform over the workgroup)
    int16_t *const wbase = y + (q.t * kIirTile + (uint32_t)wv * 64u * (uint32_t)M);  // my wavefront's 64 M samples
    uint8_t *const rows = s_t[wv];
    if (whole) {
        uint4 v[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) v[j] = *reinterpret_cast<const uint4 *>(wbase + 8 * (j * 64 + lane));  // (any int16 alignment: unaligned access is on)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t p = (uint32_t)(j * 64 + lane);
            *reinterpret_cast<uint4 *>(rows + (p / NP) * kRowB + (p % NP) * 16u) = v[j];
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint4 w = *reinterpret_cast<const uint4 *>(rows + (uint32_t)lane * kRowB + 16u * j);
            d[4 * j] = w.x; d[4 * j + 1] = w.y; d[4 * j + 2] = w.z; d[4 * j + 3] = w.w;
        }
        wave_sync();
    } else if (nv == (uint32_t)M) {
#pragma unroll
        for (int j = 0; j < M / 8; ++j) {
            const uint4 v = *reinterpret_cast<const uint4 *>(y + i0 + 8 * j);
            d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < M / 2; ++j) {
            const uint32_t a = (2u * j < nv) ? (uint32_t)(uint16_t)y[i0 + 2 * j] : 0u;
            const uint32_t b = (2u * j + 1u < nv) ? (uint32_t)(uint16_t)y[i0 + 2 * j + 1] : 0u;
            d[j] = a | (b << 16);
        }
    }
    // the recurrence over my run from state s; EMIT: the samples replace the residuals in d[]
    auto run = [&](V3 s, auto emit_tag) __attribute__((always_inline)) {
        constexpr bool EMIT = decltype(emit_tag)::value;
#pragma unroll
        for (int j = 0; j < M / 2; ++j) {
            const uint32_t a = __umul24(d[j] & 0xffffu, sg) + __umul24(c1, s.x) + __umul24(c2, s.y) + __umul24(c3, s.z);
            const uint32_t b = __umul24(d[j] >> 16, sg) + __umul24(c1, a) + __umul24(c2, s.x) + __umul24(c3, s.y);
            s = V3{b, a, s.x};
            if (EMIT) d[j] = __builtin_amdgcn_perm(b, a, 0x05040100u);
        }
        return s;
    };

    // ---- pass 1: zero-state response of my run; scan over the wavefront, then over the workgroup ----
    V3 F = lo16(run(V3{0u, 0u, 0u}, std::false_type{}));
#pragma unroll
    for (int dd = 0; dd < 6; ++dd) {
        const M3 P = load_m3(tab + kIirPL + 9 * dd);  // A^(M * 2^dd)
        const V3 up = shfl_up_v3(F, 1 << dd);
        if (lane >= (1 << dd)) F = lo16(add(F, mul(P, up)));
    }
    V3 E = shfl_up_v3(F, 1);  // zero-state state in front of my run, inside the wavefront
    if (lane == 0) E = V3{0u, 0u, 0u};
    if (lane == 63) { s_F[wv][0] = F.x; s_F[wv][1] = F.y; s_F[wv][2] = F.z; }
    __syncthreads();
    const M3 PW = load_m3(tab + kIirPL + 9 * 6);  // A^(M * 64): one wavefront
    if (wv == 0) {
        // the tile's zero-state response B, then the state in front of the tile
        V3 B{0u, 0u, 0u};
#pragma unroll
        for (int w = 0; w < NWV; ++w) B = lo16(add(mul(PW, B), V3{s_F[w][0], s_F[w][1], s_F[w][2]}));
        const uint64_t bval = (uint64_t)B.x | ((uint64_t)B.y << 16) | ((uint64_t)B.z << 32);
        V3 x{0u, 0u, 0u};
        if (q.t == 0) {
            if (lane == 0) __hip_atomic_store(state + u, kScanPrefix | bval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(state + u, kScanAgg | bval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int64_t first = (int64_t)u - (int64_t)q.first_dist;  // tile 0 of this waveform
            int64_t base = (int64_t)u - 1;
            M3 R;  // P^(kIirWin * windows walked so far)
#pragma unroll
            for (int i = 0; i < 9; ++i) R.m[i] = (i % 4 == 0) ? 1u : 0u;
            const M3 Pwin = load_m3(tab + kIirPTP + 9 * kIirWin);
            uint32_t spins = 0;
            for (;;) {  // the nearest predecessor alone first: one 8-byte load per poll (see k_encode_fused)
                uint64_t v = 0;
                if (lane == 0) v = __hip_atomic_load(state + base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 62)) != 0) break;
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1u << 22)) break;  // (the window loop below reports it)
            }
            for (;;) {
                // lane l looks at predecessors base - l (nearer) and base - 64 - l (farther); in front of tile 0 the state is zero
                const int64_t j0 = base - lane, j1 = base - 64 - lane;
                uint64_t s0v = kScanPrefix, s1v = kScanPrefix;
                if (j0 >= first) s0v = __hip_atomic_load(state + j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (j1 >= first) s1v = __hip_atomic_load(state + j1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t st0 = (uint32_t)(s0v >> 62), st1 = (uint32_t)(s1v >> 62);
                const uint64_t p0 = __ballot(st0 == 2u), z0 = __ballot(st0 == 0u);
                const uint64_t p1 = __ballot(st1 == 2u), z1 = __ballot(st1 == 0u);
                const int fp = p0 ? __builtin_ctzll(p0) : (p1 ? 64 + __builtin_ctzll(p1) : 128);  // nearest full state
                const uint64_t near0 = fp >= 64 ? ~0ull : ((1ull << fp) - 1ull);
                const uint64_t near1 = fp >= 128 ? ~0ull : (fp > 64 ? ((1ull << (fp - 64)) - 1ull) : 0ull);
                if ((z0 & near0) | (z1 & near1)) {  // a nearer predecessor has not published yet
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { if (lane == 0) atomicOr(&st->err, kErrInternal); break; }  // cannot happen; never hang
                    continue;
                }
                // a tile at distance dist contributes P^dist x its value (aggregate, or the full state that ends the sum)
                V3 c{0u, 0u, 0u};
                if (lane <= fp) c = mul(load_m3(tab + kIirPTP + 9 * lane), V3{(uint32_t)s0v, (uint32_t)(s0v >> 16), (uint32_t)(s0v >> 32)});
                if (64 + lane <= fp) c = add(c, mul(load_m3(tab + kIirPTP + 9 * (64 + lane)), V3{(uint32_t)s1v, (uint32_t)(s1v >> 16), (uint32_t)(s1v >> 32)}));
                c = mul(R, lo16(c));
                x = lo16(add(x, V3{wave_sum_u32(c.x), wave_sum_u32(c.y), wave_sum_u32(c.z)}));
                if (fp < 128) break;
                base -= 128;
                R = mul(Pwin, R);
            }
            // the full state behind this tile: its successors stop here
            const V3 X = lo16(add(mul(load_m3(tab + kIirPTP + 9), x), B));
            if (lane == 0)
                __hip_atomic_store(state + u, kScanPrefix | ((uint64_t)X.x | ((uint64_t)X.y << 16) | ((uint64_t)X.z << 32)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) { s_x[0] = x.x; s_x[1] = x.y; s_x[2] = x.z; }
    }
    __syncthreads();
    if (skip && skip[q.g]) return;  // a waveform the block decoder flagged: decoded again, serially, by the kernel behind this one

    // ---- pass 2: my true state, the samples ----
    V3 XW{s_x[0], s_x[1], s_x[2]};  // in front of my wavefront
    for (int w = 0; w < wv; ++w) XW = lo16(add(mul(PW, XW), V3{s_F[w][0], s_F[w][1], s_F[w][2]}));
    const V3 S = lo16(add(mul(load_m3(tab + kIirPLANE + 9 * lane), XW), E));
    (void)run(S, std::true_type{});
    if (whole) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
            *reinterpret_cast<uint4 *>(rows + (uint32_t)lane * kRowB + 16u * j) = make_uint4(d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3]);
        wave_sync();
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t p = (uint32_t)(j * 64 + lane);
            const uint4 w = *reinterpret_cast<const uint4 *>(rows + (p / NP) * kRowB + (p % NP) * 16u);
            *reinterpret_cast<uint4 *>(wbase + 8 * p) = w;
        }
    } else if (nv == (uint32_t)M) {
#pragma unroll
        for (int j = 0; j < M / 8; ++j)
            *reinterpret_cast<uint4 *>(y + i0 + 8 * j) = make_uint4(d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < M / 2; ++j) {
            if (2u * j < nv) y[i0 + 2 * j] = (int16_t)(uint16_t)d[j];
            if (2u * j + 1u < nv) y[i0 + 2 * j + 1] = (int16_t)(uint16_t)(d[j] >> 16);
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// the tables of a filter (kIirTabWords uint32): powers of the companion matrix modulo 2^16
void iir_tables(const uint32_t fast_nt[3], uint32_t t0neg, uint32_t *tab) {
    typedef uint32_t Mat[9];
    auto mmul = [](const Mat a, const Mat b, Mat r) {
        Mat t;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) t[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j]) & 0xffffu;
        for (int i = 0; i < 9; ++i) r[i] = t[i];
    };
    const uint32_t sg = t0neg ? 0xffffu : 1u;
    const uint32_t c[3] = {(fast_nt[0] * sg) & 0xffffu, (fast_nt[1] * sg) & 0xffffu, (fast_nt[2] * sg) & 0xffffu};
    Mat A = {c[0], c[1], c[2], 1, 0, 0, 0, 1, 0}, I = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    Mat P;  // A^M: one lane's run
    for (int i = 0; i < 9; ++i) P[i] = I[i];
    for (uint32_t i = 0; i < kIirRun; ++i) mmul(A, P, P);
    Mat cur;
    for (int i = 0; i < 9; ++i) cur[i] = P[i];
    for (int d = 0; d < 7; ++d) {  // A^(M * 2^d)
        for (int i = 0; i < 9; ++i) tab[kIirPL + 9 * d + i] = cur[i];
        mmul(cur, cur, cur);
    }
    for (int i = 0; i < 9; ++i) cur[i] = I[i];
    for (int l = 0; l < 64; ++l) {  // A^(M l)
        for (int i = 0; i < 9; ++i) tab[kIirPLANE + 9 * l + i] = cur[i];
        mmul(P, cur, cur);
    }
    Mat PT;  // A^tile = (A^(M * 64))^(threads / 64)
    for (int i = 0; i < 9; ++i) PT[i] = I[i];
    for (uint32_t w = 0; w < kIirThreads / 64u; ++w) mmul(tab + kIirPL + 9 * 6, PT, PT);
    for (int i = 0; i < 9; ++i) cur[i] = I[i];
    for (uint32_t j = 0; j <= kIirWin; ++j) {  // P^j
        for (int i = 0; i < 9; ++i) tab[kIirPTP + 9 * j + i] = cur[i];
        mmul(PT, cur, cur);
    }
    tab[kIirC] = c[0]; tab[kIirC + 1] = c[1]; tab[kIirC + 2] = c[2]; tab[kIirC + 3] = sg;
}

uint64_t iir_tiles(const Geom &G, const ChunkDesc *host_chunks, uint64_t *chunk_tile_base) {
    auto per = [](uint32_t N, uint32_t L, uint32_t W) {
        return (uint64_t)(W - 1u) * ((L + kIirTile - 1u) / kIirTile) + ((N - (W - 1u) * L + kIirTile - 1u) / kIirTile);
    };
    if (G.uniform) return G.n_chunks * per(G.u_n_samples, G.u_wave_len, G.u_n_waves);
    uint64_t t = 0;
    for (uint64_t c = 0; c < G.n_chunks; ++c) {
        if (chunk_tile_base) chunk_tile_base[c] = t;
        t += per(host_chunks[c].n_samples, host_chunks[c].wave_len, host_chunks[c].n_waves);
    }
    if (chunk_tile_base) chunk_tile_base[G.n_chunks] = t;
    return t;
}

// d_state: uint64[n_tiles] + a ticket word behind it, zeroed here
hipError_t launch_iir(const Geom &G, const uint64_t *d_chunk_tile_base, uint64_t n_tiles, const uint32_t *d_tab, uint64_t *d_state,
                      const uint32_t *d_skip, DevStatus *d_status, int16_t *d_out, hipStream_t s) {
    if (!n_tiles) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_state, 0, (n_tiles + 1) * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    uint32_t *ticket = reinterpret_cast<uint32_t *>(d_state + n_tiles);
    k_iir_tiles<<<(unsigned)n_tiles, kIirThreads, 0, s>>>(G, d_chunk_tile_base, n_tiles, d_tab, d_state, ticket, d_skip, d_status, d_out);
    return hipGetLastError();
}

}  // namespace drx
