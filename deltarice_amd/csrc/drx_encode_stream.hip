// drx_encode_stream.hip -- the headline encoder of round 4: k_encode_stream, a persistent single-pass encoder in which no
// wavefront ever waits for its place in the stream while it has something to code (gfx950 / MI355X).
//
// What it replaces, and why.  k_encode_fused (drx_encode_kernels.hip) gives a workgroup eight waveforms, codes them into eight
// LDS buffers, and then needs the sum over ALL earlier waveforms' sizes before a single word can leave the CU
// (src/deltaRice.c:427-432 does that compaction with a serial memcpy loop).  Its decoupled look-back finds the sum quickly,
// but only once every earlier workgroup has FINISHED coding: 1.0-1.5 ms of its 5.5 ms on the headline workload were eight
// wavefronts holding 66 KB of LDS and doing nothing (profiles/r03_notes.md section 2, profiles/r04_notes.md section 1), and
// pricing the obvious cure -- a second set of buffers per workgroup -- showed that the halved occupancy costs more than the
// wait (8 wavefronts per CU: 6.1-6.6 ms without any look-back against 4.2-4.5 at 16; profiles/r04_enc_occupancy.txt).
//
// The form here keeps 16 wavefronts per CU and the same 8 KB of LDS per wavefront, and changes three things:
//   * A wavefront is on its own.  It takes the next waveform of the batch (an LDS counter per workgroup; one global ticket per
//     eight waveforms, as before, so that every earlier waveform is already being coded by a RUNNING wavefront), codes it,
//     publishes its size, and goes on to the next waveform at once.  No workgroup barrier after the prologue: a slow
//     wavefront no longer holds seven others.
//   * Its LDS buffer is a RING.  Waveform A's code stays where it is while waveform B is coded behind it; A is copied out when
//     its place is known -- looked up without waiting at every third tile of B -- and B only has to stop when it would run
//     into A (with 1414-word waveforms in a 2048-word ring: six tiles, ~8 us, into B).
//   * The prefix sum is somebody else's job.  ONE workgroup of the grid (whichever draws role 0: it is running by definition)
//     does nothing but sweep the published sizes in order, up to 1024 per round trip, and publishes every waveform's exclusive
//     prefix; a coder needs one 8-byte load for its place instead of a look-back of its own.
// Output bytes are those of k_encode_fused (and of src/deltaRice.c): the waveform's place is the same prefix sum.
//
// Words of the look-back state (d_scan, zeroed before the launch): size[W] | place[W] | ticket, role.  An entry is its own
// flag (bit 63), written and polled with relaxed agent-scope atomics, as in every look-back of this library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "drx_internal.h"
#include "drx_device.h"
#include "drx_encode.h"

namespace drx {

constexpr int kEsWaves = 8;             // wavefronts per workgroup (two workgroups per CU)
constexpr uint32_t kEsFront = 4;        // pad words in front of a ring (place_words writes up to four words below a lane's last)
constexpr uint32_t kEsBack = 12;        // ... and behind it (emit_tile runs up to eight codes past a lane's first word)
constexpr uint32_t kEsBatch = 8;        // waveforms per global ticket
constexpr uint32_t kEsSlots = 4;        // tickets a workgroup keeps (ring of LDS slots)
constexpr uint64_t kEsFlag = 1ull << 63;
// ring words per wavefront: 16 x (2496 + 16) x 4 bytes = 157 KB, two workgroups per CU (the geometry fused_wide() = 1 already
// runs); a 1414-word waveform (the headline's) leaves the next one 1072 words = nine tiles before it has to know its place
#ifndef DRX_ES_RING
#define DRX_ES_RING 2496
#endif
constexpr uint32_t kEsRing = DRX_ES_RING;
constexpr uint32_t kEsScanThreads = 64 * kEsWaves, kEsScanPer = 2;  // the scanner's window: 1024 entries per round trip

// Diagnostic build (-DDRX_ENC_STAMPS, never shipped): every wavefront counts in registers and adds its counts to eight
// shared counters when it leaves (an atomic per event ran the kernel at the rate of one counter).
//   0 ticks (100 MHz) spent in wait_place   1 waits   2 waveforms placed by a look-up between tile groups   3 waveforms
//   4 waveforms streamed   5 ticks from kernel start to the wavefront's exit   6 ticks spent taking waveforms
#ifdef DRX_ENC_STAMPS
#define ES_COUNT(i, v) do { es_cnt[i] += (unsigned long long)(v); } while (0)
#else
#define ES_COUNT(i, v) do { } while (0)
#endif

__device__ __forceinline__ uint64_t es_load(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void es_store(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// the scanner: one workgroup, sizes in -> exclusive prefixes out, in order
// ---------------------------------------------------------------------------
__device__ __forceinline__ void es_scanner(uint64_t total, const uint64_t *__restrict__ size, uint64_t *__restrict__ place,
                                           DevStatus *st, uint32_t (*s_lead)[kEsWaves], uint64_t (*s_sum)[kEsWaves]) {
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint64_t pos = 0, running = 0;
    uint32_t idle = 0;
    while (pos < total) {
        uint64_t v[kEsScanPer];
        bool pub[kEsScanPer];
#pragma unroll
        for (int j = 0; j < (int)kEsScanPer; ++j) {
            const uint64_t e = pos + (uint64_t)kEsScanThreads * j + threadIdx.x;
            v[j] = e < total ? es_load(size + e) : kEsFlag;  // (beyond the batch: an empty entry; the run is cut at `total`)
            pub[j] = (v[j] >> 63) != 0;
        }
#pragma unroll
        for (int j = 0; j < (int)kEsScanPer; ++j) {
            const uint64_t b = __ballot(pub[j]);
            if (lane == 0) s_lead[j][wv] = (b == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~b);
        }
        __syncthreads();
        uint64_t r = 0;  // entries from pos on that have all been published
        bool open = true;
#pragma unroll
        for (int j = 0; j < (int)kEsScanPer; ++j)
#pragma unroll
            for (int w = 0; w < kEsWaves; ++w) {
                const uint32_t l = s_lead[j][w];
                r += open ? l : 0u;
                open = open && l == 64u;
            }
        r = r < total - pos ? r : total - pos;
        if (r == 0) {
            // nothing new: the frontier's owner is still coding.  (Bounded like every wait of this library: ~seconds.)
            if (++idle > (1u << 22)) {
                if (threadIdx.x == 0) atomicOr(&st->err, kErrInternal);
                return;
            }
            __builtin_amdgcn_s_sleep(2);
            __syncthreads();
            continue;
        }
        idle = 0;
        uint64_t inc[kEsScanPer], val[kEsScanPer];
#pragma unroll
        for (int j = 0; j < (int)kEsScanPer; ++j) {
            const uint64_t rel = (uint64_t)kEsScanThreads * j + threadIdx.x;
            val[j] = rel < r ? (v[j] & ~kEsFlag) : 0ull;
            uint64_t x = val[j];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint64_t t = __shfl_up(x, d);
                if (lane >= d) x += t;
            }
            inc[j] = x;
            if (lane == 63) s_sum[j][wv] = x;
        }
        __syncthreads();
        uint64_t before[kEsScanPer], all = 0;
#pragma unroll
        for (int j = 0; j < (int)kEsScanPer; ++j) {
            before[j] = all;
#pragma unroll
            for (int w = 0; w < kEsWaves; ++w) {
                before[j] += (w < wv) ? s_sum[j][w] : 0ull;
                all += s_sum[j][w];
            }
        }
#pragma unroll
        for (int j = 0; j < (int)kEsScanPer; ++j) {
            const uint64_t rel = (uint64_t)kEsScanThreads * j + threadIdx.x;
            if (rel < r) es_store(place + pos + rel, kEsFlag | (running + before[j] + inc[j] - val[j]));
        }
        running += all;
        pos += r;
        __syncthreads();  // (s_lead / s_sum are rewritten by the next round)
    }
}

// ---------------------------------------------------------------------------
// the coder
// ---------------------------------------------------------------------------
template <bool GEN, uint32_t RING>
__global__ __launch_bounds__(64 * kEsWaves, 4) void k_encode_stream(Geom G, const int16_t *__restrict__ in,
                                                                    uint32_t *__restrict__ out, uint64_t out_cap,
                                                                    uint64_t *__restrict__ chunk_word_off,
                                                                    uint32_t *__restrict__ wave_words, uint64_t *__restrict__ size,
                                                                    uint64_t *__restrict__ place, uint32_t *__restrict__ ctrl,
                                                                    DevStatus *st, unsigned long long *prof) {
    __shared__ __attribute__((aligned(16))) uint32_t ring_all[kEsWaves][kEsFront + RING + kEsBack];
    __shared__ uint32_t s_role, s_cnt;
    __shared__ uint32_t s_read[kEsSlots];
    __shared__ uint64_t s_tick[kEsSlots];
    __shared__ uint32_t s_lead[kEsScanPer][kEsWaves];
    __shared__ uint64_t s_sum[kEsScanPer][kEsWaves];
    static_assert((kEsFront + RING + kEsBack) % 4 == 0 && RING % 4 == 0, "16-byte LDS accesses");
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    typedef uint64_t __attribute__((address_space(3))) lds_u64;
    constexpr uint32_t kRingBits = RING * 32u;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint32_t *row = ring_all[wv];
    uint32_t *ring = row + kEsFront;
    const uint32_t ring_bits0 = lds_addr(ring) * 8u;

    if (threadIdx.x == 0) {
        s_role = atomicAdd(ctrl + 1, 1u);
        s_cnt = 0;
    }
    if (threadIdx.x < kEsSlots) { s_read[threadIdx.x] = 0; s_tick[threadIdx.x] = 0; }
    for (int i = lane; i < (int)(kEsFront + RING + kEsBack) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (s_role == 0) {  // the first workgroup to start sweeps the sizes; everybody else codes
        es_scanner(G.total_waves, size, place, st, s_lead, s_sum);
        return;
    }

    const uint32_t k = G.k;
    const u16x2 tp[4] = {splat(GEN ? G.enc_t[0] : 1u), splat(GEN ? G.enc_t[1] : 0xffffu), splat(GEN ? G.enc_t[2] : 0u),
                         splat(GEN ? G.enc_t[3] : 0u)};
#ifdef DRX_ENC_STAMPS
    unsigned long long es_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t es_t0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- the waveform whose code waits in the ring for its place (wave uniform) ----
    bool pend = false;
    uint64_t gA = 0;
    uint32_t startA = 0, nA = 0;

    // copies waveform A out (its place is `ex` words into the stream) and clears its part of the ring
    auto copy_out = [&](uint64_t ex) {
        const WaveRef rA = locate(G, gA);
        const uint64_t mine = 1ull + nA + (rA.idx == 0 ? 1ull : 0ull);
        const uint64_t pos = ex + (rA.idx == 0 ? 1ull : 0ull);  // the waveform's header word
        if (lane == 0) {
            if (rA.idx == 0) chunk_word_off[rA.chunk] = ex;
            if (gA + 1 == G.total_waves) {
                chunk_word_off[G.n_chunks] = ex + mine;
                st->total_words = ex + mine;
                if (ex + mine > out_cap) atomicOr(&st->err, kErrCapacity);
            }
        }
        const bool room = pos + 1u + nA <= out_cap;  // (otherwise the last waveform raises kErrCapacity)
        if (room && lane == 0) {
            out[pos] = nA;                                 // :379
            if (rA.idx == 0) out[pos - 1] = rA.n_samples;  // chunk header, :415
        }
        uint32_t *__restrict__ outp = out + pos + 1;
        typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
        typedef u32x4s __attribute__((address_space(1), aligned(4))) g_u32x4_a4;
        const uint32_t n4 = nA & ~3u;
        for (uint32_t i = 4u * (uint32_t)lane; i < n4; i += 256u) {
            uint32_t w = startA + i;
            w = w >= RING ? w - RING : w;  // (starts are multiples of four words: a 16-byte piece never straddles the end)
            const uint4 v = *reinterpret_cast<const uint4 *>(ring + w);
            if (room) *(g_u32x4_a4 *)(outp + i) = (u32x4s){v.x, v.y, v.z, v.w};
            *reinterpret_cast<uint4 *>(ring + w) = make_uint4(0, 0, 0, 0);
        }
        if ((uint32_t)lane < nA - n4) {
            uint32_t w = startA + n4 + (uint32_t)lane;
            w = w >= RING ? w - RING : w;
            if (room) outp[n4 + (uint32_t)lane] = ring[w];
            ring[w] = 0;
        }
        wave_sync();
        pend = false;
    };
    // the place of waveform gw, waiting for it (bounded; a wait that expires reports kErrInternal and returns a place
    // nothing is written to)
    auto wait_place = [&](uint64_t gw) -> uint64_t {
        uint32_t spins = 0;
#ifdef DRX_ENC_STAMPS
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (;;) {
            uint64_t v = 0;
            if (lane == 0) v = es_load(place + gw);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
            if (hi >> 31) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
#ifdef DRX_ENC_STAMPS
                ES_COUNT(0, __builtin_amdgcn_s_memrealtime() - t0);
                ES_COUNT(1, 1);
#endif
                return ((uint64_t)(hi & 0x7fffffffu) << 32) | lo;
            }
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) {
                if (lane == 0) atomicOr(&st->err, kErrInternal);
                return out_cap;  // (room == false for every waveform: nothing is stored)
            }
        }
    };

    uint32_t start = 0;  // first ring word of the waveform being coded (a multiple of four)
    for (;;) {
        // ---- the next waveform of the batch ----
#ifdef DRX_ENC_STAMPS
        const uint64_t es_t1 = __builtin_amdgcn_s_memrealtime();
#endif
        uint32_t c = 0;
        if (lane == 0) c = __hip_atomic_fetch_add((lds_u32 *)&s_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        const uint32_t b = c / kEsBatch, j = c % kEsBatch, slot = b % kEsSlots;
        uint32_t T = 0;
        if (j == 0) {
            // the slot's previous ticket (batch b - kEsSlots) has been read by all its takers?
            const uint32_t want = kEsBatch * (b / kEsSlots);
            while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((lds_u32 *)&s_read[slot], __ATOMIC_RELAXED,
                                                                                  __HIP_MEMORY_SCOPE_WORKGROUP)) != want)
                __builtin_amdgcn_s_sleep(1);
            if (lane == 0) T = atomicAdd(ctrl, 1u);
            T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
            if (lane == 0)
                __hip_atomic_store((lds_u64 *)&s_tick[slot], ((uint64_t)(b + 1u) << 32) | T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            for (;;) {
                const uint64_t v = __hip_atomic_load((lds_u64 *)&s_tick[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
                if (tag == b + 1u) { T = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (lane == 0) __hip_atomic_fetch_add((lds_u32 *)&s_read[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t g = (uint64_t)T * kEsBatch + j;
        ES_COUNT(6, __builtin_amdgcn_s_memrealtime() - es_t1);
        if (g >= G.total_waves) break;

        WaveRef r = locate(G, g);
        const int16_t *x = in + r.sample_off;
        const uint32_t wlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.len);

        // words this waveform may take: up to the waveform in front of it in the ring, or the whole ring
        auto gap = [&]() -> uint32_t {
            if (!pend) return RING - 8u;
            const uint32_t d = startA >= start ? startA - start : startA + RING - start;
            return d > 8u ? d - 8u : 0u;
        };
        if (pend && gap() < 512u) copy_out(wait_place(gA));  // (short waveforms behind a long one: no room to start)
        uint32_t limit = gap();

        uint64_t P = 0;        // bits so far (wave uniform)
        bool fits = true;      // everything so far is in the ring (wave uniform)
        uint32_t carry = 0;    // dword whose high half is the sample before the tile (x[-1] := 0, :53-54)
        uint32_t carry2 = 0;   // GEN: the dword before that one
        auto wrap = [&](uint32_t bits) -> uint32_t { return bits >= kRingBits ? bits - kRingBits : bits; };
        auto process_tile = [&](const uint32_t (&w)[4], int nv, auto full_tag) {
            constexpr bool FULLT = decltype(full_tag)::value;
            uint32_t xprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], 0x138, 0xf, 0xf, false);  // wave_shr:1
            if (lane == 0) xprev = carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)w[3], 63);
            uint32_t xprev2 = 0;
            if (GEN) {
                xprev2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[2], 0x138, 0xf, 0xf, false);
                if (lane == 0) xprev2 = carry2;
                carry2 = (uint32_t)__builtin_amdgcn_readlane((int)w[2], 63);
            }
            PackedCodes cc;
            packed_codes<GEN>(w, xprev, xprev2, tp, k, cc);
            if (!FULLT) mask_tail(cc, nv);
            const uint32_t lane_bits = lane_tile_bits(cc);
            uint32_t cw[4];
            if (FULLT) concat_codes(cc, cw);  // independent of the scan: fills its DPP wait states
            const uint32_t incl = wave_incl_scan_dpp(lane_bits);
            const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (fits && ((P + tile_bits + 31u) >> 5) < (uint64_t)limit) {
                const uint32_t s0 = start * 32u + (uint32_t)P;  // the tile's first bit, from the ring's word 0, not wrapped
                if (FULLT && !__any(lane_bits > 128u))
                    place_words(cw, ring_bits0 + wrap(s0 + incl));
                else
                    emit_tile<FULLT>(cc, ring_bits0 + wrap(s0 + incl - lane_bits));
                if (s0 < kRingBits && s0 + tile_bits >= kRingBits) {
                    // the tile ran across the end of the ring: what its lanes wrote into the pads belongs to the other end
                    wave_sync();
                    if (lane < (int)kEsFront) {
                        const uint32_t v = row[lane];
                        if (v) { ring[RING - kEsFront + lane] |= v; row[lane] = 0; }
                    } else if (lane < (int)(kEsFront + kEsBack)) {
                        const uint32_t i = (uint32_t)lane - kEsFront, v = ring[RING + i];
                        if (v) { ring[i] |= v; ring[RING + i] = 0; }
                    }
                    wave_sync();
                }
            } else {
                fits = false;
            }
            P += tile_bits;
        };
        const uint32_t n_full = wlen / kTile;
        // between tile groups: is waveform A's place known by now?  The load travels while a group is coded.
        uint64_t pollv = 0;
        bool polling = false;
        auto between = [&](uint32_t tiles_done) {
            if (!pend) return;
            if (polling) {
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pollv >> 32));
                polling = false;
                if (hi >> 31) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pollv);
                    ES_COUNT(2, 1);
                    copy_out(((uint64_t)(hi & 0x7fffffffu) << 32) | lo);
                    limit = gap();
                    return;
                }
            }
            // would the next group run into A?  (an estimate from the waveform's own rate so far; if it is wrong the tile
            // test above keeps the ring intact and the waveform takes the streaming path)
            const uint32_t used = (uint32_t)((P + 31u) >> 5);
            // (tiles until the next look-up: a group, and behind the last group the trailing partial tile)
            const uint32_t left = n_full - tiles_done, next = left > 3u ? 3u : left + 1u;
            const uint32_t est = tiles_done ? ((used * next) / tiles_done) * 9u / 8u + 24u : 512u;
            if (fits && used + est >= limit) {
                copy_out(wait_place(gA));
                limit = gap();
                return;
            }
            if (lane == 0) pollv = es_load(place + gA);
            polling = true;
        };

        {
            constexpr int kDepth = 3;
            const uint4 *xv = reinterpret_cast<const uint4 *>(x) + lane;  // tile t: xv[64 * t]
            uint4 q[kDepth];
            uint32_t t = 0;
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                q[u] = make_uint4(0, 0, 0, 0);
                if ((uint32_t)u < n_full) q[u] = xv[64 * (size_t)u];
            }
#pragma unroll 1
            for (; t + 2u * kDepth <= n_full; t += kDepth) {
                if (t) between(t);
#pragma unroll
                for (int u = 0; u < kDepth; ++u) {
                    const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                    process_tile(w, 8, std::true_type{});
                    q[u] = xv[64 * (size_t)(t + u + kDepth)];
                }
            }
#pragma unroll 1
            for (; t < n_full; t += kDepth) {
                if (t) between(t);
#pragma unroll
                for (int u = 0; u < kDepth; ++u) {
                    if (t + (uint32_t)u < n_full) {
                        const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                        process_tile(w, 8, std::true_type{});
                        if (t + (uint32_t)u + kDepth < n_full) q[u] = xv[64 * (size_t)(t + u + kDepth)];
                    }
                }
            }
            // the trailing partial tile
            for (uint32_t t0 = n_full * kTile; t0 < wlen; t0 += kTile) {
                uint32_t w[4];
                const int nv = load8_dwords(x, wlen, t0, lane, true, w);
                process_tile(w, nv, std::false_type{});
            }
        }
        const uint32_t n = (uint32_t)((P + 31u) >> 5);  // payload words n_i
        wave_sync();
        if (lane == 0) {
            wave_words[g] = n;
            es_store(size + g, kEsFlag | (1ull + n + (r.idx == 0 ? 1ull : 0ull)));
        }
        ES_COUNT(3, 1);
        if (pend) copy_out(wait_place(gA));  // (waveforms too short for the look-ups between tile groups)
        if (fits) {
            pend = true;
            gA = g;
            startA = start;
            nA = n;
            start += (n + 3u) & ~3u;
            start = start >= RING ? start - RING : start;
            continue;
        }

        // ---- the code did not fit the ring: stream it tile by tile to its final position (second read of the samples) ----
        ES_COUNT(4, 1);
        const uint64_t ex = wait_place(g);
        const uint64_t mine = 1ull + n + (r.idx == 0 ? 1ull : 0ull);
        const uint64_t pos = ex + (r.idx == 0 ? 1ull : 0ull);
        if (lane == 0) {
            if (r.idx == 0) chunk_word_off[r.chunk] = ex;
            if (g + 1 == G.total_waves) {
                chunk_word_off[G.n_chunks] = ex + mine;
                st->total_words = ex + mine;
                if (ex + mine > out_cap) atomicOr(&st->err, kErrCapacity);
            }
        }
        for (int i = lane; i < (int)(kEsFront + RING + kEsBack) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
        start = 0;
        wave_sync();
        if (pos + 1u + n > out_cap) continue;  // the last waveform raises kErrCapacity
        if (lane == 0) {
            out[pos] = n;
            if (r.idx == 0) out[pos - 1] = r.n_samples;
        }
        uint32_t *__restrict__ outp = out + pos + 1;
        uint32_t *buf = ring;
        P = 0;
        carry = 0;
        carry2 = 0;
        uint32_t wn[4];
        int nvn = wlen ? load8_dwords(x, wlen, 0u, lane, true, wn) : 0;
        for (uint32_t t0 = 0; t0 < wlen; t0 += kTile) {
            uint32_t w[4] = {wn[0], wn[1], wn[2], wn[3]};
            const int nv = nvn;
            if (t0 + kTile < wlen) nvn = load8_dwords(x, wlen, t0 + kTile, lane, true, wn);  // (travels while this tile is coded)
            uint32_t xprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], 0x138, 0xf, 0xf, false);
            if (lane == 0) xprev = carry;
            carry = (uint32_t)__shfl((int)w[3], 63);
            uint32_t xprev2 = 0;
            if (GEN) {
                xprev2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[2], 0x138, 0xf, 0xf, false);
                if (lane == 0) xprev2 = carry2;
                carry2 = (uint32_t)__shfl((int)w[2], 63);
            }
            PackedCodes cc;
            packed_codes<GEN>(w, xprev, xprev2, tp, k, cc);
            mask_tail(cc, nv);
            const uint32_t lane_bits = lane_tile_bits(cc);
            const uint32_t incl = wave_incl_scan_dpp(lane_bits);
            const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint64_t w0 = P >> 5;  // first staged word
            emit_tile<false>(cc, ring_bits0 + (uint32_t)(P & 31u) + incl - lane_bits);
            P += tile_bits;
            wave_sync();
            const uint32_t nfull = (uint32_t)((P >> 5) - w0);
            for (uint32_t i = lane; i < nfull; i += 64) { outp[w0 + i] = buf[i]; buf[i] = 0; }
            wave_sync();
            if (nfull && lane == 0) { const uint32_t cwd = buf[nfull]; buf[nfull] = 0; buf[0] = cwd; }
            wave_sync();
        }
        if (lane == 0) {
            if (P & 31u) outp[P >> 5] = buf[0];
            buf[0] = 0;
        }
        wave_sync();
    }
    if (pend) copy_out(wait_place(gA));
#ifdef DRX_ENC_STAMPS
    es_cnt[5] = __builtin_amdgcn_s_memrealtime() - es_t0;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(prof + i, es_cnt[i]);
#endif
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
// d_scan: uint64[2 * total_waves + 10], zeroed here on the stream before every launch (the last eight words are the
// diagnostic build's counters and are left alone).
hipError_t launch_encode_stream(const Geom &G, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                                uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint64_t *d_scan,
                                DevStatus *d_status, hipEvent_t *ev, hipStream_t s) {
    if (G.total_waves == 0) return hipSuccess;
    const uint64_t W = G.total_waves;
    mark(ev, 0, s);
    hipError_t e = hipMemsetAsync(d_scan, 0, (2 * W + 2) * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    mark(ev, 1, s);
    mark(ev, 2, s);
    uint64_t *size = d_scan, *place = d_scan + W;
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(d_scan + 2 * W);
    unsigned long long *prof = reinterpret_cast<unsigned long long *>(d_scan + 2 * W + 2);
    // persistent: two workgroups per CU (66 KB of LDS each), one of them the scanner; never more than the batch can feed
    const uint64_t tickets = (W + kEsBatch - 1) / kEsBatch;
    // (debug flag 262144: three workgroups -- a scanner and two coders -- so that a small test batch takes every wavefront
    // through many waveforms, i.e. around its ring)
    const unsigned grid = (G.dbg & 262144u) ? 3u : (unsigned)(tickets + 1 < 512u ? tickets + 1 : 512u);
    if (G.n_taps)
        k_encode_stream<true, kEsRing><<<grid, 64 * kEsWaves, 0, s>>>(G, d_in, d_out, out_cap, d_chunk_word_off, d_wave_words, size, place,
                                                                          ctrl, d_status, prof);
    else
        k_encode_stream<false, kEsRing><<<grid, 64 * kEsWaves, 0, s>>>(G, d_in, d_out, out_cap, d_chunk_word_off, d_wave_words, size, place,
                                                                           ctrl, d_status, prof);
    mark(ev, 3, s);
#ifdef DRX_ENC_STAMPS
    {
        unsigned long long h[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, prof, sizeof h, hipMemcpyDeviceToHost);
        (void)hipMemset(prof, 0, sizeof h);
        const double wf = h[3] ? (double)h[3] : 1.0;
        fprintf(stderr, "enc stream stamps: %llu waveforms: %llu placed between tile groups without waiting, %llu waits (%.2f us each), %llu streamed; "
                "per waveform: %.2f us in all, %.2f us waiting for the place, %.2f us taking the waveform\n",
                h[3], h[2], h[1], h[1] ? h[0] / (double)h[1] / 100.0 : 0.0, h[4], h[5] / wf / 100.0, h[0] / wf / 100.0, h[6] / wf / 100.0);
    }
#endif
    return hipGetLastError();
}

}  // namespace drx
