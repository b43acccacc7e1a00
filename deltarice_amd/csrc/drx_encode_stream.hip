// drx_encode_stream.hip -- the headline encoder of round 4: k_encode_stream, a persistent single-pass encoder whose
// wavefronts do not wait for their place in the stream while they have something to code (gfx950 / MI355X).
//
// What it replaces, and why.  k_encode_fused (drx_encode_kernels.hip) gives a workgroup eight waveforms, codes them into eight
// LDS buffers, and then needs the sum over ALL earlier waveforms' sizes before a single word can leave the CU
// (src/deltaRice.c:427-432 does that compaction with a serial memcpy loop).  Its decoupled look-back finds the sum quickly,
// but only once every earlier workgroup has FINISHED coding: 1.0-1.5 ms of its 5.5 ms on the headline workload were eight
// wavefronts holding 66 KB of LDS and doing nothing (profiles/r03_notes.md section 2, profiles/r04_notes.md section 1), and
// pricing the obvious cure -- a second set of buffers per workgroup -- showed that the halved occupancy costs more than the
// wait (8 wavefronts per CU: 6.1-6.6 ms without any look-back against 4.2-4.5 at 16; profiles/r04_enc_occupancy.txt).
//
// The form here keeps 16 wavefronts per CU and ~10 KB of LDS per wavefront, and changes three things:
//   * The kernel is persistent and a wavefront's LDS buffer is a RING.  Waveform A's code stays where it is while the
//     wavefront codes its next waveform B behind it; A is copied out when its place is known -- looked up without waiting at
//     every third tile of B -- and B only has to stop when it would run into A (with 1414-word waveforms in a 2496-word
//     ring: nine tiles, ~10 us, into B).
//   * The prefix sum is somebody else's job.  ONE wavefront of the grid (in whichever workgroup draws role 0: it is running by
//     definition) does nothing but sweep the published sizes in order and publish exclusive prefixes; a coder needs one
//     8-byte load for its place instead of a look-back of its own.
//   * What is published and swept is one total per TICKET (a workgroup's WV waveforms, taken together at a rendezvous): an
//     agent-scope load costs the sweeping CU ~35 ns per 64 bytes beside the coders' streaming, and a sweep over one entry per
//     waveform was the whole kernel's rate (profiles/r04_notes.md section 1).
// Output bytes are those of k_encode_fused (and of src/deltaRice.c): a waveform's place is the same prefix sum.
//
// Look-back state (d_scan, zeroed before the launch): total[tickets] | place[tickets] | role counter, ticket counter (a
// 128-byte line each).  An entry is its own flag (bit 63), written and polled with relaxed agent-scope atomics, as in every
// look-back of this library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "drx_internal.h"
#include "drx_device.h"
#include "drx_encode.h"

namespace drx {

// wavefronts per workgroup = waveforms per ticket; 16 wavefronts per CU either way (4: 4.95 ms on the headline, 8: 5.1 -- the
// rendezvous of four costs less than that of eight; with the priority feedback below 4.80 / 4.84)
#ifndef DRX_ES_WAVES
#define DRX_ES_WAVES 4
#endif
constexpr int kEsWaves = DRX_ES_WAVES;
constexpr uint32_t kEsFront = 4;   // pad words in front of a ring (place_words writes up to four words below a lane's last)
constexpr uint32_t kEsBack = 12;   // ... and behind it (emit_tile runs up to eight codes past a lane's first word)
constexpr uint64_t kEsFlag = 1ull << 63;
constexpr uint32_t kEsCtrlWords = 32;  // uint64 words of control state: 128 bytes each for the role and the ticket counter
// ring words per wavefront: 16 x (2496 + 16) x 4 bytes = 157 KB, 16 wavefronts per CU (the geometry fused_wide() = 1 already
// runs); a 1414-word waveform (the headline's) leaves the next one 1072 words = nine tiles before it has to know its place
constexpr uint32_t kEsRing = kEsRingWords;  // (drx_internal.h: the dispatch needs it too)
#ifndef DRX_ES_PRIO
#define DRX_ES_PRIO 1
#endif
#ifndef DRX_ES_POLL_SLEEP
#define DRX_ES_POLL_SLEEP 4
#endif

// Diagnostic build (-DDRX_ENC_STAMPS, never shipped): every workgroup counts in LDS and adds its counts to eight shared
// counters when its wavefronts leave (a global atomic per event ran the kernel at the rate of one counter; counters in
// per-lane registers spilled).
//   0 ticks (100 MHz) spent in wait_place   1 waits   2 waveforms placed by a look-up between tile groups   3 waveforms
//   4 waveforms streamed   5 ticks from kernel start to the wavefront's exit   6 ticks at the rendezvous   7 failed polls
// ... and, with DRX_ES_TRACE=file in the environment, five 100 MHz stamps per waveform, written to the file after the launch
// (tools/r04_es_trace.py reads it): 1 begun, 2 size known, 3 place seen by its coder; in the slot of a ticket's first
// waveform also 0 the ticket's total published and 4 its place stored by the scanner
#ifdef DRX_ENC_STAMPS
__device__ uint64_t *g_es_trace = nullptr;
#define ES_COUNT(i, v) do { if (lane == 0) __hip_atomic_fetch_add((lds_u64 *)&s_prof[i], (uint64_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#define ES_TRACE(g, i) do { if (trace && lane == 0) trace[5ull * (g) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ES_COUNT(i, v) do { } while (0)
#define ES_TRACE(g, i) do { } while (0)
#endif

__device__ __forceinline__ uint64_t es_load(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void es_store(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// the scanner: ticket totals in -> exclusive prefixes out, in order
// ---------------------------------------------------------------------------
// ONE wavefront of the scanner workgroup, at the highest priority its CU gives: a window of kScPer x 64 entries from the
// frontier on per round trip, no barrier, no LDS.  (Forms measured before this one, profiles/r04_notes.md section 1, all
// over one entry per WAVEFORM: a 512-lane window with three workgroup barriers per round, eight wavefronts on a block of 512
// entries each chained through LDS, one wavefront with 1024- and 2048-entry windows -- a round took 5 / 7 / 5 / 16 us and
// the sweep, 100-200 entries per microsecond, was the kernel's rate every time.)
#ifndef DRX_ES_SCAN_PER
#define DRX_ES_SCAN_PER 8
#endif
constexpr uint32_t kScPer = DRX_ES_SCAN_PER;

__device__ __forceinline__ void es_scanner(uint64_t total, uint32_t wv_per_ticket, const uint64_t *__restrict__ size,
                                           uint64_t *__restrict__ place, DevStatus *st, unsigned long long *prof, uint64_t *trace) {
    const int lane = lane_id();
    if (threadIdx.x >= 64) return;
    __builtin_amdgcn_s_setprio(3);
    uint64_t pos = 0, base = 0;  // the frontier: every entry below pos has its place; base = words in front of entry pos
    uint32_t idle = 0;
#ifdef DRX_ENC_STAMPS
    unsigned long long sc_rounds = 0, sc_idle = 0, sc_full = 0;
    const uint64_t sc_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto sample = [&](uint64_t (&v)[kScPer], uint64_t at) {
#pragma unroll
        for (int j = 0; j < (int)kScPer; ++j) {
            const uint64_t e = at + 64u * j + (uint32_t)lane;
            v[j] = e < total ? es_load(size + e) : kEsFlag;  // (beyond the batch: an empty entry; the run is cut at `total`)
        }
    };
    while (pos < total) {
#ifdef DRX_ENC_STAMPS
        ++sc_rounds;
#endif
        uint64_t v[kScPer];
        const uint64_t bs = pos;
        sample(v, pos);
        uint64_t run = 0;  // entries from bs on that have their place or have been published
        bool open = true, big = false;
#pragma unroll
        for (int j = 0; j < (int)kScPer; ++j) {
            const uint64_t e = bs + 64u * j + (uint32_t)lane;
            const uint64_t m = __ballot(e < pos || (v[j] >> 63) != 0);
            const uint32_t l = (m == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~m);
            run += open ? l : 0u;
            open = open && l == 64u;
            big = big || __any((v[j] & ~kEsFlag) >= (1ull << 24));
        }
        uint64_t end = bs + run;  // the new frontier
        end = end < total ? end : total;
        if (end <= pos) {
#ifdef DRX_ENC_STAMPS
            ++sc_idle;
#endif
            if (++idle > (1u << 22)) {  // (bounded like every wait of this library: seconds)
                if (lane == 0) atomicOr(&st->err, kErrInternal);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        idle = 0;
#ifdef DRX_ENC_STAMPS
        if (end - bs == 64u * kScPer) ++sc_full;
#endif
        uint64_t running = 0;
#pragma unroll
        for (int j = 0; j < (int)kScPer; ++j) {
            if (bs + 64u * j < end) {  // (wave uniform)
                const uint64_t e = bs + 64u * j + (uint32_t)lane;
                const bool in = e >= pos && e < end;
                const uint64_t val = in ? (v[j] & ~kEsFlag) : 0ull;
                uint64_t inc;
                if (!big) {
                    inc = wave_incl_scan_dpp((uint32_t)val);
                } else {
                    inc = val;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint64_t t = __shfl_up(inc, d);
                        if (lane >= d) inc += t;
                    }
                }
                if (in) es_store(place + e, kEsFlag | (base + running + inc - val));
#ifdef DRX_ENC_STAMPS
                if (in && trace) trace[5ull * e * wv_per_ticket + 4] = __builtin_amdgcn_s_memrealtime();
#endif
                running += big ? __shfl(inc, 63) : (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)inc, 63);
            }
        }
        base += running;
        pos = end;
    }
#ifdef DRX_ENC_STAMPS
    if (lane == 0) {
        prof[8] = sc_rounds; prof[9] = sc_idle; prof[10] = sc_full;
        prof[11] = __builtin_amdgcn_s_memrealtime() - sc_t0;
    }
#endif
}

// ---------------------------------------------------------------------------
// the coders
// ---------------------------------------------------------------------------
template <bool GEN, int WV, uint32_t RING>
__global__ __launch_bounds__(64 * WV, 4) void k_encode_stream(Geom G, const int16_t *__restrict__ in, uint32_t *__restrict__ out,
                                                              uint64_t out_cap, uint64_t *__restrict__ chunk_word_off,
                                                              uint32_t *__restrict__ wave_words, uint64_t *__restrict__ size,
                                                              uint64_t *__restrict__ place, uint32_t *__restrict__ ctrl,
                                                              DevStatus *st, unsigned long long *prof) {
    __shared__ __attribute__((aligned(16))) uint32_t ring_all[WV][kEsFront + RING + kEsBack];
    __shared__ uint32_t s_role, s_arrive, s_ticket[4];
    __shared__ uint64_t s_mine[2][WV];              // sizes of the waveforms of the last two tickets
    __shared__ uint64_t s_place_v[2];               // place of a ticket, once one wavefront has seen it ...
    __shared__ uint32_t s_place_t[2];               // ... and which ticket (+ 1) that was
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    typedef uint64_t __attribute__((address_space(3))) lds_u64;
#ifdef DRX_ENC_STAMPS
    __shared__ uint64_t s_prof[8];
    if (threadIdx.x < 8) s_prof[threadIdx.x] = 0;
    uint64_t *trace = g_es_trace;
#else
    uint64_t *trace = nullptr;
#endif
    static_assert((kEsFront + RING + kEsBack) % 4 == 0 && RING % 4 == 0, "16-byte LDS accesses");
    constexpr uint32_t kRingBits = RING * 32u;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint32_t *row = ring_all[wv];
    uint32_t *ring = row + kEsFront;
    const uint32_t ring_bits0 = lds_addr(ring) * 8u;

    if (threadIdx.x == 0) {
        s_role = atomicAdd(ctrl, 1u);
        s_arrive = 0;
        s_place_t[0] = s_place_t[1] = 0;
    }
    for (int i = lane; i < (int)(kEsFront + RING + kEsBack) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint64_t n_tickets = (G.total_waves + WV - 1) / WV;
    if (s_role == 0) {  // the first workgroup to start sweeps the totals; everybody else codes
        es_scanner(n_tickets, (uint32_t)WV, size, place, st, prof, trace);
        return;
    }

    const uint32_t k = G.k;
    const u16x2 tp[4] = {splat(GEN ? G.enc_t[0] : 1u), splat(GEN ? G.enc_t[1] : 0xffffu), splat(GEN ? G.enc_t[2] : 0u),
                         splat(GEN ? G.enc_t[3] : 0u)};
#ifdef DRX_ENC_STAMPS
    const uint64_t es_t0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- waveform A: coded, waiting in the ring (or, if it did not fit, not at all) for its place.  Wave uniform. ----
    bool pend = false, fitsA = true;
    uint64_t gA = 0, offA = 0;   // its index; words of its ticket in front of it (known behind the next rendezvous)
    uint32_t TA = 0, startA = 0, nA = 0;

    // header words and bookkeeping of a waveform whose place is known; returns whether its words may be stored
    auto place_header = [&](const WaveRef &r, uint64_t g, uint32_t n, uint64_t ex, uint64_t &pos) -> bool {
        const uint64_t mine = 1ull + n + (r.idx == 0 ? 1ull : 0ull);
        pos = ex + (r.idx == 0 ? 1ull : 0ull);  // the waveform's header word
        if (lane == 0) {
            if (r.idx == 0) chunk_word_off[r.chunk] = ex;
            if (g + 1 == G.total_waves) {
                chunk_word_off[G.n_chunks] = ex + mine;
                st->total_words = ex + mine;
                if (G.host_words) *G.host_words = ex + mine;
                if (ex + mine > out_cap) atomicOr(&st->err, kErrCapacity);
            }
        }
        const bool room = pos + 1u + n <= out_cap;  // (otherwise the last waveform raises kErrCapacity)
        if (room && lane == 0) {
            out[pos] = n;                               // :379
            if (r.idx == 0) out[pos - 1] = r.n_samples;  // chunk header, :415
        }
        return room;
    };
    // copies waveform A out (its ticket begins `ex` words into the stream) and clears its part of the ring
    auto copy_out = [&](uint64_t ex) {
        ES_TRACE(gA, 3);
        const WaveRef rA = locate(G, gA);
        uint64_t pos;
        const bool room = place_header(rA, gA, nA, ex + offA, pos);
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
    // Ticket TA's place, if a wavefront of this workgroup has seen it already (they all need the same word).
    auto place_in_lds = [&](uint64_t &ex) -> bool {
        const uint32_t t = __hip_atomic_load((lds_u32 *)&s_place_t[TA & 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)t) != TA + 1u) return false;
        ex = __hip_atomic_load((lds_u64 *)&s_place_v[TA & 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return true;
    };
    auto place_to_lds = [&](uint64_t ex) {
        if (lane == 0) {  // (value, then tag: LDS operations of one wavefront are performed in order)
            __hip_atomic_store((lds_u64 *)&s_place_v[TA & 1u], ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store((lds_u32 *)&s_place_t[TA & 1u], TA + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // the same, waiting for it (bounded; a wait that expires reports kErrInternal and returns a place nothing is written to)
    auto wait_place = [&]() -> uint64_t {
        if (kAblate && (G.dbg & 128u)) return (uint64_t)TA * WV * 2048ull;  // ablation: no waiting at all (positions are wrong)
        uint32_t spins = 0;
#ifdef DRX_ENC_STAMPS
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (;;) {
            uint64_t ex;
            bool have = place_in_lds(ex);
            if (!have) {
                uint64_t v = 0;
                if (lane == 0) v = es_load(place + TA);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
                if (hi >> 31) {
                    ex = ((uint64_t)(hi & 0x7fffffffu) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
                    place_to_lds(ex);
                    have = true;
                }
            }
            if (have) {
#ifdef DRX_ENC_STAMPS
                ES_COUNT(0, __builtin_amdgcn_s_memrealtime() - t0);
                ES_COUNT(1, 1);
#endif
                return ex;
            }
            __builtin_amdgcn_s_sleep(DRX_ES_POLL_SLEEP);
            ES_COUNT(7, 1);
            if (++spins > (1u << 22)) {
                if (lane == 0) atomicOr(&st->err, kErrInternal);
                return out_cap;  // (no room for any waveform there: nothing is stored)
            }
        }
    };

    // A waveform that outgrew the ring (incompressible data, long waveforms) is coded a second time, tile by tile through the
    // (empty) ring's first words, straight to its place: a second read of its samples.
    auto stream_out = [&](uint64_t ex) {
        ES_COUNT(4, 1);
        ES_TRACE(gA, 3);
        const WaveRef r = locate(G, gA);
        const int16_t *x = in + r.sample_off;
        const uint32_t wlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.len);
        uint64_t pos;
        const bool room = place_header(r, gA, nA, ex + offA, pos);
        for (int i = lane; i < (int)(kEsFront + RING + kEsBack) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
        wave_sync();
        pend = false;
        if (!room) return;
        uint32_t *__restrict__ outp = out + pos + 1;
        uint32_t *buf = ring;
        uint64_t P = 0;
        uint32_t carry = 0, carry2 = 0;
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
    };

    // ---- which waveforms: a ticket per workgroup and WV waveforms, taken by its wavefronts TOGETHER ----
    // A waveform's place needs every EARLIER waveform's size, so the order of the indices has to be the order in which the work
    // is started, to within the slack a ring gives (~10 us).  Two forms that broke this were measured (profiles/r04_notes.md
    // section 1): wavefronts taking the next index of their workgroup's ticket whenever they were free drift apart, a
    // ticket's last waveform is then begun a whole waveform's time after its first, and every place arrives that much later;
    // an index per wavefront from eight global counters keeps the order but puts a 3 us returning atomic in front of every
    // waveform's first load (data returns in order) and arrives in bursts.  So: a rendezvous of the workgroup's wavefronts per
    // waveform.  Whoever arrives first draws the ticket -- its round trip runs while the others finish -- and the barrier
    // costs what the slowest of WV costs.
    uint32_t start = 0;  // first ring word of the waveform being coded (a multiple of four)
    uint32_t Tprev = 0;
    for (uint32_t cyc = 0;; ++cyc) {
#ifdef DRX_ENC_STAMPS
        const uint64_t es_tr = __builtin_amdgcn_s_memrealtime();
#endif
        uint32_t arr = 0;
        if (lane == 0) arr = __hip_atomic_fetch_add((lds_u32 *)&s_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        arr = (uint32_t)__builtin_amdgcn_readfirstlane((int)arr);
        if (arr == (uint32_t)WV * cyc) {
            if (lane == 0) s_ticket[cyc & 3u] = atomicAdd(ctrl + 32, 1u);
        }
        __syncthreads();
        ES_COUNT(6, __builtin_amdgcn_s_memrealtime() - es_tr);
        const uint32_t T = s_ticket[cyc & 3u];
        // the ticket before: its total to the scanner, and where in it this wavefront's waveform lies
        if (cyc) {
            uint64_t sum = 0, before = 0;
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                const uint64_t m = s_mine[(cyc - 1u) & 1u][i];
                sum += m;
                before += i < wv ? m : 0ull;
            }
            offA = before;
            if (threadIdx.x == 0) {
                es_store(size + Tprev, kEsFlag | sum);
                ES_TRACE((uint64_t)Tprev * WV, 0);
            }
        }
        // a waveform that did not fit is streamed before the next one takes the ring
        if (pend && !fitsA) stream_out(wait_place());
        if ((uint64_t)T * WV >= G.total_waves) break;
        Tprev = T;
        const uint64_t g = (uint64_t)T * WV + (uint32_t)wv;
        if (g >= G.total_waves) {  // (the batch's last ticket: this wavefront only keeps the rendezvous)
            if (lane == 0) s_mine[cyc & 1u][wv] = 0;
            if (pend) copy_out(wait_place());  // (offA is this waveform's only until the next rendezvous)
            continue;
        }
        ES_TRACE(g, 1);

        WaveRef r = locate(G, g);
        const int16_t *x = in + r.sample_off;
        const uint32_t wlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.len);

        // words this waveform may take: up to the waveform in front of it in the ring, or the whole ring
        auto gap = [&]() -> uint32_t {
            if (!pend) return RING - 8u;
            const uint32_t d = startA >= start ? startA - start : startA + RING - start;
            return d > 8u ? d - 8u : 0u;
        };
        if (pend && gap() < 512u) copy_out(wait_place());  // (short waveforms behind a long one: no room to start)
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
            // Waveform A leaves the ring here if a sibling has seen the ticket's place meanwhile (a word in LDS) -- or if this
            // tile would run into it: the one point at which a wavefront may have to WAIT for a place.
            if (pend) {
                uint64_t ex;
                const bool seen = place_in_lds(ex);
                const bool must = fits && ((P + tile_bits + 31u) >> 5) >= (uint64_t)limit;
                if (seen || must) {
                    if (seen) ES_COUNT(2, 1);
#if DRX_ES_PRIO
                    // A wavefront that has to wait for its place is AHEAD of the stream's frontier; one that finds the place of
                    // its last waveform almost as soon as it looks is what the others are waiting for.  Issue priority on
                    // the SIMD follows (4.96-5.02 -> 4.79-4.83 ms on the headline; more levels or other thresholds: the same).
                    if (!seen) __builtin_amdgcn_s_setprio(0);
                    else if (P < (uint64_t)(4u * kTile * 7u)) __builtin_amdgcn_s_setprio(2);
                    else __builtin_amdgcn_s_setprio(1);
#endif
                    copy_out(seen ? ex : wait_place());
                    limit = gap();
                }
            }
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
        // between tile groups: is waveform A's place known by now?  The load travels while a group is coded; whoever of the
        // workgroup sees the place first leaves it in LDS for the others (they look there at every tile).
        uint64_t pollv = 0;
        bool polling = false;
        auto between = [&]() {
            if (!pend) return;
            if (kAblate && (G.dbg & 128u)) { place_to_lds(wait_place()); return; }
            if (polling) {
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pollv >> 32));
                polling = false;
                if (hi >> 31) {  // (the next tile finds it in LDS, like the siblings' next tiles)
                    place_to_lds(((uint64_t)(hi & 0x7fffffffu) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pollv));
                    return;
                }
            }
            if (lane == 0) pollv = es_load(place + TA);
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
                between();
#pragma unroll
                for (int u = 0; u < kDepth; ++u) {
                    const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                    process_tile(w, 8, std::true_type{});
                    q[u] = xv[64 * (size_t)(t + u + kDepth)];
                }
            }
#pragma unroll 1
            for (; t < n_full; t += kDepth) {
                between();
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
            s_mine[cyc & 1u][wv] = 1ull + n + (r.idx == 0 ? 1ull : 0ull);
        }
        ES_COUNT(3, 1);
        ES_TRACE(g, 2);
        if (pend) copy_out(wait_place());  // (waveforms too short for the look-ups between tile groups)
        pend = true;
        fitsA = fits;
        gA = g;
        TA = T;
        startA = start;
        nA = n;
        if (fits) {
            start += (n + 3u) & ~3u;
            start = start >= RING ? start - RING : start;
        }
    }
    if (pend) copy_out(wait_place());
#ifdef DRX_ENC_STAMPS
    ES_COUNT(5, __builtin_amdgcn_s_memrealtime() - es_t0);
    __syncthreads();
    if (threadIdx.x < 8) atomicAdd(prof + threadIdx.x, (unsigned long long)s_prof[threadIdx.x]);
#endif
}

// ---------------------------------------------------------------------------
// LONG waveforms: the same machine over SEGMENTS of a waveform
// ---------------------------------------------------------------------------
// A waveform of WaveformLength >> 7000 (nEDM 81 920, NOPTREX 500 000, the reference's default of one waveform per chunk) was
// k_encode_pieces' (drx_pieces.hip): one-shot workgroups of eight wavefronts with ten tiles each -- prologue, ten tiles,
// barrier, look-back, shifted copy-out, exit: 63 % of all wave-cycles in an s_waitcnt, 0.31-0.37 of the roofline where
// k_encode_stream reaches 0.52.  Here a wavefront's unit is a SEGMENT of seg_len samples of one waveform (es_seg_shape()),
// coded from bit 0 into its ring exactly as a short waveform is; what differs:
//   * a ticket is kEsSegWaves consecutive segments of ONE waveform (tickets never span waveforms; a waveform's last ticket may
//     have idle wavefronts), and its total is a count of BITS;
//   * the scanner turns ticket totals into TWO words per ticket: the word index of the waveform's header and the bits of the
//     waveform in front of the ticket (a segmented scan: a waveform's last ticket closes it, 1 + ceil(bits / 32) words,
//     and the chunk header in front of a chunk's first waveform);
//   * a segment leaves the ring SHIFTED to its bit position.  It writes the output words that START inside its bit range: the
//     word it begins in belongs to the segment in front, and the word it ends in is completed from the codes of the (up to) 32
//     samples that follow it, which the wavefront codes behind its segment for that purpose (at least a bit each; fewer than
//     32 only at the waveform's end, where zero padding follows) -- wavefronts exchange nothing, no output word is written
//     twice, nothing is zeroed (src/deltaRice.c:237-241: the stream's bits, MSB first, n_i = ceil(bits / 32));
//   * the waveform's header n_i and its wave_words entry are written by the wavefront of its last segment, the chunk header by
//     that of a chunk's first.
// A segment whose code outgrows the ring (noisier data than the plan expected) is coded again, tile by tile to its place.
struct EsPlace2 { uint64_t whdr, bits; };

__device__ __forceinline__ void es_scanner_segs(uint64_t total, uint32_t tpw, uint32_t waves_per_chunk, const uint64_t *__restrict__ size,
                                                uint64_t *__restrict__ place, DevStatus *st) {
    const int lane = lane_id();
    if (threadIdx.x >= 64) return;
    __builtin_amdgcn_s_setprio(3);
    uint64_t pos = 0;       // the frontier: every ticket below pos has its place
    uint64_t whdr = 1;      // header word of the waveform of ticket pos (word 0: the first chunk's header, :415)
    uint64_t bits_in = 0;   // bits of that waveform in front of ticket pos
    uint32_t idle = 0;
    const double rcp_tpw = 1.0 / (double)tpw;
    while (pos < total) {
        uint64_t v[kScPer];
        const uint64_t bs = pos;
#pragma unroll
        for (int j = 0; j < (int)kScPer; ++j) {
            const uint64_t e = bs + 64u * j + (uint32_t)lane;
            v[j] = e < total ? es_load(size + e) : 0ull;
        }
        uint64_t run = 0;
        bool open = true;
#pragma unroll
        for (int j = 0; j < (int)kScPer; ++j) {
            const uint64_t m = __ballot((v[j] >> 63) != 0);
            const uint32_t l = (m == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~m);
            run += open ? l : 0u;
            open = open && l == 64u;
        }
        const uint64_t end = bs + run;  // (entries beyond `total` are never flagged: end <= total)
        if (end <= pos) {
            if (++idle > (1u << 22)) {
                if (lane == 0) atomicOr(&st->err, kErrInternal);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        idle = 0;
#pragma unroll
        for (int j = 0; j < (int)kScPer; ++j) {
            if (bs + 64u * j < end) {  // (wave uniform)
                const uint64_t e = bs + 64u * j + (uint32_t)lane;
                const bool in = e < end;
                const uint32_t val = in ? (uint32_t)(v[j] & 0xffffffffull) : 0u;  // bits of a ticket: < 2^22
                const uint32_t e32 = (uint32_t)e;                                 // (the host keeps tickets below 2^32)
                // e / tpw: a double-precision estimate (53 bits: off by at most one), put right (a 32-bit division is ~30
                // instructions, per group)
                uint32_t g = (uint32_t)((double)e32 * rcp_tpw);
                uint32_t jj = e32 - g * tpw;
                if ((int32_t)jj < 0) { g -= 1u; jj += tpw; }
                if (jj >= tpw) { g += 1u; jj -= tpw; }
                const bool closing = in && jj + 1u == tpw;
                const uint32_t inc = wave_incl_scan_dpp(val), ex = inc - val;
                const int f = lane - (int)jj;  // lane of my waveform's first ticket (negative: in front of this group)
                const uint32_t exf = (uint32_t)__shfl((int)ex, f < 0 ? 0 : f);
                const uint64_t before = f < 0 ? bits_in + ex : (uint64_t)(ex - exf);
                const uint64_t nw = closing ? (before + val + 31ull) >> 5 : 0ull;
                const uint64_t contrib = closing ? 1ull + nw + (((g + 1u) % waves_per_chunk == 0u) ? 1ull : 0ull) : 0ull;
                uint64_t cinc;
                if (!__any(contrib >= (1ull << 24))) {
                    cinc = wave_incl_scan_dpp((uint32_t)contrib);
                } else {
                    cinc = contrib;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint64_t t = __shfl_up(cinc, d);
                        if (lane >= d) cinc += t;
                    }
                }
                const uint64_t my_whdr = whdr + cinc - contrib;
                if (in) {
                    es_store(place + 2u * e, kEsFlag | my_whdr);
                    es_store(place + 2u * e + 1u, kEsFlag | before);
                }
                // the state behind this group's last ticket
                // (v_readlane with a scalar lane select: the two words are a chain from group to group, and a 64-bit shuffle is
                // two LDS permutes of ~100 cycles each)
                const uint64_t left = end - (bs + 64u * j);
                const int li = __builtin_amdgcn_readfirstlane(left >= 64u ? 63 : (int)left - 1);
                const uint64_t wn = my_whdr + contrib, bn = closing ? 0ull : before + val;
                whdr = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wn >> 32), li) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wn, li);
                bits_in = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bn >> 32), li) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bn, li);
            }
        }
        pos = end;
    }
}

template <bool GEN, int WV, uint32_t RING>
__global__ __launch_bounds__(64 * WV, 4) void k_encode_stream_segs(Geom G, uint32_t seg_len, uint32_t tpw, const int16_t *__restrict__ in,
                                                                   uint32_t *__restrict__ out, uint64_t out_cap,
                                                                   uint64_t *__restrict__ chunk_word_off, uint32_t *__restrict__ wave_words,
                                                                   uint64_t *__restrict__ size, uint64_t *__restrict__ place,
                                                                   uint32_t *__restrict__ ctrl, DevStatus *st) {
    __shared__ __attribute__((aligned(16))) uint32_t ring_all[WV][kEsFront + RING + kEsBack];
    __shared__ uint32_t s_role, s_arrive, s_ticket[4];
    __shared__ uint64_t s_mine[2][WV];   // bits of the segments of the last two tickets
    __shared__ uint64_t s_place_v[2][2];  // place of a ticket (header word, bits in front), once one wavefront has seen it ...
    __shared__ uint32_t s_place_t[2];     // ... and which ticket (+ 1) that was
    __shared__ uint32_t s_first[2][WV];   // first 32 bits of the segments of the last two tickets (what completes the word the segment in front ends in)
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    typedef uint64_t __attribute__((address_space(3))) lds_u64;
    static_assert((kEsFront + RING + kEsBack) % 4 == 0 && RING % 4 == 0, "16-byte LDS accesses");
    static_assert(WV == (int)kEsSegWaves, "es_seg_shape() counts tickets of kEsSegWaves segments");
    constexpr uint32_t kRingBits = RING * 32u;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint32_t *row = ring_all[wv];
    uint32_t *ring = row + kEsFront;
    const uint32_t ring_bits0 = lds_addr(ring) * 8u;

    if (threadIdx.x == 0) {
        s_role = atomicAdd(ctrl, 1u);
        s_arrive = 0;
        s_place_t[0] = s_place_t[1] = 0;
    }
    for (int i = lane; i < (int)(kEsFront + RING + kEsBack) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint64_t n_tickets = G.total_waves * tpw;
    if (s_role == 0) {
        es_scanner_segs(n_tickets, tpw, G.u_n_waves, size, place, st);
        return;
    }
    const uint32_t k = G.k;
    const u16x2 tp[4] = {splat(GEN ? G.enc_t[0] : 1u), splat(GEN ? G.enc_t[1] : 0xffffu), splat(GEN ? G.enc_t[2] : 0u),
                         splat(GEN ? G.enc_t[3] : 0u)};

    // ---- segment A: coded, waiting in the ring (or, if it did not fit, not at all) for its place.  Wave uniform. ----
    bool pend = false, fitsA = true, lastA = false, sibA = false;  // sibA: the next segment is the next wavefront's, same ticket
    uint64_t gA = 0, offA = 0;  // its waveform; bits of its ticket in front of it (known behind the next rendezvous)
    uint32_t TA = 0, startA = 0, nA = 0, bitsA = 0, sbeginA = 0, sendA = 0, parA = 0;  // nA: ring words, bitsA: bits of the segment proper, parA: its cycle's parity

    // header words and bookkeeping of segment A at its place; B = the waveform-relative bit it begins at
    auto headers = [&](const WaveRef &r, const EsPlace2 &pl, uint64_t B) {
        if (lane != 0) return;
        if (sbeginA == 0u && r.idx == 0u) {
            chunk_word_off[r.chunk] = pl.whdr - 1u;
            if (pl.whdr - 1u < out_cap) out[pl.whdr - 1u] = r.n_samples;  // chunk header, :415
        }
        if (lastA) {
            const uint64_t n = (B + bitsA + 31ull) >> 5;  // n_i, :237-241
            wave_words[gA] = (uint32_t)n;
            if (pl.whdr < out_cap) out[pl.whdr] = (uint32_t)n;  // :379
            if (gA + 1u == G.total_waves) {
                const uint64_t tot = pl.whdr + 1ull + n;
                chunk_word_off[G.n_chunks] = tot;
                st->total_words = tot;
                if (G.host_words) *G.host_words = tot;
                if (tot > out_cap) atomicOr(&st->err, kErrCapacity);
            }
        }
    };
    // copies segment A out, shifted to its bit position, and clears its part of the ring
    auto copy_out = [&](const EsPlace2 &pl) {
        const WaveRef rA = locate(G, gA);
        const uint64_t B = pl.bits + offA;               // bits of the waveform in front of the segment
        headers(rA, pl, B);
        const uint64_t w0 = pl.whdr + 1ull + (B >> 5);   // output word that holds the segment's first bit
        const uint32_t s = (uint32_t)B & 31u;
        const uint32_t i_lo = s ? 1u : 0u, i_hi = (s + bitsA + 31u) >> 5;  // the segment's words: w0 + [i_lo, i_hi)
        const uint32_t nA4 = (nA + 3u) & ~3u, top = nA4 > i_hi ? nA4 : ((i_hi + 3u) & ~3u);
        if (sibA) {
            // the word this segment ends in is completed by the first bits of the next segment: the next wavefront's, same
            // ticket -- it finished that segment before the ticket's total could be published, i.e. before this place existed
            if (lane == 0) {
                const uint32_t F = s_first[parA][wv + 1], sh = bitsA & 31u;
                uint32_t wi = startA + (bitsA >> 5);
                wi = wi >= RING ? wi - RING : wi;
                ring[wi] |= F >> sh;
                if (sh) {
                    wi = wi + 1u >= RING ? wi + 1u - RING : wi + 1u;
                    ring[wi] |= F << (32u - sh);
                }
            }
            wave_sync();
        }
        uint32_t *__restrict__ outp = out + w0;
        const uint64_t room = out_cap > w0 ? out_cap - w0 : 0ull;  // words from w0 on that exist
        typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
        typedef u32x4s __attribute__((address_space(1), aligned(4))) g_u32x4_a4;
        uint32_t carryw = 0;  // the ring word in front of this round's first piece
        for (uint32_t i0 = 0; i0 < top; i0 += 256u) {
            const uint32_t i = i0 + 4u * (uint32_t)lane;
            uint4 v = make_uint4(0, 0, 0, 0);
            uint32_t w = startA + i;
            w = w >= RING ? w - RING : w;  // (starts are multiples of four words: a 16-byte piece never straddles the end)
            const bool inr = i < nA4;
            if (inr) v = *reinterpret_cast<const uint4 *>(ring + w);
            uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x138, 0xf, 0xf, false);  // wave_shr:1
            if (lane == 0) pw = carryw;
            carryw = (uint32_t)__builtin_amdgcn_readlane((int)v.w, 63);
            const uint32_t o0 = __builtin_amdgcn_alignbit(pw, v.x, s), o1 = __builtin_amdgcn_alignbit(v.x, v.y, s);
            const uint32_t o2 = __builtin_amdgcn_alignbit(v.y, v.z, s), o3 = __builtin_amdgcn_alignbit(v.z, v.w, s);
            if (i >= i_lo && i + 4u <= i_hi && (uint64_t)i + 4u <= room) {
                *(g_u32x4_a4 *)(outp + i) = (u32x4s){o0, o1, o2, o3};
            } else {
                if (i >= i_lo && i < i_hi && (uint64_t)i < room) outp[i] = o0;
                if (i + 1u >= i_lo && i + 1u < i_hi && (uint64_t)i + 1u < room) outp[i + 1u] = o1;
                if (i + 2u >= i_lo && i + 2u < i_hi && (uint64_t)i + 2u < room) outp[i + 2u] = o2;
                if (i + 3u >= i_lo && i + 3u < i_hi && (uint64_t)i + 3u < room) outp[i + 3u] = o3;
            }
            if (inr) *reinterpret_cast<uint4 *>(ring + w) = make_uint4(0, 0, 0, 0);
        }
        wave_sync();
        pend = false;
    };
    auto place_in_lds = [&](EsPlace2 &pl) -> bool {
        const uint32_t t = __hip_atomic_load((lds_u32 *)&s_place_t[TA & 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)t) != TA + 1u) return false;
        pl.whdr = __hip_atomic_load((lds_u64 *)&s_place_v[TA & 1u][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pl.bits = __hip_atomic_load((lds_u64 *)&s_place_v[TA & 1u][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return true;
    };
    auto place_to_lds = [&](const EsPlace2 &pl) {
        if (lane == 0) {  // (values, then tag: LDS operations of one wavefront are performed in order)
            __hip_atomic_store((lds_u64 *)&s_place_v[TA & 1u][0], pl.whdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store((lds_u64 *)&s_place_v[TA & 1u][1], pl.bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store((lds_u32 *)&s_place_t[TA & 1u], TA + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // lanes 0 and 1 hold the two words of a place (one load instruction): both flagged -> the place
    auto place_of = [&](uint64_t pv, EsPlace2 &pl) -> bool {
        const uint32_t lo = (uint32_t)pv, hi = (uint32_t)(pv >> 32);
        const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)hi, 0), h1 = (uint32_t)__builtin_amdgcn_readlane((int)hi, 1);
        if (!((h0 & h1) >> 31)) return false;
        pl.whdr = ((uint64_t)(h0 & 0x7fffffffu) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, 0);
        pl.bits = ((uint64_t)(h1 & 0x7fffffffu) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, 1);
        return true;
    };
    auto wait_place = [&]() -> EsPlace2 {
        if (kAblate && (G.dbg & 128u)) return EsPlace2{(uint64_t)TA * WV * 2048ull, 0ull};  // ablation: no waiting at all (positions are wrong)
        uint32_t spins = 0;
        for (;;) {
            EsPlace2 pl;
            if (place_in_lds(pl)) return pl;
            uint64_t pv = 0;
            if (lane < 2) pv = es_load(place + 2ull * TA + (uint32_t)lane);
            if (place_of(pv, pl)) {
                place_to_lds(pl);
                return pl;
            }
            __builtin_amdgcn_s_sleep(DRX_ES_POLL_SLEEP);
            if (++spins > (1u << 22)) {
                if (lane == 0) atomicOr(&st->err, kErrInternal);
                pl.whdr = out_cap;  // (no room there: nothing is stored)
                pl.bits = 0;
                return pl;
            }
        }
    };

    // A segment that outgrew the ring is coded a second time, tile by tile through the (empty) ring's first words, straight to
    // its place: from the bit it starts at inside its first output word, so that staged words ARE output words.
    auto stream_out = [&](const EsPlace2 &pl) {
        const WaveRef r = locate(G, gA);
        const uint64_t B = pl.bits + offA;
        headers(r, pl, B);
        for (int i = lane; i < (int)(kEsFront + RING + kEsBack) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
        wave_sync();
        pend = false;
        const uint64_t wbase = pl.whdr + 1ull + (B >> 5);
        const uint32_t P0 = (uint32_t)B & 31u;
        const uint32_t limit = (P0 + bitsA + 31u) >> 5, skip = P0 ? 1u : 0u;  // words [skip, limit) from wbase are mine
        const uint32_t more = r.len - sendA < 32u ? r.len - sendA : 32u;
        const uint32_t len = sendA - sbeginA + more;
        const int16_t *x = in + r.sample_off + sbeginA;
        uint32_t *buf = ring;
        uint64_t P = P0;
        uint32_t carry = 0, carry2 = 0;
        if (sbeginA) {
            carry = (uint32_t)(uint16_t)x[-1] << 16;
            if (GEN) {
                carry |= (uint32_t)(uint16_t)x[-2];
                carry2 = (uint32_t)(uint16_t)x[-4] | ((uint32_t)(uint16_t)x[-3] << 16);
            }
        }
        uint32_t wn[4];
        int nvn = load8_dwords(x, len, 0u, lane, true, wn);
        for (uint32_t t0 = 0; t0 < len; t0 += kTile) {
            uint32_t w[4] = {wn[0], wn[1], wn[2], wn[3]};
            const int nv = nvn;
            if (t0 + kTile < len) nvn = load8_dwords(x, len, t0 + kTile, lane, true, wn);  // (travels while this tile is coded)
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
            const uint64_t wfirst = P >> 5;  // first staged word
            emit_tile<false>(cc, ring_bits0 + (uint32_t)(P & 31u) + incl - lane_bits);
            P += tile_bits;
            wave_sync();
            const uint32_t nfull = (uint32_t)((P >> 5) - wfirst);
            for (uint32_t i = lane; i < nfull; i += 64) {
                const uint64_t idx = wfirst + i;
                if (idx >= skip && idx < limit && wbase + idx < out_cap) out[wbase + idx] = buf[i];
                buf[i] = 0;
            }
            wave_sync();
            if (nfull && lane == 0) { const uint32_t cwd = buf[nfull]; buf[nfull] = 0; buf[0] = cwd; }
            wave_sync();
        }
        if (lane == 0) {
            const uint64_t idx = P >> 5;
            if ((P & 31u) && idx >= skip && idx < limit && wbase + idx < out_cap) out[wbase + idx] = buf[0];
            buf[0] = 0;
        }
        wave_sync();
    };

    uint32_t start = 0;  // first ring word of the segment being coded (a multiple of four)
    uint32_t Tprev = 0;
    for (uint32_t cyc = 0;; ++cyc) {
        uint32_t arr = 0;
        if (lane == 0) arr = __hip_atomic_fetch_add((lds_u32 *)&s_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        arr = (uint32_t)__builtin_amdgcn_readfirstlane((int)arr);
        if (arr == (uint32_t)WV * cyc) {
            if (lane == 0) s_ticket[cyc & 3u] = atomicAdd(ctrl + 32, 1u);
        }
        __syncthreads();
        const uint32_t T = s_ticket[cyc & 3u];
        // the ticket before: its total to the scanner, and the bits of it in front of this wavefront's segment
        if (cyc) {
            uint64_t sum = 0, before = 0;
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                const uint64_t m = s_mine[(cyc - 1u) & 1u][i];
                sum += m;
                before += i < wv ? m : 0ull;
            }
            offA = before;
            if (threadIdx.x == 0) es_store(size + Tprev, kEsFlag | sum);
        }
        if (pend && !fitsA) stream_out(wait_place());
        if ((uint64_t)T >= n_tickets) break;
        Tprev = T;
        const uint32_t gw = T / tpw, tj = T - gw * tpw;
        const uint64_t g = gw;
        WaveRef r = locate(G, g);
        const uint32_t wlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.len);
        const uint32_t sgi = tj * (uint32_t)WV + (uint32_t)wv;
        const uint64_t sb64 = (uint64_t)sgi * seg_len;
        if (sb64 >= wlen) {  // (a waveform's last ticket: no segment for this wavefront, it only keeps the rendezvous)
            if (lane == 0) s_mine[cyc & 1u][wv] = 0;
            if (pend) copy_out(wait_place());  // (offA is segment A's only until the next rendezvous)
            continue;
        }
        const uint32_t s_begin = (uint32_t)sb64;
        const uint32_t s_end = wlen - s_begin > seg_len ? s_begin + seg_len : wlen;
        // what completes the word the segment ends in: the next wavefront's first bits (LDS), or, for a ticket's last segment,
        // the codes of the up to 32 samples behind it, coded here
        // (a sibling's segment of 32 samples or more has 32 bits or more; shorter ones -- tiny waveforms, a waveform's last
        // few samples -- may not complete the word)
        const bool sib = wv + 1 < WV && seg_len >= 32u && wlen - s_end >= 32u;
        const uint32_t more = sib ? 0u : (wlen - s_end < 32u ? wlen - s_end : 32u);
        const uint32_t slen = s_end - s_begin;       // samples of the segment proper (a multiple of 8 unless it is the last)
        const uint32_t clen = slen + more;           // samples coded
        const int16_t *x = in + r.sample_off + s_begin;

        auto gap = [&]() -> uint32_t {
            if (!pend) return RING - 8u;
            const uint32_t d = startA >= start ? startA - start : startA + RING - start;
            return d > 12u ? d - 12u : 0u;  // (a segment may be given two words behind its code: copy_out's completion)
        };
        if (pend && gap() < 512u) copy_out(wait_place());
        uint32_t limit = gap();

        uint64_t P = 0;        // bits so far (wave uniform)
        uint32_t seg_bits = 0;
        bool first = true;     // no tile coded yet
        bool fits = true;
        uint32_t carry = 0, carry2 = 0;
        if (s_begin) {  // the samples in front of the segment (a waveform's first sample has none: x[-1] := 0, :53-54)
            carry = (uint32_t)(uint16_t)x[-1] << 16;
            if (GEN) {
                carry |= (uint32_t)(uint16_t)x[-2];
                carry2 = (uint32_t)(uint16_t)x[-4] | ((uint32_t)(uint16_t)x[-3] << 16);
            }
        }
        auto wrap = [&](uint32_t bits) -> uint32_t { return bits >= kRingBits ? bits - kRingBits : bits; };
        // mark: the lane (wave uniform, -1: none) whose first sample is the first one BEHIND the segment proper
        auto process_tile = [&](const uint32_t (&w)[4], int nv, int mark, auto full_tag) {
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
            if (FULLT) concat_codes(cc, cw);
            const uint32_t incl = wave_incl_scan_dpp(lane_bits);
            const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (!FULLT && mark >= 0) seg_bits = (uint32_t)P + (mark ? (uint32_t)__builtin_amdgcn_readlane((int)incl, mark - 1) : 0u);
            if (pend) {
                EsPlace2 pl;
                const bool seen = place_in_lds(pl);
                const bool must = fits && ((P + tile_bits + 31u) >> 5) >= (uint64_t)limit;
                if (seen || must) {
#if DRX_ES_PRIO
                    if (!seen) __builtin_amdgcn_s_setprio(0);
                    else if (P < (uint64_t)(4u * kTile * 7u)) __builtin_amdgcn_s_setprio(2);
                    else __builtin_amdgcn_s_setprio(1);
#endif
                    copy_out(seen ? pl : wait_place());
                    limit = gap();
                }
            }
            if (fits && ((P + tile_bits + 31u) >> 5) < (uint64_t)limit) {
                const uint32_t s0 = start * 32u + (uint32_t)P;
                if (FULLT && !__any(lane_bits > 128u))
                    place_words(cw, ring_bits0 + wrap(s0 + incl));
                else
                    emit_tile<FULLT>(cc, ring_bits0 + wrap(s0 + incl - lane_bits));
                if (s0 < kRingBits && s0 + tile_bits >= kRingBits) {
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
            if (first) {  // (the first tile always fits: at least 500 words are free, a tile is at most 400)
                first = false;
                wave_sync();
                if (lane == 0) s_first[cyc & 1u][wv] = ring[start];
            }
        };
        // full tiles in front of the one that holds the segment's end take the fast loop
        const uint32_t n_fast = more ? slen / kTile : clen / kTile;
        uint64_t pollv = 0;
        bool polling = false;
        auto between = [&]() {
            if (!pend) return;
            if (kAblate && (G.dbg & 128u)) { place_to_lds(wait_place()); return; }
            if (polling) {
                polling = false;
                EsPlace2 pl;
                if (place_of(pollv, pl)) {
                    place_to_lds(pl);
                    return;
                }
            }
            pollv = 0;
            if (lane < 2) pollv = es_load(place + 2ull * TA + (uint32_t)lane);
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
                if ((uint32_t)u < n_fast) q[u] = xv[64 * (size_t)u];
            }
#pragma unroll 1
            for (; t + 2u * kDepth <= n_fast; t += kDepth) {
                between();
#pragma unroll
                for (int u = 0; u < kDepth; ++u) {
                    const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                    process_tile(w, 8, -1, std::true_type{});
                    q[u] = xv[64 * (size_t)(t + u + kDepth)];
                }
            }
#pragma unroll 1
            for (; t < n_fast; t += kDepth) {
                between();
#pragma unroll
                for (int u = 0; u < kDepth; ++u) {
                    if (t + (uint32_t)u < n_fast) {
                        const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                        process_tile(w, 8, -1, std::true_type{});
                        if (t + (uint32_t)u + kDepth < n_fast) q[u] = xv[64 * (size_t)(t + u + kDepth)];
                    }
                }
            }
            // the tile(s) with the segment's end and the samples coded behind it
            for (uint32_t t0 = n_fast * kTile; t0 < clen; t0 += kTile) {
                uint32_t w[4];
                const int nv = load8_dwords(x, clen, t0, lane, true, w);
                const int mark = (more && slen >= t0 && slen < t0 + (uint32_t)kTile) ? (int)((slen - t0) >> 3) : -1;
                process_tile(w, nv, mark, std::false_type{});
            }
        }
        if (!more) seg_bits = (uint32_t)P;  // (nothing coded behind the segment)
        const uint32_t n = (uint32_t)((P + (sib ? 63u : 31u)) >> 5);  // ring words (sib: + the 32 bits copy_out puts behind the code)
        wave_sync();
        if (lane == 0) s_mine[cyc & 1u][wv] = seg_bits;
        if (pend) copy_out(wait_place());
        pend = true;
        fitsA = fits;
        gA = g;
        TA = T;
        startA = start;
        nA = fits ? n : 0u;
        bitsA = seg_bits;
        sbeginA = s_begin;
        sendA = s_end;
        lastA = s_end == wlen;
        sibA = sib;
        parA = cyc & 1u;
        if (fits) {
            start += (n + 3u) & ~3u;
            start = start >= RING ? start - RING : start;
        }
    }
    if (pend) copy_out(wait_place());
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
// d_scan: uint64[2 * tickets + kEsCtrlWords + 16] (tickets <= total_waves): total[tickets] | place[tickets] | control, zeroed
// here on the stream before every launch | the diagnostic build's counters (left alone).
hipError_t launch_encode_stream(const Geom &G, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                                uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint64_t *d_scan,
                                DevStatus *d_status, hipEvent_t *ev, hipStream_t s) {
    if (G.total_waves == 0) return hipSuccess;
    const uint64_t W = G.total_waves;
    const uint64_t tickets = (W + kEsWaves - 1) / kEsWaves;
    mark(ev, 0, s);
    hipError_t e = hipMemsetAsync(d_scan, 0, (2 * tickets + kEsCtrlWords) * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    mark(ev, 1, s);
    mark(ev, 2, s);
    uint64_t *size = d_scan, *place = d_scan + tickets;
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(d_scan + 2 * tickets);
    unsigned long long *prof = reinterpret_cast<unsigned long long *>(d_scan + 2 * tickets + kEsCtrlWords);
#ifdef DRX_ENC_STAMPS
    static uint64_t *d_trace = nullptr;
    const char *trace_path = getenv("DRX_ES_TRACE");
    if (trace_path && !d_trace) {
        (void)hipMalloc((void **)&d_trace, 5 * W * sizeof(uint64_t));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_es_trace), &d_trace, sizeof d_trace);
    }
    if (d_trace) (void)hipMemsetAsync(d_trace, 0, 5 * W * sizeof(uint64_t), s);
#endif
    // persistent: 16 wavefronts per CU, one workgroup of them the scanner's; never more than the batch can feed
    // (debug flag 262144: three workgroups -- a scanner and two coders -- so that a small test batch takes every wavefront
    // through many waveforms, i.e. around its ring)
    const unsigned full = 256u * (16u / kEsWaves);
    const unsigned grid = (G.dbg & 262144u) ? 3u : (unsigned)(tickets + 1 < full ? tickets + 1 : full);
    if (G.n_taps)
        k_encode_stream<true, kEsWaves, kEsRing><<<grid, 64 * kEsWaves, 0, s>>>(G, d_in, d_out, out_cap, d_chunk_word_off, d_wave_words, size,
                                                                              place, ctrl, d_status, prof);
    else
        k_encode_stream<false, kEsWaves, kEsRing><<<grid, 64 * kEsWaves, 0, s>>>(G, d_in, d_out, out_cap, d_chunk_word_off, d_wave_words, size,
                                                                               place, ctrl, d_status, prof);
    mark(ev, 3, s);
#ifdef DRX_ENC_STAMPS
    {
        unsigned long long h[12];
        (void)hipStreamSynchronize(s);
        if (d_trace && trace_path) {
            uint64_t *ht = (uint64_t *)malloc(5 * W * sizeof(uint64_t));
            (void)hipMemcpy(ht, d_trace, 5 * W * sizeof(uint64_t), hipMemcpyDeviceToHost);
            FILE *f = fopen(trace_path, "wb");
            if (f) { fwrite(ht, sizeof(uint64_t), 5 * W, f); fclose(f); }
            free(ht);
        }
        (void)hipMemcpy(h, prof, sizeof h, hipMemcpyDeviceToHost);
        (void)hipMemset(prof, 0, sizeof h);
        const double wf = h[3] ? (double)h[3] : 1.0;
        fprintf(stderr, "enc stream stamps: %llu waveforms: %llu placed between tile groups without waiting, %llu waits (%.2f us each), %llu streamed; "
                "per waveform: %.2f us in all, %.2f us waiting for the place, %.2f us at the rendezvous\n",
                h[3], h[2], h[1], h[1] ? h[0] / (double)h[1] / 100.0 : 0.0, h[4], h[5] / wf / 100.0, h[0] / wf / 100.0, h[6] / wf / 100.0);
        fprintf(stderr, "enc stream scanner: %llu rounds in %.1f us (%.2f us each), %llu found nothing new, %llu a full window; "
                "%.1f failed polls per wait of a coder\n",
                h[8], h[11] / 100.0, h[8] ? h[11] / 100.0 / h[8] : 0.0, h[9], h[10], h[1] ? h[7] / (double)h[1] : 0.0);
    }
#endif
    return hipGetLastError();
}

// d_scan: uint64[3 * tickets + kEsCtrlWords]: total[tickets] | place[2 * tickets] | control, zeroed here before every launch.
// The caller has checked the geometry (uniform, tickets below 2^32 - 2^16) and sized d_scan for the smallest segments.
hipError_t launch_encode_stream_segs(const Geom &G, uint32_t seg_target, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap,
                                     uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint64_t *d_scan,
                                     DevStatus *d_status, hipEvent_t *ev, hipStream_t s) {
    if (G.total_waves == 0) return hipSuccess;
    const EsSegShape sh = es_seg_shape(G.u_wave_len, seg_target);
    const uint64_t tickets = G.total_waves * sh.tpw;
    mark(ev, 0, s);
    hipError_t e = hipMemsetAsync(d_scan, 0, (3 * tickets + kEsCtrlWords) * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    mark(ev, 1, s);
    mark(ev, 2, s);
    uint64_t *size = d_scan, *place = d_scan + tickets;
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(d_scan + 3 * tickets);
    const unsigned full = 256u * (16u / kEsWaves);
    const unsigned grid = (G.dbg & 262144u) ? 3u : (unsigned)(tickets + 1 < full ? tickets + 1 : full);
    if (G.n_taps)
        k_encode_stream_segs<true, kEsWaves, kEsRing><<<grid, 64 * kEsWaves, 0, s>>>(G, sh.seg_len, sh.tpw, d_in, d_out, out_cap, d_chunk_word_off,
                                                                                   d_wave_words, size, place, ctrl, d_status);
    else
        k_encode_stream_segs<false, kEsWaves, kEsRing><<<grid, 64 * kEsWaves, 0, s>>>(G, sh.seg_len, sh.tpw, d_in, d_out, out_cap, d_chunk_word_off,
                                                                                    d_wave_words, size, place, ctrl, d_status);
    mark(ev, 3, s);
    return hipGetLastError();
}

}  // namespace drx
