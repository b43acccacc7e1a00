// drx_walk.h -- the header-chain walk: device functions and kernels.  Included by drx_decode_kernels.hip alone (the decode
// launch runs walkers inside k_decode_lanes and launches the walk kernels).
//
// The format's only way to waveform i + 1 is the length header of waveform i (src/deltaRice.c:320-325).  Four forms:
//   scalar chains     one lane (eight scalar-load chains per wave) per chunk, hop by hop: inside the decode launch of large
//                     batches, where the serial latency hides behind the decoding;
//   LDS block walker  chunks of short waveforms streamed through LDS by a whole wave;
//   chunk-wide walk   k_pw_scan + k_walk_parallel: candidate headers, binary lifting -- a handful of chunks of long waveforms;
//   block-parallel    k_bw_blocks / k_bw_scan / k_bw_emit: every B-word block of a chunk of short waveforms holds a header.
// Every walker validates while it walks (sample count, n_i bounds, the chain ends at the chunk's end).
#ifndef DRX_WALK_H
#define DRX_WALK_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "drx_internal.h"
#include "drx_device.h"

namespace drx {


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

// ---------------------------------------------------------------------------
// The chunk-wide walk WITHOUT reading the chunk (round 4): 64 chains per chunk, chased in parallel
// ---------------------------------------------------------------------------
// k_pw_scan reads every word of a chunk to find the ~2000 that are headers -- a second pass over the whole stream, 0.26 ms in
// front of a 1.2 ms decode at 100 chunks, 0.48 at 224.  A chain needs none of that: a header says where the next one is.  What
// a chain needs is a true header to START from, and those are easy to come by: a payload word practically never looks like
// a length header (n_i lies in a range of a few thousand out of 2^32), so the first plausible word at or behind ANY position
// is the next header.  The chunk is cut at 64 word positions; from each, one wavefront looks forward for the first plausible
// word (on average half a waveform's code away); then 64 lanes chase their chains at once, each until it arrives at the next
// lane's start -- which it must hit exactly: with lane 0 starting at the chunk's first header and the last chain ending at
// the chunk's end, the chain of equalities proves every start a true header, as in the block-parallel walk.  W / 64 dependent
// loads per lane instead of W, a few hundred KB read instead of the chunk.  Anything else (an impostor picked as a start, a
// broken chain, a count that is not W) flags the chunk for the scalar walker, which also judges it.
constexpr int kSwThreads = 1024;     // (at most: chunks of few waveforms get 256, the kernel strides by blockDim)
constexpr uint32_t kSwSegs = 64;    // chains per chunk
constexpr uint32_t kSwCap = 192;    // headers a chain may collect: chains are equal in WORDS, so one through quiet waveforms holds more than
                                    // the average W / 64 <= 56 -- up to 3.4 x (W = 2000: 6 x) before the scalar walker has to take the chunk; 48 KB of LDS

__global__ __launch_bounds__(kSwThreads) void k_walk_sparse(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                            const uint64_t *__restrict__ chunk_word_off,
                                                            uint64_t *__restrict__ wave_off, uint32_t *__restrict__ wave_words,
                                                            uint32_t *__restrict__ fail, const uint32_t *__restrict__ list) {
    __shared__ uint32_t s_a[kSwSegs + 1];          // where chain s starts (word of the chunk); s_a[S] = the chunk's length
    __shared__ uint32_t s_cnt[kSwSegs], s_base[kSwSegs + 1];
    __shared__ uint32_t s_list[kSwSegs][kSwCap];   // position of every header a chain found (n_i = the distance to the next one - 1)
    __shared__ uint32_t s_bad;
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id();
    const uint32_t wv = tid >> 6;
    const uint64_t c = list ? (uint64_t)list[blockIdx.x] : blockIdx.x;
    uint32_t W, L, N;
    uint64_t base;
    if (G.uniform) { W = G.u_n_waves; L = G.u_wave_len; N = G.u_n_samples; base = c * W; }
    else { const ChunkDesc d = G.chunks[c]; W = d.n_waves; L = d.wave_len; N = d.n_samples; base = d.wave_base; }
    const uint64_t begin = chunk_word_off[c];
    const uint64_t end = chunk_word_off[c + 1];
    if (tid == 0) s_bad = 0;
    bool ok = !(end > in_words || begin + 2 > end || end - begin > 0x7fffffffull);
    if (ok && in[begin] != N) ok = false;
    if (!ok) { if (tid == 0) fail[c] = 1u; return; }  // (uniform across the workgroup)
    const uint32_t len_w = (uint32_t)(end - begin);
    const uint32_t last_len = N - (W - 1u) * L;
    const uint32_t max_full = max_payload_words(L), min_full = min_payload_words(L, G.k);
    const uint32_t max_last = max_payload_words(last_len), min_last = min_payload_words(last_len, G.k);
    const uint32_t lo_any = min_full < min_last ? min_full : min_last, hi_any = max_full > max_last ? max_full : max_last;
    // chains: as many as give each a few waveforms -- and only where finding a start (half a waveform's code, read by one wavefront
    // 512 words at a time) costs less than the hops it saves: chunks of 32 x 500 000 samples are one chain of 32 hops
    uint32_t S = W / 8u;
    S = S > kSwSegs ? kSwSegs : (S < 1u ? 1u : S);
    if (len_w / W > 8192u) S = 1u;
    const uint32_t *cw = in + begin;  // the chunk's words
    // ---- 1. a start for every chain: the first plausible header at or behind its cut ----
    if (tid == 0) { s_a[0] = 1u; s_a[S] = len_w; }
    __syncthreads();  // (s_bad is cleared before any wavefront may raise it)
    auto plausible = [&](uint32_t v, uint32_t i) { return v >= lo_any && v <= hi_any && (uint64_t)i + 1u + v <= len_w; };
    for (uint32_t sg = 1u + wv; sg < S; sg += blockDim.x >> 6) {
        uint32_t from = 1u + (uint32_t)(((uint64_t)(len_w - 1u) * sg) / S);
        // (a waveform's code is at most hi_any words: a stream without a header in twice that is corrupt, and is not read to its end
        // by every cut)
        const uint32_t stop = (uint64_t)from + 2u * (hi_any + 2u) < len_w ? from + 2u * (hi_any + 2u) : len_w;
        uint32_t found = len_w;  // (none: the chains in front run to the chunk's end)
        for (uint32_t tries = 0; tries < 64u; ++tries) {
            constexpr uint32_t U = 8;
            found = len_w;
            for (uint32_t j0 = from; j0 < stop && found == len_w; j0 += 64u * U) {
                uint32_t v[U];
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) {
                    const uint32_t i = j0 + 64u * u + (uint32_t)lane;
                    v[u] = i < len_w ? cw[i] : 0xffffffffu;
                }
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) {
                    const uint64_t m = __ballot(plausible(v[u], j0 + 64u * u + (uint32_t)lane));
                    if (m && found == len_w) found = j0 + 64u * u + (uint32_t)__builtin_ctzll(m);
                }
            }
            if (found >= stop) {  // (nothing in front of the stop)
                if (stop < len_w && lane == 0) atomicOr(&s_bad, 1u);
                found = len_w;
                break;
            }
            // a payload word is plausible once in a million, and this kernel looks at millions: a start counts only if the
            // word it points to is plausible too (or the chunk's end): one dependent load per cut
            const uint32_t nxt = found + 1u + cw[found];  // (<= len_w: plausible())
            if (nxt == len_w || plausible(cw[nxt], nxt)) break;
            from = found + 1u;
            found = len_w;
        }
        if (lane == 0) s_a[sg] = found;
    }
    __syncthreads();
    // ---- 2. the chases ----
    if (tid < S) {
        uint32_t pos = s_a[tid];
        const uint32_t target = s_a[tid + 1u];
        uint32_t cnt = 0;
        bool bad = false;
        while (pos < target) {
            const uint32_t n = cw[pos];  // (pos < len_w: inside the chunk, and the chunk inside the stream)
            if (n < lo_any || n > hi_any || cnt >= kSwCap) { bad = true; break; }
            s_list[tid][cnt] = pos;
            ++cnt;
            pos += n + 1u;  // (<= len_w + hi_any: no overflow, chunks have fewer than 2^31 words)
        }
        if (pos != target) bad = true;  // (a chain must arrive exactly where the next one started)
        s_cnt[tid] = cnt;
        if (bad) atomicOr(&s_bad, 1u);
    }
    __syncthreads();
    // ---- 3. waveform numbers, the tables ----
    if (tid < 64u) {
        const uint32_t cnt = tid < S ? s_cnt[tid] : 0u;
        const uint32_t incl = wave_incl_scan_dpp(cnt);
        s_base[tid] = incl - cnt;
        if (tid == 63u) s_base[64] = incl;
    }
    __syncthreads();
    if (s_bad || s_base[64] != W) { if (tid == 0) fail[c] = 1u; return; }
    bool bad = false;
    for (uint32_t sg = wv; sg < S; sg += blockDim.x >> 6) {
        const uint32_t cnt = s_cnt[sg], b0 = s_base[sg];
        for (uint32_t i = (uint32_t)lane; i < cnt; i += 64u) {
            const uint32_t at = s_list[sg][i];
            const uint32_t n = (i + 1u < cnt ? s_list[sg][i + 1u] : s_a[sg + 1u]) - at - 1u;  // (the chain arrived at the next one's start)
            const uint32_t w = b0 + i;
            if (n > ((w + 1u == W) ? max_last : max_full) || n < ((w + 1u == W) ? min_last : min_full)) bad = true;
            wave_off[base + w] = begin + at;
            wave_words[base + w] = n;
        }
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
constexpr uint32_t kBwCandCap = 512;  // the fast path's room for a block's headers (more: the chase)
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
    __shared__ uint16_t cand[EMIT ? 2 : kBwCandCap];  // the fast path's candidates, in position order
    __shared__ uint32_t pre[kBwMaxList + 1];
    const int lane = lane_id();
    bw_block_prefix<B>(chunk_word_off, in_words, list, n_list, blocks_max, pre, lane);
    const uint32_t total = pre[n_list];
    // A block at a time per wavefront, the NEXT block's words in flight (registers) while the current one is worked on in LDS:
    // without that a unit was a chain of dependent latencies -- chunk table, 8 KB of loads, the LDS work, the stores --
    // of ~3.5 us, and 84 units per wavefront were the kernel's 0.3 ms on config 5 whatever the chase cost (round 4).
    struct Unit {
        bool ok;
        uint64_t c, begin, wbase;
        uint32_t b, b0, len_w, blk_len, W, L, n_samples;
    };
    uint32_t cached_slot = 0xffffffffu;
    Unit cached{};
    auto locate_unit = [&](uint32_t unit) __attribute__((always_inline)) {
        uint32_t lo = 0, hi = n_list;  // invariant: pre[lo] <= unit < pre[hi]
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pre[mid] <= unit) lo = mid; else hi = mid;
        }
        const uint32_t slot = lo;
        if (slot != cached_slot) {  // (a wavefront's consecutive units mostly lie in one chunk: its table entry is read once)
            cached_slot = slot;
            const uint64_t cc = list ? (uint64_t)list[slot] : slot;
            cached.c = cc;
            if (G.uniform) { cached.W = G.u_n_waves; cached.L = G.u_wave_len; cached.n_samples = G.u_n_samples; cached.wbase = cc * cached.W; }
            else { const ChunkDesc d = G.chunks[cc]; cached.W = d.n_waves; cached.L = d.wave_len; cached.n_samples = d.n_samples; cached.wbase = d.wave_base; }
            cached.begin = chunk_word_off[cc];
            cached.len_w = (uint32_t)(chunk_word_off[cc + 1] - cached.begin);  // (a chunk with an unusable extent has no blocks)
            cached.ok = !(EMIT && fail[cc]);
        }
        Unit u = cached;
        u.b = unit - pre[slot];
        u.b0 = u.b * B;  // block = words [b0, b0 + B) of the chunk
        if (u.b0 >= u.len_w) u.ok = false;  // (only a chunk cut at the host's bound; flagged by k_bw_scan)
        u.blk_len = u.ok ? (u.len_w - u.b0 < B ? u.len_w - u.b0 : B) : 0u;
        return u;
    };
    // the block's words: unconditional 16-byte loads (any 4-byte alignment: a chunk starts anywhere) from an address clamped
    // into the stream; what the clamp moved and what lies behind the block is sorted out when the registers go to LDS
    const int64_t a_max = (int64_t)in_words - 4;
    auto fetch = [&](const Unit &u, uint4 (&v)[NV]) __attribute__((always_inline)) {
        const int64_t a0 = (int64_t)(u.begin + u.b0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t a = a0 + (int64_t)((uint32_t)(j * 64 + lane) * 4u);
            const int64_t ac = a > a_max ? (a_max < 0 ? 0 : a_max) : a;
            v[j] = *reinterpret_cast<const uint4 *>(in + ac);
        }
    };
    auto store = [&](const Unit &u, const uint4 (&v)[NV]) __attribute__((always_inline)) {
        const uint64_t a0 = u.begin + u.b0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const uint32_t i = (uint32_t)(j * 64 + lane) * 4u;
            uint4 w = v[j];
            if ((int64_t)(a0 + i) > a_max) {  // the clamp moved this piece (the last words of the batch): word by word
                auto ld = [&](uint64_t q) { return q < in_words ? in[q] : 0xffffffffu; };
                w = make_uint4(ld(a0 + i), ld(a0 + i + 1u), ld(a0 + i + 2u), ld(a0 + i + 3u));
            }
            w.x = (i + 0u < u.blk_len) ? w.x : 0xffffffffu;
            w.y = (i + 1u < u.blk_len) ? w.y : 0xffffffffu;
            w.z = (i + 2u < u.blk_len) ? w.z : 0xffffffffu;
            w.w = (i + 3u < u.blk_len) ? w.w : 0xffffffffu;
            *reinterpret_cast<uint4 *>(blk + i) = w;
        }
    };
    uint4 img[NV];
    Unit cur{};
    cur.ok = false;
    if (blockIdx.x < total) { cur = locate_unit(blockIdx.x); if (cur.ok) fetch(cur, img); }
    for (uint32_t unit = blockIdx.x; unit < total; unit += gridDim.x) {
        wave_sync();  // (the previous block's LDS reads are done)
        const Unit me_u = cur;
        if (me_u.ok) store(me_u, img);
        if (unit + gridDim.x < total) { cur = locate_unit(unit + gridDim.x); if (cur.ok) fetch(cur, img); }
        if (!me_u.ok) continue;
        const uint64_t c = me_u.c, begin = me_u.begin, wbase = me_u.wbase;
        const uint32_t b = me_u.b, b0 = me_u.b0, len_w = me_u.len_w, blk_len = me_u.blk_len, W = me_u.W, L = me_u.L, n_samples = me_u.n_samples;
        const uint32_t max_full = (uint32_t)(((uint64_t)L * 25u + 31u) >> 5);
        const uint32_t min_words = min_payload_words(L, G.k);
        (void)c; (void)wbase; (void)W; (void)n_samples; (void)begin; (void)min_words;
        wave_sync();
        if (!EMIT) {
            // FAST PATH (round 4).  A payload word is 32 bits of dense code: it lies in [1, max_words] (400 for WaveformLength
            // 512) about once in ten million words, so the block's words in that range ARE its headers, nearly always.  All of
            // them are found at once (the block is read 16 bytes per lane and step, as below), put in position order by ballots,
            // and held to the chain's own equalities in parallel: every candidate's position + n + 1 must be the next
            // candidate's position, the last one's must leave the block.  One impostor (or a header beyond the list's room)
            // and the block takes the chase below, as before.  The chase is one lane following ~20 dependent LDS reads per
            // 2048-word block while 63 lanes wait: 0.13 us per hop, 0.33 ms for config 5's 278 000 blocks (profiles/r04_notes.md section 5).
            {
                const uint32_t from0 = b == 0 ? 1u : 0u;
                uint32_t ncand = 0;
                bool overflow = false;
                for (uint32_t base = 0; base < blk_len; base += 256u) {
                    const uint32_t i = base + 4u * (uint32_t)lane;
                    const uint4 v = *reinterpret_cast<const uint4 *>(blk + i);  // (all B words were written above)
                    const bool f0 = i + 0u >= from0 && i + 0u < blk_len && v.x - 1u < max_full;
                    const bool f1 = i + 1u >= from0 && i + 1u < blk_len && v.y - 1u < max_full;
                    const bool f2 = i + 2u >= from0 && i + 2u < blk_len && v.z - 1u < max_full;
                    const bool f3 = i + 3u >= from0 && i + 3u < blk_len && v.w - 1u < max_full;
                    const uint64_t m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2), m3 = __ballot(f3);
                    if ((m0 | m1 | m2 | m3) == 0ull) continue;
                    const uint64_t below = (1ull << lane) - 1ull;
                    // candidates of lower lanes come first, then this lane's own in component order
                    uint32_t r = ncand + (uint32_t)(__builtin_popcountll(m0 & below) + __builtin_popcountll(m1 & below) +
                                                    __builtin_popcountll(m2 & below) + __builtin_popcountll(m3 & below));
                    if (f0) { if (r < kBwCandCap) cand[r] = (uint16_t)(i + 0u); ++r; }
                    if (f1) { if (r < kBwCandCap) cand[r] = (uint16_t)(i + 1u); ++r; }
                    if (f2) { if (r < kBwCandCap) cand[r] = (uint16_t)(i + 2u); ++r; }
                    if (f3) { if (r < kBwCandCap) cand[r] = (uint16_t)(i + 3u); ++r; }
                    ncand += (uint32_t)(__builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
                    overflow = overflow || ncand > kBwCandCap;
                }
                wave_sync();
                bool fast = ncand != 0u && !overflow && (!LIST || ncand <= hop_cap);
                uint32_t exit_rel = 0;
                if (fast) {
                    bool bad = false;
                    for (uint32_t j0 = 0; j0 < ncand; j0 += 64u) {
                        const uint32_t j = j0 + (uint32_t)lane;
                        if (j < ncand) {
                            const uint32_t pos = cand[j], n = blk[pos], nxt = pos + n + 1u;
                            // (as in the chase: the waveform must end inside the chunk; LIST: not below 1 + k bits per sample
                            // unless it is the chunk's last)
                            if ((uint64_t)b0 + nxt > len_w) bad = true;
                            if (LIST && n < min_words && b0 + nxt != len_w) bad = true;
                            if (j + 1u < ncand) { if (nxt != (uint32_t)cand[j + 1u]) bad = true; }
                            else { if (nxt < blk_len) bad = true; exit_rel = nxt; }
                        }
                    }
                    fast = !__any(bad);
                }
                if (fast) {
                    exit_rel = (uint32_t)__builtin_amdgcn_readlane((int)exit_rel, (int)((ncand - 1u) & 63u));
                    if (LIST) {
                        for (uint32_t j = (uint32_t)lane; j < ncand; j += 64u) {
                            const uint32_t pos = cand[j];
                            hops[(uint64_t)unit * hop_cap + j] = pos | (blk[pos] << 12);
                        }
                    }
                    if (lane == 0) {
                        BwBlock o;
                        o.entry = b0 + (uint32_t)cand[0];
                        o.count = ncand;
                        o.exit = b0 + exit_rel;
                        o.base = 0;
                        info[unit] = o;
                    }
                    continue;
                }
            }
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
    uint32_t carry_exit = 1u;  // where the block in front of this group of 64 left (the first header follows the chunk's)
    // (four groups' records in flight: one wavefront per chunk, 88 groups for config 5's chunks -- a dependent 16-byte load per
    // group was the kernel's whole time, 0.065 ms)
    constexpr uint32_t UG = 4;
    for (uint32_t g0 = 0; g0 < n_blocks && !bad; g0 += 64u * UG) {
        BwBlock og[UG];
#pragma unroll
        for (uint32_t u = 0; u < UG; ++u) {
            const uint32_t b = g0 + 64u * u + (uint32_t)lane;
            og[u] = BwBlock{0xffffffffu, 0, 0, 0};
            if (b < n_blocks) og[u] = my[b];
        }
#pragma unroll
        for (uint32_t u = 0; u < UG; ++u) {
            const uint32_t b0 = g0 + 64u * u;
            if (b0 >= n_blocks || bad) break;  // (wave uniform)
            const uint32_t b = b0 + (uint32_t)lane;
            BwBlock o = og[u];
            // every block must have been entered, start where its predecessor left, and the ends must be the chunk's
            uint32_t prev_exit = (uint32_t)__shfl_up((int)o.exit, 1);
            if (lane == 0) prev_exit = carry_exit;
            carry_exit = (uint32_t)__builtin_amdgcn_readlane((int)o.exit, 63);
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
    // A LANE per block (round 4; a wavefront per block before): a block holds ~20 headers, so 44 lanes of a wavefront had
    // nothing to do, and every block paid its chain of dependent loads (list, table entry, info) alone -- 0.12-0.14 ms on
    // config 5 for 11 MB of table.  64 blocks' chains now travel together.
    for (uint32_t unit = blockIdx.x * 256u + threadIdx.x; unit < total; unit += gridDim.x * 256u) {
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
        const uint32_t *hp = hops + (uint64_t)unit * hop_cap;
        for (uint32_t i = 0; i < me.count; ++i) {
            const uint32_t h = hp[i], pos = h & 0xfffu, n = h >> 12, wi = me.base + i;
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

}  // namespace drx
#endif
