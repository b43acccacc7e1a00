// drx_device.h -- device-side helpers shared by the kernel translation units (not installed).
#ifndef DRX_DEVICE_H
#define DRX_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "drx_internal.h"

namespace drx {

// Ablation switches inside the hot loops (Geom::dbg bits 1, 2, 4, 16, 32, 64, 128) exist only in builds made with
// -DDRX_ABLATION (loaded through DRX_LIB_PATH for A/B timing); the shipped kernels carry none of those branches.
#ifdef DRX_ABLATION
constexpr bool kAblate = true;
#else
constexpr bool kAblate = false;
#endif

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

// Bounds on a waveform's payload word count n_i: a code has between 1 + k (q = 0, src/deltaRice.c:215-222) and 25 bits
// (escape, :223-228).  Every walker rejects a header outside [min, max]: the chain of a valid stream never leaves them.
__host__ __device__ __forceinline__ uint32_t max_payload_words(uint32_t len) { return (uint32_t)(((uint64_t)len * 25u + 31u) >> 5); }
__host__ __device__ __forceinline__ uint32_t min_payload_words(uint32_t len, uint32_t k) {
    return (uint32_t)(((uint64_t)len * (k + 1u) + 31u) >> 5);
}

// count-leading-zeros with the ISA's result for 0 (-1) instead of the source language's undefined behaviour: the
// decoders meet an all-zero window only past the end of a corrupt stream, where any value will do, but it has to BE a value
#ifndef DRX_FFBH_MODE
#define DRX_FFBH_MODE 0
#endif
__device__ __forceinline__ uint32_t ffbh(uint32_t x) {
#if DRX_FFBH_MODE == 0
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#elif DRX_FFBH_MODE == 1
    return (uint32_t)__builtin_clz(x | 1u);
#else
    return (uint32_t)__builtin_clz(x);  // (A/B only: undefined for 0 in the source language)
#endif
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

// inclusive prefix sum over the 64 lanes (DPP: row shifts, then row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

// decoupled look-back entries: 2 status bits on top of the value (the word is its own flag; zero = not yet)
constexpr uint64_t kScanAgg = 1ull << 62, kScanPrefix = 2ull << 62, kScanValMask = (1ull << 62) - 1ull;

__device__ __forceinline__ uint32_t lds_addr(const uint32_t *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)p;
}

}  // namespace drx
#endif
