// Packed 16-bit Rice coding of a 512-sample tile (64 lanes x 8 samples) and its emission into an LDS bit buffer: shared by
// the single-pass encoders (k_encode_fused in drx_encode_kernels.hip, k_encode_pieces in drx_pieces.hip) and the segment
// encoder.  Device code only; included by the .hip translation units.
#ifndef DRX_ENCODE_H
#define DRX_ENCODE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "drx_device.h"

namespace drx {

constexpr int kTile = 512;  // samples per wave tile: 64 lanes x 8 samples (16 B per lane)

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ i16x2 as_i16x2(uint32_t x) { return __builtin_bit_cast(i16x2, x); }
__device__ __forceinline__ uint32_t as_u32(u16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ u16x2 splat(uint32_t v) { return (u16x2){(uint16_t)v, (uint16_t)v}; }

// Packed code parameters of the 8 samples a lane holds as 4 dwords (low half = earlier sample).
//   nb   code length,
//   c16  the code's low 16 bits: payload with the terminating '1' above it (k low bits of z | 1 << k), or the
//        16 payload bits of an escape, whose terminator is bit 16:
//   e    1 for an escape, else 0.  The code word is (e << 16) | c16, its leading zeros are implicit.
struct PackedCodes { uint32_t nb[4], c16[4], e[4]; };

// x[j]: samples 2j, 2j+1; xprev: dword whose HIGH half is the sample just before x[0]'s low half.
// GEN: general forward filter (src/deltaRice.c:64-74) d[i] = sum_j taps[j] x[i-j] modulo 2^16, at most four taps
// (tp[j] = taps[j] in both halves); xprev2: the dword before xprev (samples i-4, i-3 of the lane's first pair).
template <bool GEN>
__device__ __forceinline__ void packed_codes(const uint32_t x[4], uint32_t xprev, uint32_t xprev2, const u16x2 (&tp)[4],
                                             uint32_t k, PackedCodes &c) {
    const u16x2 kv = splat(k), kp1 = splat(k + 1u), c16k = splat(16u - k);
    const u16x2 mlo = splat((1u << k) - 1u), mdelta = splat(0xffffu - ((1u << k) - 1u));
    // stage by stage over the four dwords rather than dword by dword: consecutive instructions are then
    // independent and the packed-math / op_sel hazards need no s_nop (21 per tile before)
    u16x2 z[4], qc[4], e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t xm1 = j ? x[j - 1] : xprev;                                           // samples 2j-2, 2j-1
        const uint32_t before = __builtin_amdgcn_alignbit(x[j], xm1, 16);                    // samples 2j-1, 2j
        i16x2 d;
        if (GEN) {
            const uint32_t xm2 = j >= 2 ? x[j - 2] : (j == 1 ? xprev : xprev2);
            const uint32_t before3 = __builtin_amdgcn_alignbit(xm1, xm2, 16);                // samples 2j-3, 2j-2
            d = __builtin_bit_cast(i16x2, (u16x2)(as_u16x2(x[j]) * tp[0] + as_u16x2(before) * tp[1] +
                                                  as_u16x2(xm1) * tp[2] + as_u16x2(before3) * tp[3]));
        } else {
            d = as_i16x2(x[j]) - as_i16x2(before);                                           // :51-63, mod 2^16
        }
        z[j] = __builtin_bit_cast(u16x2, (i16x2)(d << (int16_t)1)) ^
               __builtin_bit_cast(u16x2, (i16x2)(d >> (int16_t)15));                         // zig-zag :207-211
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) qc[j] = __builtin_elementwise_min((u16x2)(z[j] >> kv), splat(8u));  // min(q, 8)
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = qc[j] >> (uint16_t)3;                                         // 1 = escape (:215)
#pragma unroll
    for (int j = 0; j < 4; ++j) c.nb[j] = as_u32(e[j] * c16k + (qc[j] + kp1));   // q+1+k, or 8+1+16
    // c16 = z & (M-1) | M, or z for an escape: one v_bfi_b32 per dword, mask = M-1 or 0xffff per half (bit k of the
    // word 1 << k lies outside M-1, and an escape's mask lets nothing of it through)
    const uint32_t mword = as_u32(splat(1u << k));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t mask = as_u32(e[j] * mdelta + mlo);
        c.c16[j] = (mask & as_u32(z[j])) | (~mask & mword);
        c.e[j] = as_u32(e[j]);
    }
}

#ifndef DRX_ENC_CAP_WORDS
#define DRX_ENC_CAP_WORDS 2048
#endif
constexpr uint32_t kEncCapWords = DRX_ENC_CAP_WORDS;  // LDS words per waveform buffer (8 KB): 9.3 bits/sample at L = 7000

// Loads this lane's 8 samples of the tile as 4 dwords; returns the number that exist.
__device__ __forceinline__ int load8_dwords(const int16_t *__restrict__ x, uint32_t len, uint32_t t0, int lane,
                                            bool vec_ok, uint32_t w[4]) {
    const uint32_t i0 = t0 + 8u * (uint32_t)lane;
    const int nv = (i0 >= len) ? 0 : (int)((len - i0) < 8u ? (len - i0) : 8u);
    if (vec_ok && nv == 8) {
        const uint4 q = *reinterpret_cast<const uint4 *>(x + i0);
        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a = (2 * j < nv) ? (uint32_t)(uint16_t)x[i0 + 2 * j] : 0u;
            const uint32_t b = (2 * j + 1 < nv) ? (uint32_t)(uint16_t)x[i0 + 2 * j + 1] : 0u;
            w[j] = a | (b << 16);
        }
    }
    return nv;
}

// Zeroes the code lengths of the samples a lane does not have (trailing partial tile): a
// zero-length code contributes no bits and, in emit_tile<false>, no set bits either.
__device__ __forceinline__ void mask_tail(PackedCodes &c, int nv) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t m = (2 * j + 1 < nv) ? 0xffffffffu : ((2 * j < nv) ? 0x0000ffffu : 0u);
        c.nb[j] &= m;
    }
}

// Bits of this lane's 8 codes.
__device__ __forceinline__ uint32_t lane_tile_bits(const PackedCodes &c) {
    const u16x2 s = as_u16x2(c.nb[0]) + as_u16x2(c.nb[1]) + as_u16x2(c.nb[2]) + as_u16x2(c.nb[3]);
    const uint32_t v = as_u32(s);
    return (v & 0xffffu) + (v >> 16);
}

// ORs this lane's codes into LDS.  pb = 8 * (LDS byte address of the buffer's word 0) + bit position
// of the lane's first code, so (pb >> 3) & ~3 is the LDS byte address of the word holding that bit.
// FULL = false: lengths may have been zeroed by mask_tail(); such codes must not set any bit.
template <bool FULL>
__device__ __forceinline__ void emit_tile(const PackedCodes &c, uint32_t pb) {
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t n = (j & 1) ? (c.nb[j >> 1] >> 16) : (c.nb[j >> 1] & 0xffffu);
        // terminator + payload; the leading zeros are implicit
        uint32_t code32 = __builtin_amdgcn_perm(c.e[j >> 1], c.c16[j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
        if (!FULL) code32 = n ? code32 : 0u;
        const uint32_t pe = pb + n;        // end of this code = start of the next
        // left-align the code at bit (pb & 31) of a 64-bit window: shift = 64 - (pb & 31) - n,
        // which is ((pb & 32) - pe) mod 64; v_lshlrev_b64 reads 6 bits of the shift
        const uint64_t v = (uint64_t)code32 << (((pb & 32u) - pe) & 63u);
        lds_u32 *w = (lds_u32 *)(uintptr_t)((pb >> 3) & ~3u);
        __hip_atomic_fetch_or(w, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)v) __hip_atomic_fetch_or(w + 1, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pb = pe;
    }
}

// The same for a full tile, with fewer LDS operations: the lane first concatenates its 8 codes in
// registers (a right-aligned 128-bit string w3:w2:w1:w0, one funnel shift per word and code), then ORs
// whole words.  pe = 8 * (LDS byte address of the buffer's word 0) + bit position of the END of the lane's
// last code; lane_bits <= 128 (the caller checks; 8 codes are 52 bits on the headline data and can
// only pass 128 with three escapes or more).  Every code is at least one bit long (full tile), so
// 32 - n is a valid funnel shift.  Words before the lane's first one receive an OR with zero: the
// buffers carry a 4-word pad in front for that.
__device__ __forceinline__ void concat_codes(const PackedCodes &c, uint32_t (&w)[4]) {
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // v_alignbit_b32 / v_lshl_or_b32 read 5 bits of the shift: for the low half the packed register serves as is
        const uint32_t nbj = c.nb[j >> 1];
        const uint32_t n = (j & 1) ? (nbj >> 16) : nbj;
        const uint32_t code32 = __builtin_amdgcn_perm(c.e[j >> 1], c.c16[j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
        const uint32_t s = 0u - n;  // == 32 - n (mod 32)
        if (j >= 3) w3 = __builtin_amdgcn_alignbit(w3, w2, s);
        if (j >= 2) w2 = __builtin_amdgcn_alignbit(w2, w1, s);
        if (j >= 1) w1 = __builtin_amdgcn_alignbit(w1, w0, s);
        w0 = (w0 << (n & 31u)) | code32;
    }
    w[0] = w0; w[1] = w1; w[2] = w2; w[3] = w3;
}

__device__ __forceinline__ void place_words(const uint32_t (&wd)[4], uint32_t pe) {
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    // B << (32 - e) == (B << 32) >> e with e = pe & 31: x0 is the word that holds bit pe
    const uint32_t x0 = __builtin_amdgcn_alignbit(wd[0], 0u, pe);
    const uint32_t x1 = __builtin_amdgcn_alignbit(wd[1], wd[0], pe);
    const uint32_t x2 = __builtin_amdgcn_alignbit(wd[2], wd[1], pe);
    const uint32_t x3 = __builtin_amdgcn_alignbit(wd[3], wd[2], pe);
    const uint32_t x4 = __builtin_amdgcn_alignbit(0u, wd[3], pe);
    lds_u32 *w = (lds_u32 *)(uintptr_t)(((pe >> 3) & ~3u) - 16u);  // word of x4: positive DS offsets from here
    __hip_atomic_fetch_or(w + 4, x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(w + 3, x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(w + 2, x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (x3 | x4) {
        __hip_atomic_fetch_or(w + 1, x3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (x4) __hip_atomic_fetch_or(w, x4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

}  // namespace drx
#endif
