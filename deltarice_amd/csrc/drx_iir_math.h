// drx_iir_math.h -- 3 x 3 matrices and vectors over Z / 2^16 for the inverse of a general prediction filter (drx_iir.hip: a pass of
// its own over decoded residuals; drx_blocks.hip: inside the block decoder).  Device code; included by the .hip units.
#ifndef DRX_IIR_MATH_H
#define DRX_IIR_MATH_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace drx {

struct V3 { uint32_t x, y, z; };
struct M3 { uint32_t m[9]; };
__device__ __forceinline__ M3 load_m3(const uint32_t *__restrict__ t) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = t[i];
    return r;
}
__device__ __forceinline__ V3 mul(const M3 &a, const V3 &v) {
    V3 r;
    r.x = __umul24(a.m[0], v.x) + __umul24(a.m[1], v.y) + __umul24(a.m[2], v.z);
    r.y = __umul24(a.m[3], v.x) + __umul24(a.m[4], v.y) + __umul24(a.m[5], v.z);
    r.z = __umul24(a.m[6], v.x) + __umul24(a.m[7], v.y) + __umul24(a.m[8], v.z);
    return r;
}
__device__ __forceinline__ M3 mul(const M3 &a, const M3 &b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = (__umul24(a.m[3 * i], b.m[j]) + __umul24(a.m[3 * i + 1], b.m[3 + j]) + __umul24(a.m[3 * i + 2], b.m[6 + j])) & 0xffffu;
    return r;
}
__device__ __forceinline__ V3 add(const V3 &a, const V3 &b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 lo16(const V3 &a) { return V3{a.x & 0xffffu, a.y & 0xffffu, a.z & 0xffffu}; }
__device__ __forceinline__ V3 shfl_up_v3(const V3 &a, int d) {
    return V3{(uint32_t)__shfl_up((int)a.x, d), (uint32_t)__shfl_up((int)a.y, d), (uint32_t)__shfl_up((int)a.z, d)};
}

// Tables of a filter for runs of M samples per lane (host): PL[8][9] = A^(M 2^d), d = 0..7 | PLANE[64][9] = A^(M l) | c1 c2 c3 sgn
constexpr uint32_t kRunTabPL = 0, kRunTabPLANE = 8 * 9, kRunTabC = kRunTabPLANE + 64 * 9, kRunTabWords = kRunTabC + 4;
inline void iir_run_tables(const uint32_t fast_nt[3], uint32_t t0neg, uint32_t M, uint32_t *tab) {
    typedef uint32_t Mat[9];
    auto mmul = [](const uint32_t *a, const uint32_t *b, uint32_t *r) {
        Mat t;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) t[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j]) & 0xffffu;
        for (int i = 0; i < 9; ++i) r[i] = t[i];
    };
    const uint32_t sg = t0neg ? 0xffffu : 1u;
    const uint32_t c[3] = {(fast_nt[0] * sg) & 0xffffu, (fast_nt[1] * sg) & 0xffffu, (fast_nt[2] * sg) & 0xffffu};
    Mat A = {c[0], c[1], c[2], 1, 0, 0, 0, 1, 0}, I = {1, 0, 0, 0, 1, 0, 0, 0, 1}, P, cur;
    for (int i = 0; i < 9; ++i) P[i] = I[i];
    for (uint32_t i = 0; i < M; ++i) mmul(A, P, P);  // A^M: one lane's run
    for (int i = 0; i < 9; ++i) cur[i] = P[i];
    for (int d = 0; d < 8; ++d) {
        for (int i = 0; i < 9; ++i) tab[kRunTabPL + 9 * d + i] = cur[i];
        mmul(cur, cur, cur);
    }
    for (int i = 0; i < 9; ++i) cur[i] = I[i];
    for (int l = 0; l < 64; ++l) {
        for (int i = 0; i < 9; ++i) tab[kRunTabPLANE + 9 * l + i] = cur[i];
        mmul(P, cur, cur);
    }
    tab[kRunTabC] = c[0]; tab[kRunTabC + 1] = c[1]; tab[kRunTabC + 2] = c[2]; tab[kRunTabC + 3] = sg;
}

}  // namespace drx
#endif
