// Issue-rate micro-benchmark for the VALU instructions the decode kernel is made of (gfx950).
// For each instruction: a dependent chain and four independent chains, at 1, 2 and 4 waves per SIMD;
// prints SIMD cycles per wave64 instruction (wall time x measured clock / instructions per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kUnroll = 64;   // instructions per chain per loop trip
constexpr int kTrips = 1000;

#define DEP4(OP)  OP(a, a, b, c) OP(a, a, b, c) OP(a, a, b, c) OP(a, a, b, c)
#define IND4(OP)  OP(a, a, b, c) OP(d, d, b, c) OP(e, e, b, c) OP(f, f, b, c)

#define DEFINE_KERNEL(NAME, ASM3)                                                                       \
    __global__ __launch_bounds__(64) void dep_##NAME(uint32_t *out, uint32_t seed, uint64_t *cyc) {     \
        uint32_t a = threadIdx.x + seed, b = seed * 3u + 1u, c = (seed & 7u) + 3u;                      \
        const uint64_t t0 = clock64();                                                                  \
        for (int t = 0; t < kTrips; ++t) {                                                              \
            _Pragma("unroll") for (int i = 0; i < kUnroll; ++i) asm volatile(ASM3 : "+v"(a) : "v"(b), "v"(c)); \
        }                                                                                               \
        const uint64_t t1 = clock64();                                                                  \
        out[blockIdx.x * 64 + threadIdx.x] = a;                                                         \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                \
    }                                                                                                   \
    __global__ __launch_bounds__(64) void ind_##NAME(uint32_t *out, uint32_t seed, uint64_t *cyc) {     \
        uint32_t a = threadIdx.x + seed, d = a + 1, e = a + 2, f = a + 3, b = seed * 3u + 1u, c = (seed & 7u) + 3u; \
        const uint64_t t0 = clock64();                                                                  \
        for (int t = 0; t < kTrips; ++t) {                                                              \
            _Pragma("unroll") for (int i = 0; i < kUnroll / 4; ++i) {                                   \
                asm volatile(ASM3 : "+v"(a) : "v"(b), "v"(c));                                          \
                asm volatile(ASM3 : "+v"(d) : "v"(b), "v"(c));                                          \
                asm volatile(ASM3 : "+v"(e) : "v"(b), "v"(c));                                          \
                asm volatile(ASM3 : "+v"(f) : "v"(b), "v"(c));                                          \
            }                                                                                           \
        }                                                                                               \
        const uint64_t t1 = clock64();                                                                  \
        out[blockIdx.x * 64 + threadIdx.x] = a + d + e + f;                                             \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                \
    }

DEFINE_KERNEL(add, "v_add_u32 %0, %0, %1")
DEFINE_KERNEL(alignbit, "v_alignbit_b32 %0, %0, %1, %2")
DEFINE_KERNEL(bfe, "v_bfe_u32 %0, %0, %2, %1")
DEFINE_KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, %2, %1")
DEFINE_KERNEL(xad, "v_xad_u32 %0, %0, %1, %2")
DEFINE_KERNEL(add3, "v_add3_u32 %0, %0, %1, %2")
DEFINE_KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL(ffbh, "v_ffbh_u32 %0, %0")
DEFINE_KERNEL(not_, "v_not_b32 %0, %0")
DEFINE_KERNEL(lshr, "v_lshrrev_b32 %0, 1, %0")
DEFINE_KERNEL(bfe_i, "v_bfe_i32 %0, %0, 0, 1")
DEFINE_KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(cmp, "v_cmp_lt_u32 vcc, %0, %1")
DEFINE_KERNEL(alignbit_k, "v_alignbit_b32 %0, %0, %1, 7")
DEFINE_KERNEL(bfe_k, "v_bfe_u32 %0, %0, 3, 5")

DEFINE_KERNEL(cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
DEFINE_KERNEL(cmp_cnd, "v_cmp_lt_u32 vcc, %2, %0\n v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(cmp_s_cnd, "v_cmp_lt_u32_e64 s[10:11], %2, %0\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
DEFINE_KERNEL(max_, "v_max_u32 %0, %0, %1")
DEFINE_KERNEL(min_, "v_min_u32 %0, %0, %1")
DEFINE_KERNEL(and_, "v_and_b32 %0, %0, %1")
DEFINE_KERNEL(sub_, "v_sub_u32 %0, %0, %1")
DEFINE_KERNEL(lshl_v, "v_lshlrev_b32 %0, %2, %0")
DEFINE_KERNEL(mul24, "v_mul_u32_u24 %0, %0, %1")
DEFINE_KERNEL(mad24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFINE_KERNEL(med3, "v_med3_u32 %0, %0, %1, %2")
DEFINE_KERNEL(perm, "v_perm_b32 %0, %0, %1, %2")
DEFINE_KERNEL(or_sdwa, "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0")
DEFINE_KERNEL(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEFINE_KERNEL(sad, "v_sad_u32 %0, %0, %1, %2")
DEFINE_KERNEL(pk_add, "v_pk_add_u16 %0, %0, %1")
DEFINE_KERNEL(lshlrev_k, "v_lshlrev_b32 %0, 3, %0")
DEFINE_KERNEL(add_lshl, "v_add_lshl_u32 %0, %0, %1, %2")
// round 3: which other single-operation instructions are in the fast class
DEFINE_KERNEL(or_, "v_or_b32 %0, %0, %1")
DEFINE_KERNEL(xor_, "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL(ashr, "v_ashrrev_i32 %0, 1, %0")
DEFINE_KERNEL(mov, "v_mov_b32 %0, %1")
DEFINE_KERNEL(subrev, "v_subrev_u32 %0, %0, %1")
DEFINE_KERNEL(xnor, "v_xnor_b32 %0, %0, %1")
DEFINE_KERNEL(lshr_v, "v_lshrrev_b32 %0, %2, %0")
DEFINE_KERNEL(add_lit, "v_add_u32 %0, 0x12345, %0")
DEFINE_KERNEL(and_e64, "v_and_b32_e64 %0, %0, %1")
DEFINE_KERNEL(add_co, "v_add_co_u32 %0, vcc, %0, %1")
DEFINE_KERNEL(lshl_or, "v_lshl_or_b32 %0, %0, %2, %1")
DEFINE_KERNEL(bfi, "v_bfi_b32 %0, %1, %0, %2")
DEFINE_KERNEL(bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEFINE_KERNEL(mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEFINE_KERNEL(pk_lshr, "v_pk_lshrrev_b16 %0, %2, %0")
DEFINE_KERNEL(add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:DWORD")
DEFINE_KERNEL(min_i, "v_min_i32 %0, %0, %1")
DEFINE_KERNEL(cvt, "v_cvt_f32_u32 %0, %0")
DEFINE_KERNEL(addf, "v_add_f32 %0, %0, %1")

typedef void (*kern_t)(uint32_t *, uint32_t, uint64_t *);
struct Case { const char *name; kern_t dep, ind; };
#define CASE(NAME) {#NAME, dep_##NAME, ind_##NAME}

int main() {
    const Case cases[] = {CASE(add), CASE(alignbit), CASE(alignbit_k), CASE(bfe), CASE(bfe_k), CASE(lshl_add), CASE(xad), CASE(add3),
                          CASE(and_or), CASE(ffbh), CASE(not_), CASE(lshr), CASE(bfe_i), CASE(cndmask), CASE(cmp), CASE(cndmask_s), CASE(cmp_cnd), CASE(cmp_s_cnd), CASE(max_), CASE(min_), CASE(and_), CASE(sub_),
                          CASE(lshl_v), CASE(lshlrev_k), CASE(mul24), CASE(mad24), CASE(med3), CASE(perm), CASE(or_sdwa), CASE(mov_dpp), CASE(addc),
                          CASE(sad), CASE(pk_add), CASE(add_lshl),
                          CASE(or_), CASE(xor_), CASE(ashr), CASE(mov), CASE(subrev), CASE(xnor), CASE(lshr_v), CASE(add_lit), CASE(and_e64),
                          CASE(add_co), CASE(lshl_or), CASE(bfi), CASE(bitop3), CASE(mul_lo), CASE(pk_lshr), CASE(add_sdwa), CASE(min_i),
                          CASE(cvt), CASE(addf)};
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    uint64_t *cyc;
    const int max_blocks = cus * 4 * 4;
    CK(hipMalloc(&out, (size_t)max_blocks * 64 * 4));
    CK(hipMalloc(&cyc, (size_t)max_blocks * 8));
    std::vector<uint64_t> h(max_blocks);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d CUs; cycles per wave64 instruction and wave (clock64 of the wave, mean over waves): dependent chain | 4 independent chains\n", cus);
    printf("%-12s %30s %30s %30s\n", "", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    printf("(in brackets: wall ns per instruction issued on one SIMD)\n");
    for (const Case &c : cases) {
        printf("%-12s", c.name);
        for (int wps : {1, 2, 4}) {
            const int blocks = cus * 4 * wps;
            double r[2], ns[2];
            for (int v = 0; v < 2; ++v) {
                kern_t k = v ? c.ind : c.dep;
                k<<<blocks, 64>>>(out, 1u, cyc);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                k<<<blocks, 64>>>(out, 2u, cyc);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
                double s = 0;
                for (int i = 0; i < blocks; ++i) s += (double)h[i];
                r[v] = s / blocks / ((double)kUnroll * kTrips);
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ns[v] = ms * 1e6 / ((double)kUnroll * kTrips * wps);  // wall ns per instruction issued on one SIMD
            }
            printf("  %6.2f|%6.2f (%4.2f|%4.2f ns)", r[0], r[1], ns[0], ns[1]);
        }
        printf("\n");
    }
    return 0;
}
