// ubench_flag.hip -- what a flag exchange between two workgroups costs on the MI355X: same XCD or another, agent-scope or
// workgroup-scope atomics (the decoupled look-backs of the encoders, the block decoder and the inverse filter are chains of
// such exchanges).  Two single-wavefront workgroups play ping-pong through two words of device memory; the others of the grid
// exit at once.  Workgroups are dealt to the XCDs round-robin, so workgroups 0 and 1 sit on different XCDs and 0 and 8 on the
// same one; every workgroup reports its XCC_ID so that the pairing can be checked.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_flag.hip -o /tmp/ubench_flag && /tmp/ubench_flag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

// 0: agent scope, 1: workgroup scope, 2: polls through the SCALAR cache with glc (stores agent scope) -- the scalar path does
// not queue behind a CU's vector memory traffic (round 1: the header walkers), but is it coherent?
template <int SCOPE>
__device__ __forceinline__ uint32_t ld(const uint32_t *p) {
    if (SCOPE == 2) {
        uint32_t v;
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
        return v;
    }
    return SCOPE == 0 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int SCOPE>
__device__ __forceinline__ void st(uint32_t *p, uint32_t v) {
    if (SCOPE == 0 || SCOPE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// workgroup `a` serves, workgroup `b` answers; `iters` round trips; flags[0] a -> b, flags[32] b -> a (different lines)
template <int SCOPE>
__global__ void k_pingpong(uint32_t *flags, uint32_t *xcc, uint32_t a, uint32_t b, uint32_t iters, unsigned long long *cycles,
                           uint32_t *timeouts) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    if (threadIdx.x != 0 || (blockIdx.x != a && blockIdx.x != b)) return;
    const bool server = blockIdx.x == a;
    uint32_t *mine = flags + (server ? 0 : 32), *theirs = flags + (server ? 32 : 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t lost = 0;
    for (uint32_t i = 1; i <= iters; ++i) {
        if (server) st<SCOPE>(mine, i);
        uint32_t spins = 0;
        while (ld<SCOPE>(theirs) != i) {
            if (++spins > (1u << 22)) { ++lost; break; }  // never hang: a scope that does not reach the other side shows up here
        }
        if (lost) break;
        if (!server) st<SCOPE>(mine, i);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (server) { cycles[0] = t1 - t0; timeouts[0] = lost; }
}

// one wavefront polls a word nobody writes: what a poll alone costs
template <int SCOPE>
__global__ void k_poll(const uint32_t *flag, uint32_t iters, unsigned long long *cycles, uint32_t *sink) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) acc += ld<SCOPE>(flag);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    cycles[0] = t1 - t0;
    sink[0] = acc;
}

int main() {
    uint32_t *flags, *xcc, *timeouts, *sink;
    unsigned long long *cycles;
    CHECK(hipMalloc(&flags, 4096));
    CHECK(hipMalloc(&xcc, 64 * 4));
    CHECK(hipMalloc(&timeouts, 4));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&cycles, 8));
    const uint32_t iters = 2000;
    struct Case { const char *name; int scope; uint32_t a, b, grid; };
    const Case cases[] = {
        {"agent scope, workgroups 0 and 1 (two XCDs)", 0, 0, 1, 16},
        {"agent scope, workgroups 0 and 8 (one XCD)", 0, 0, 8, 16},
        {"workgroup scope, workgroups 0 and 8 (one XCD)", 1, 0, 8, 16},
        {"workgroup scope, workgroups 0 and 1 (two XCDs: may never arrive)", 1, 0, 1, 16},
        {"scalar glc polls, workgroups 0 and 8 (one XCD)", 2, 0, 8, 16},
        {"scalar glc polls, workgroups 0 and 1 (two XCDs)", 2, 0, 1, 16},
    };
    for (const Case &c : cases) {
        CHECK(hipMemset(flags, 0, 4096));
        CHECK(hipMemset(timeouts, 0, 4));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        if (c.scope == 0) k_pingpong<0><<<c.grid, 64>>>(flags, xcc, c.a, c.b, iters, cycles, timeouts);
        else if (c.scope == 1) k_pingpong<1><<<c.grid, 64>>>(flags, xcc, c.a, c.b, iters, cycles, timeouts);
        else k_pingpong<2><<<c.grid, 64>>>(flags, xcc, c.a, c.b, iters, cycles, timeouts);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint32_t> hx(16);
        uint32_t lost = 0;
        unsigned long long cyc = 0;
        CHECK(hipMemcpy(hx.data(), xcc, 16 * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&lost, timeouts, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&cyc, cycles, 8, hipMemcpyDeviceToHost));
        printf("%-68s XCC %u / %u: %s%.0f ns per round trip (two hops; %.3f ms for %u)\n", c.name, hx[c.a], hx[c.b],
               lost ? "TIMED OUT, " : "", lost ? 0.0 : ms * 1e6 / iters, ms, iters);
    }
    for (int scope = 0; scope < 3; ++scope) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        if (scope == 0) k_poll<0><<<1, 64>>>(flags, 20000, cycles, sink);
        else if (scope == 1) k_poll<1><<<1, 64>>>(flags, 20000, cycles, sink);
        else k_poll<2><<<1, 64>>>(flags, 20000, cycles, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("a dependent poll of an idle word, %s scope: %.0f ns\n", scope == 0 ? "agent" : (scope == 1 ? "workgroup" : "scalar glc,"), ms * 1e6 / 20000);
    }
    printf("XCC ids of workgroups 0..15:");
    std::vector<uint32_t> hx(16);
    CHECK(hipMemcpy(hx.data(), xcc, 16 * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) printf(" %u", hx[i]);
    printf("\n");
    return 0;
}
