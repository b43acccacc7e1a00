// Micro-benchmark: the MEMORY side of the lane-per-waveform decoder alone (no Rice parse) -- what the chip gives this access
// pattern when nothing else limits it.  Each wavefront owns 64 streams: per round every stream's 128-byte output line is
// written whole (8 lanes x 16 bytes, as the decoder's write-out does), and a lane fetches the next PIECE bytes of its own
// compressed stream (16-byte loads, one lane per stream) whenever the round's consumption (13 words) crosses a piece.
// usage: ubench_pattern [waveforms] [out_row_bytes] [in_row_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int PIECE, bool LOADS, bool STORES>
__global__ __launch_bounds__(64) void k_pattern(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t in_row, size_t out_row, int rounds) {
    extern __shared__ uint32_t pad_[];
    const int lane = threadIdx.x;
    const size_t wave = blockIdx.x;
    const char *src = reinterpret_cast<const char *>(in) + (wave * 64 + lane) * in_row;
    src = reinterpret_cast<const char *>((reinterpret_cast<uintptr_t>(src) + PIECE - 1) & ~(uintptr_t)(PIECE - 1));
    char *base = reinterpret_cast<char *>(out) + wave * 64 * out_row;
    uint4 v = make_uint4(lane, (uint32_t)wave, 0, 0);
    uint32_t words = 0, fetched = 0;
    for (int r = 0; r < rounds; ++r) {
        if (LOADS) {
            words += 13;  // 6.47 bits per sample x 64 samples
            if (words * 4 > fetched) {  // (same round for every lane: the real lanes drift, the byte count is what matters)
#pragma unroll
                for (int j = 0; j < PIECE / 16; ++j) {
                    const uint4 w = *reinterpret_cast<const uint4 *>(src + fetched + 16 * j);
                    v.x ^= w.x; v.y ^= w.y; v.z ^= w.z; v.w ^= w.w;
                }
                fetched += PIECE;
            }
        }
        if (STORES) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = i * 8 + lane / 8, p = lane % 8;
                *reinterpret_cast<uint4 *>(base + row * out_row + (size_t)r * 128 + 16 * p) = v;
            }
        }
        v.z += 1;
    }
    if (!STORES) *reinterpret_cast<uint4 *>(base + lane * out_row) = v;
}

// Reference points for the stores (round 4): the same 1M rows written W bytes per visit (W = 128 is k_pattern's store side: a
// wavefront's 1 KB store instruction covers 1024 / W rows), and a LINEAR write of the same bytes (every instruction 1 KB of
// consecutive addresses, workgroups striding over the buffer) -- is "a 128-byte line per waveform and round" itself the cost,
// or do 14 GB of stores never go faster?
template <int W>
__global__ __launch_bounds__(64) void k_rows(uint4 *__restrict__ out, size_t out_row, int visits) {
    extern __shared__ uint32_t pad_[];
    const int lane = threadIdx.x;
    char *base = reinterpret_cast<char *>(out) + (size_t)blockIdx.x * 64 * out_row;
    constexpr int LPR = W / 16;       // lanes per row
    constexpr int RPI = 64 / LPR;     // rows per store instruction
    uint4 v = make_uint4(lane, blockIdx.x, 0, 0);
    for (int r = 0; r < visits; ++r) {
#pragma unroll
        for (int i = 0; i < 64 / RPI; ++i) {
            const int row = i * RPI + lane / LPR, p = lane % LPR;
            *reinterpret_cast<uint4 *>(base + row * out_row + (size_t)r * W + 16 * p) = v;
        }
        v.z += 1;
    }
}
__global__ __launch_bounds__(64) void k_linear_write(uint4 *__restrict__ out, size_t n16, int per_wave) {
    extern __shared__ uint32_t pad_[];
    uint4 v = make_uint4(threadIdx.x, blockIdx.x, 0, 0);
    // a wavefront owns per_wave consecutive KB (what a copy kernel does)
    size_t i = (size_t)blockIdx.x * per_wave * 64 + threadIdx.x;
    for (int r = 0; r < per_wave; ++r, i += 64)
        if (i < n16) out[i] = v;
}
__global__ __launch_bounds__(64) void k_linear_read(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16, int per_wave) {
    extern __shared__ uint32_t pad_[];
    uint4 v = make_uint4(0, 0, 0, 0);
    size_t i = (size_t)blockIdx.x * per_wave * 64 + threadIdx.x;
    for (int r = 0; r < per_wave; ++r, i += 64)
        if (i < n16) { const uint4 w = in[i]; v.x ^= w.x; v.y ^= w.y; v.z ^= w.z; v.w ^= w.w; }
    if (v.x == 0x12345678u) out[threadIdx.x] = v;
}
template <typename F>
float time3(F &&launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

template <int PIECE, bool LOADS, bool STORES>
void run(const uint4 *din, uint4 *dout, size_t rows, size_t in_row, size_t out_row, int rounds, unsigned lds) {
    const unsigned blocks = (unsigned)(rows / 64);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_pattern<PIECE, LOADS, STORES><<<blocks, 64, lds>>>(din, dout, in_row, out_row, rounds);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; ++i) k_pattern<PIECE, LOADS, STORES><<<blocks, 64, lds>>>(din, dout, in_row, out_row, rounds);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double wr = STORES ? (double)rows * rounds * 128 : 0, rd = LOADS ? (double)rows * rounds * 52 : 0;
    printf("piece %3d  %s%s  LDS %6u B/wave (%2u waves/CU): %.3f ms  %.2f TB/s (%.1f GB written, %.1f GB read)\n", PIECE, LOADS ? "loads " : "", STORES ? "stores" : "",
           lds, lds ? 163840u / lds : 32u, ms, (wr + rd) / ms / 1e9, wr / 1e9, rd / 1e9);
}

int main(int argc, char **argv) {
    size_t rows = argc > 1 ? atol(argv[1]) : 1000000;
    size_t out_row = argc > 2 ? atol(argv[2]) : 14080;  // (128-byte multiple: the decoder's start delays make its lines whole)
    size_t in_row = argc > 3 ? atol(argv[3]) : 5660;
    rows = rows / 64 * 64;
    const int rounds = 109;
    uint4 *din, *dout;
    if (hipMalloc(&din, rows * in_row + 65536) != hipSuccess || hipMalloc(&dout, rows * out_row + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(din, 1, rows * in_row + 65536);
    printf("%zu waveforms: out rows of %zu B, streams of %zu B, %d rounds\n", rows, out_row, in_row, rounds);
    for (unsigned lds : {27000u, 20000u, 18000u, 13500u, 10000u, 5000u}) {
        run<64, true, true>(din, dout, rows, in_row, out_row, rounds, lds);
        run<128, true, true>(din, dout, rows, in_row, out_row, rounds, lds);
    }
    run<64, false, true>(din, dout, rows, in_row, out_row, rounds, 20000u);
    run<64, true, false>(din, dout, rows, in_row, out_row, rounds, 20000u);
    run<128, true, false>(din, dout, rows, in_row, out_row, rounds, 20000u);
    // ---- reference points (round 4) ----
    const size_t out_bytes = rows * (out_row / 512) * 512;  // (14.0 GB)
    for (unsigned lds : {20000u, 5000u}) {
        float ms = time3([&] { k_rows<128><<<(unsigned)(rows / 64), 64, lds>>>(dout, out_row, (int)(out_row / 128)); });
        printf("rows, 128 B per visit  (%2u waves/CU): %.3f ms  %.2f TB/s\n", 163840u / lds, ms, rows * (out_row / 128) * 128.0 / ms / 1e9);
        ms = time3([&] { k_rows<256><<<(unsigned)(rows / 64), 64, lds>>>(dout, out_row, (int)(out_row / 256)); });
        printf("rows, 256 B per visit  (%2u waves/CU): %.3f ms  %.2f TB/s\n", 163840u / lds, ms, rows * (out_row / 256) * 256.0 / ms / 1e9);
        ms = time3([&] { k_rows<512><<<(unsigned)(rows / 64), 64, lds>>>(dout, out_row, (int)(out_row / 512)); });
        printf("rows, 512 B per visit  (%2u waves/CU): %.3f ms  %.2f TB/s\n", 163840u / lds, ms, rows * (out_row / 512) * 512.0 / ms / 1e9);
        ms = time3([&] { k_rows<1024><<<(unsigned)(rows / 64), 64, lds>>>(dout, out_row, (int)(out_row / 1024)); });
        printf("rows, 1024 B per visit (%2u waves/CU): %.3f ms  %.2f TB/s\n", 163840u / lds, ms, rows * (out_row / 1024) * 1024.0 / ms / 1e9);
        for (int per_wave : {16, 220}) {  // 16 KB / 220 KB of consecutive addresses per wavefront
            const size_t n16 = out_bytes / 16;
            const unsigned blocks = (unsigned)((n16 + (size_t)per_wave * 64 - 1) / ((size_t)per_wave * 64));
            ms = time3([&] { k_linear_write<<<blocks, 64, lds>>>(dout, n16, per_wave); });
            printf("linear write, %3d KB per wavefront (%2u waves/CU): %.3f ms  %.2f TB/s\n", per_wave, 163840u / lds, ms, out_bytes / ms / 1e9);
        }
        {
            const size_t n16 = rows * in_row / 16;
            const int per_wave = 16;
            const unsigned blocks = (unsigned)((n16 + (size_t)per_wave * 64 - 1) / ((size_t)per_wave * 64));
            ms = time3([&] { k_linear_read<<<blocks, 64, lds>>>(din, dout, n16, per_wave); });
            printf("linear read of the streams' 5.7 GB (%2u waves/CU): %.3f ms  %.2f TB/s\n", 163840u / lds, ms, n16 * 16.0 / ms / 1e9);
        }
    }
    return 0;
}
