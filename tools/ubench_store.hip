// Micro-benchmark: HBM write bandwidth of the decoder's store pattern.
// Each wave owns 64 rows (stride ROW bytes); per round it writes SEG contiguous bytes
// to each row with 16-byte stores (SEG/16 lanes per row), then moves on by SEG.
// usage: ubench_store [rows_total] [row_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int SEG>
__global__ __launch_bounds__(64) void k_store(uint4 *out, size_t row_bytes, int rounds) {
    constexpr int LPR = SEG / 16;   // lanes per row segment
    constexpr int RPI = 64 / LPR;   // rows per instruction
    constexpr int IT = 64 / RPI;    // instructions per round (64 rows)
    const int lane = threadIdx.x;
    const size_t wave = blockIdx.x;
    char *base = reinterpret_cast<char *>(out) + wave * 64 * row_bytes;
    uint4 v = make_uint4(lane, wave, 0, 0);
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int row = i * RPI + lane / LPR, p = lane % LPR;
            *reinterpret_cast<uint4 *>(base + row * row_bytes + (size_t)r * SEG + 16 * p) = v;
        }
        v.z += 1;
    }
}

template <int SEG>
void run(uint4 *d, size_t rows, size_t row_bytes, int align_off) {
    const int rounds = (int)((row_bytes - 512) / SEG);
    const unsigned blocks = (unsigned)(rows / 64);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    char *p = reinterpret_cast<char *>(d) + align_off;
    k_store<SEG><<<blocks, 64>>>((uint4 *)p, row_bytes, rounds);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; ++i) k_store<SEG><<<blocks, 64>>>((uint4 *)p, row_bytes, rounds);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 3;
    double bytes = (double)rows * rounds * SEG;
    printf("SEG %4d B  align_off %3d: %.3f ms  %.1f GB/s\n", SEG, align_off, ms, bytes / ms / 1e6);
}

int main(int argc, char **argv) {
    size_t rows = argc > 1 ? atol(argv[1]) : 1000000;
    size_t row_bytes = argc > 2 ? atol(argv[2]) : 14000;
    rows = rows / 64 * 64;
    uint4 *d;
    if (hipMalloc(&d, rows * (row_bytes + 256) + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("rows %zu x %zu B = %.2f GB\n", rows, row_bytes, rows * row_bytes / 1e9);
    for (int off : {0}) {
        run<32>(d, rows, row_bytes, off);
        run<64>(d, rows, row_bytes, off);
        run<128>(d, rows, row_bytes, off);
        run<256>(d, rows, row_bytes, off);
        run<512>(d, rows, row_bytes, off);
        run<1024>(d, rows, row_bytes, off);
    }
    // row stride that is a multiple of 128: every segment line-aligned
    const size_t padded = (row_bytes + 127) / 128 * 128;  // <= row_bytes + 127: inside the allocation
    printf("row stride %zu (128-byte multiple):\n", padded);
    run<64>(d, rows, padded, 0);
    run<128>(d, rows, padded, 0);
    run<256>(d, rows, padded, 0);
    (void)hipFree(d);
    return 0;
}
