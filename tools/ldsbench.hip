// tools/ldsbench.hip -- what the LDS access shapes of k_huf_decode cost (not product code).
// One-wave workgroups, W per CU; every lane issues `iters` x 16 independent LDS operations of one shape.
// Prints CU cycles per wave-instruction at the given occupancy (throughput, not latency).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *sink, int iters, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[16 * 1024];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = seed * 747796405u + lane * 2891336453u + blockIdx.x, acc = 0;
    uint8_t *row = lds + 4096 + lane * 136;
    const uint8_t *tbl = lds + (lane >> 2) * 512;          // 16 tables of 512 B, 4 lanes each
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t rnd = x >> 8;
            if (MODE == 0) acc += *reinterpret_cast<const uint32_t *>(tbl + ((rnd & 127u) << 2));            // random dword in the lane group's table
            if (MODE == 1) acc += *reinterpret_cast<const uint16_t *>(tbl + ((rnd & 255u) << 1));            // random u16
            if (MODE == 2) acc += *reinterpret_cast<const uint32_t *>(lds + ((rnd & 15u) << 2) + 8192);      // 16 hot dwords shared by the wave
            const uint32_t w4 = (rnd & 60u);                                                                  // varying dword offset inside the row
            if (MODE == 3) *reinterpret_cast<uint32_t *>(row + w4) = rnd;                            // aligned dword into the lane's row
            if (MODE == 4) { uint16_t *p = reinterpret_cast<uint16_t *>(row + w4 + 2); p[0] = rnd; p[1] = rnd >> 16; }   // 2-byte aligned dword, two b16 writes
            if (MODE == 8) { uint32_t a = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(row)) + w4 + 2; asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(rnd) : "memory"); }   // ONE dword write at a 2-byte aligned address (what hipcc emits in k_huf_decode)
            if (MODE == 5) acc += *reinterpret_cast<const uint32_t *>(lds + 12288 + (rnd & 15u) * 256 + lane * 4);   // transposed ring read: conflict-free
            if (MODE == 6) { const uint2 v = *reinterpret_cast<const uint2 *>(row + (rnd & 56u)); acc += v.x + v.y; }   // b64 from the lane's row
            if (MODE == 7) *reinterpret_cast<uint2 *>(row + (rnd & 56u)) = make_uint2(rnd, u);     // b64 into the lane's row
            if (MODE == 10) { volatile uint16_t *p = reinterpret_cast<volatile uint16_t *>(row + w4 + 2); p[0] = rnd; p[1] = rnd >> 16; }   // two separate b16 writes
            if (MODE == 11) { uint32_t a = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(row)) + w4; asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(a), "v"(rnd), "v"(acc) : "memory"); }   // two consecutive aligned dwords, one op
            if (MODE == 12) { uint32_t a = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(row)) + w4; unsigned long long v = (static_cast<unsigned long long>(acc) << 32) | rnd; asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }   // b64 at a 4-byte aligned address
            if (MODE == 13) { *reinterpret_cast<volatile uint16_t *>(row + w4 + 2) = rnd; }   // one b16 write
            if (MODE == 9) { const uint4 v = *reinterpret_cast<const uint4 *>(lds + 4096 + ((rnd >> 3) & 63u) * 136 + (lane & 3) * 16); acc += v.x + v.w; }   // b128: 4 lanes per random row (flush)
        }
    }
    __syncthreads();
    for (uint32_t i = 0; i < 136; i += 4) acc += *reinterpret_cast<const uint32_t *>(row + i);
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int MODE>
float run(int wg, int iters, uint32_t *d) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(64), 0, 0, d, 10, 1u);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(64), 0, 0, d, iters, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main(int argc, char **argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 8, iters = 5000;
    uint32_t *d;
    hipMalloc(&d, 64);
    const int wg = 256 * per_cu;
    const char *names[14] = {"read_b32 random (table)", "read_u16 random", "read_b32 16 hot dwords", "write_b32 aligned row", "write 2xb16 misaligned row",
                            "read_b32 transposed ring", "read_b64 row", "write_b64 row", "write_b32 at 2 mod 4 (one op)", "read_b128 4 lanes/random row", "2 x write_b16 (kept apart)", "write2_b32 (8 B, 4-aligned)", "write_b64 (4-aligned)", "1 x write_b16"};
    float ms[14] = {run<0>(wg, iters, d), run<1>(wg, iters, d), run<2>(wg, iters, d), run<3>(wg, iters, d),
                    run<4>(wg, iters, d), run<5>(wg, iters, d), run<6>(wg, iters, d), run<7>(wg, iters, d), run<8>(wg, iters, d), run<9>(wg, iters, d), run<10>(wg, iters, d), run<11>(wg, iters, d), run<12>(wg, iters, d), run<13>(wg, iters, d)};
    for (int m = 0; m < 14; m++) {
        const double instr_per_cu = double(per_cu) * iters * 16;
        printf("%-28s %2d waves/CU: %8.3f ms  -> %.2f cycles per wave-instruction per CU (at 2.1 GHz)\n", names[m], per_cu, ms[m],
               ms[m] * 1e-3 * 2.1e9 / instr_per_cu);
    }
    return 0;
}
