// tools/frontbench.hip -- memory-pattern microbenchmark behind K1's design choices.
// Models "lane = stream": every lane owns an input front (16 KiB apart) and an output front (64 KiB
// apart); per iteration a stream emits U bytes (stored cooperatively, U/16 lanes per stream) and
// consumes U/4 input bytes (loaded by its own lane in granules of R bytes).  An optional chain of
// dependent LDS reads stands in for the decode time between flushes.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o frontbench tools/frontbench.hip && ./frontbench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr uint32_t kOutStride = 64u << 10, kInStride = 16u << 10;

template <int U, int R, bool LOAD, bool STORE, bool HALF>
__global__ __launch_bounds__(64) void k_fronts(const uint8_t *in, uint8_t *out, uint32_t delay, uint32_t *sink) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 1024; i += 64) lds[i] = (i * 2654435761u) & 1023u;
    __syncthreads();
    const uint64_t s0 = static_cast<uint64_t>(blockIdx.x) * 64;
    const uint8_t *ip = in + (s0 + lane) * kInStride;
    constexpr int LPR = U / 16;                 // lanes per row
    constexpr int RPI = 64 / LPR;               // rows per store instruction
    constexpr int NST = U / 16;                 // store instructions per iteration (64 rows)
    constexpr int LOAD_EVERY = 4 * R / U > 0 ? 4 * R / U : 1;
    uint4 acc = make_uint4(lane, 1, 2, 3);
    uint32_t chain = lane;
    uint32_t ldoff = 0;
    for (uint32_t it2 = 0; it2 < (HALF ? 2 : 1) * kOutStride / U; it2++) {
        const uint32_t it = HALF ? it2 >> 1 : it2;
        if (LOAD && it2 % ((HALF ? 2 : 1) * LOAD_EVERY) == 0) {
#pragma unroll
            for (int j = 0; j < R / 16; j++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(ip + ldoff + 16 * j);
                acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
            }
            ldoff += R;
        }
        for (uint32_t d = 0; d < delay; d++) chain = lds[chain & 1023u] + d;
        if (STORE) {
#pragma unroll
            for (int k = 0; k < NST; k++) {
                const uint32_t row = k * RPI + lane / LPR;
                uint8_t *p = out + (s0 + row) * kOutStride + static_cast<uint64_t>(it) * U + (lane % LPR) * 16;
                if (!HALF || ((row ^ it2) & 1u)) *reinterpret_cast<uint4 *>(p) = make_uint4(acc.x + k, acc.y, acc.z, chain);
            }
        }
    }
    if (acc.x == 0x12345678u && chain == 77u) sink[0] = acc.y;
}

template <int U, int R, bool LOAD, bool STORE, bool HALF = false>
static void run(const char *name, const uint8_t *in, uint8_t *out, uint32_t n_waves, uint32_t delay, uint32_t lds_bytes, uint32_t *sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_fronts<U, R, LOAD, STORE, HALF>), dim3(n_waves), dim3(64), lds_bytes, 0, in, out, delay, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double ob = STORE ? double(n_waves) * 64 * kOutStride : 0, ib = LOAD ? double(n_waves) * 64 * kInStride : 0;
    std::printf("%-20s half=%d U=%3d R=%3d delay=%2u lds=%5u  %7.2f ms  %6.0f GB/s\n", name, int(HALF), U, R, delay, lds_bytes, best, (ob + ib) / best * 1e-6);
    std::fflush(stdout);
}

int main(int argc, char **argv) {
    const uint32_t n_waves = argc > 1 ? std::atoi(argv[1]) : 9537;          // 610k streams = the 10 GB archive
    uint8_t *in, *out;
    uint32_t *sink;
    CK(hipMalloc(&in, size_t(n_waves) * 64 * kInStride + 4096));
    CK(hipMalloc(&out, size_t(n_waves) * 64 * kOutStride + 4096));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 1, size_t(n_waves) * 64 * kInStride));
    CK(hipMemset(out, 0, size_t(n_waves) * 64 * kOutStride));
    for (uint32_t lds : {16384u, 20480u}) {
        for (uint32_t dl : {0u, 1u}) {
            const uint32_t d64 = dl ? 16 : 0, d128 = dl ? 32 : 0;
            run<64, 32, false, true>("store", in, out, n_waves, d64, lds, sink);
            run<64, 32, false, true, true>("store", in, out, n_waves, d64 / 2, lds, sink);
            run<128, 32, false, true>("store", in, out, n_waves, d128, lds, sink);
            run<128, 32, false, true, true>("store", in, out, n_waves, d128 / 2, lds, sink);
            run<64, 32, true, true>("load+store", in, out, n_waves, d64, lds, sink);
            run<64, 32, true, true, true>("load+store", in, out, n_waves, d64 / 2, lds, sink);
            run<128, 32, true, true>("load+store", in, out, n_waves, d128, lds, sink);
            run<128, 32, true, true, true>("load+store", in, out, n_waves, d128 / 2, lds, sink);
            run<128, 128, true, true, true>("load+store", in, out, n_waves, d128 / 2, lds, sink);
        }
    }
    return 0;
}
