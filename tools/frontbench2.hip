// tools/frontbench2.hip -- second design sandbox for K1 (see frontbench.hip): does input prefetch
// depth matter when the refill loads share the vector-memory pipeline with the flush stores?
// Every iteration ("round") of a lane: [request 32 B of its input front] [store one 64- or 128-byte
// unit of each ready row, cooperatively] [a chain of dependent LDS reads = decode] [consume the
// request issued PF rounds ago: the next chain depends on it].  Not part of the product.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr uint32_t kOutStride = 64u << 10, kInStride = 16u << 10;

template <int U, int PF, bool LOAD, bool STORE>
__global__ __launch_bounds__(64) void k_fronts(const uint8_t *in, uint8_t *out, uint32_t delay, uint32_t *sink) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 1024; i += 64) lds[i] = (i * 2654435761u) & 1023u;
    __syncthreads();
    const uint64_t s0 = static_cast<uint64_t>(blockIdx.x) * 64;
    const uint8_t *ip = in + (s0 + lane) * kInStride;
    constexpr int LPR = U / 16, RPI = 64 / LPR, NST = U / 16;
    constexpr int ROUNDS_PER_UNIT = U / 64;      // a round produces 64 bytes per row
    uint4 t[PF + 1][2];
#pragma unroll
    for (int i = 0; i <= PF; i++) t[i][0] = t[i][1] = make_uint4(0, 0, 0, 0);
    uint32_t chain = lane, ldoff = 0;
    const uint32_t rounds = kOutStride / 64;
    for (uint32_t r0 = 0; r0 < rounds; r0 += PF + 1) {
#pragma unroll
        for (int i = 0; i <= PF; i++) {
            const uint32_t r = r0 + i;
            if (LOAD && (r & 1u) == 0) {         // 32 B per two rounds = 16 B per 64 B of output
                t[i][0] = *reinterpret_cast<const uint4 *>(ip + ldoff);
                t[i][1] = *reinterpret_cast<const uint4 *>(ip + ldoff + 16);
                ldoff += 32;
            }
            if (STORE && (r % ROUNDS_PER_UNIT) == ROUNDS_PER_UNIT - 1) {
                const uint32_t unit = r / ROUNDS_PER_UNIT;
#pragma unroll
                for (int k = 0; k < NST; k++) {
                    const uint32_t row = k * RPI + lane / LPR;
                    uint8_t *p = out + (s0 + row) * kOutStride + static_cast<uint64_t>(unit) * U + (lane % LPR) * 16;
                    *reinterpret_cast<uint4 *>(p) = make_uint4(chain + k, r, lane, 7);
                }
            }
            for (uint32_t d = 0; d < delay; d++) chain = lds[chain & 1023u] + d;
            constexpr int j = 0;                  // consume the oldest slot: (i + 1) % (PF + 1)
            const uint4 a = t[(i + 1) % (PF + 1)][j], b = t[(i + 1) % (PF + 1)][1];
            chain += (a.x ^ b.y) & 1u;
        }
    }
    if (chain == 0x12345678u) sink[0] = chain;
}

template <int U, int PF, bool LOAD, bool STORE>
static void run(const char *name, const uint8_t *in, uint8_t *out, uint32_t n_waves, uint32_t delay, uint32_t lds_bytes, uint32_t *sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_fronts<U, PF, LOAD, STORE>), dim3(n_waves), dim3(64), lds_bytes, 0, in, out, delay, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    std::printf("%-12s U=%3d PF=%d delay=%3u lds=%5u  %7.2f ms\n", name, U, PF, delay, lds_bytes, best);
    std::fflush(stdout);
}

int main(int argc, char **argv) {
    const uint32_t n_waves = 9537;
    uint8_t *in, *out;
    uint32_t *sink;
    CK(hipMalloc(&in, size_t(n_waves) * 64 * kInStride + 4096));
    CK(hipMalloc(&out, size_t(n_waves) * 64 * kOutStride + 4096));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 1, size_t(n_waves) * 64 * kInStride));
    CK(hipMemset(out, 0, size_t(n_waves) * 64 * kOutStride));
    const uint32_t lds = 16384;
    for (uint32_t delay : {16u, 32u, 48u, 64u}) {
        run<64, 0, true, false>("load", in, out, n_waves, delay, lds, sink);
        run<64, 0, false, true>("store", in, out, n_waves, delay, lds, sink);
        run<64, 0, true, true>("load+store", in, out, n_waves, delay, lds, sink);
        run<64, 1, true, true>("load+store", in, out, n_waves, delay, lds, sink);
        run<64, 3, true, true>("load+store", in, out, n_waves, delay, lds, sink);
        run<128, 0, false, true>("store", in, out, n_waves, delay, lds, sink);
        run<128, 0, true, true>("load+store", in, out, n_waves, delay, lds, sink);
        run<128, 1, true, true>("load+store", in, out, n_waves, delay, lds, sink);
        run<128, 3, true, true>("load+store", in, out, n_waves, delay, lds, sink);
    }
    return 0;
}
