// tools/frontbench3.hip -- third sandbox for K1's memory pattern (see frontbench.hip, frontbench2.hip): does it matter that
// all write fronts move in LOCKSTEP?  610 k streams of exactly 64 KiB of output each, one lane per stream, every lane of
// every resident wave at about the same offset x inside its 64 KiB region at any time: the low sixteen address bits of all
// ~131 k live fronts agree, and which DRAM channel a front hits is left to what the address hash makes of the upper bits,
// i.e. to the physical placement of the buffer (K1 takes 10.7-11.9 ms depending on the allocation).
// PHASE 0: lockstep, as in the product.  PHASE 1: every wave starts at its own unit of the region and wraps around.
// PHASE 2: every row (stream) does.  The buffers are freed and allocated again between repetitions.  Not part of the product.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr uint32_t kOutStride = 64u << 10, kInStride = 16u << 10, kUnit = 128, kUnits = kOutStride / kUnit;

template <int PHASE, bool LOAD>
__global__ __launch_bounds__(64) void k_fronts(const uint8_t *in, uint8_t *out, uint32_t delay, uint32_t *sink) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 1024; i += 64) lds[i] = (i * 2654435761u) & 1023u;
    __syncthreads();
    const uint64_t s0 = static_cast<uint64_t>(blockIdx.x) * 64;
    const uint8_t *ip = in + (s0 + lane) * kInStride;
    uint32_t chain = lane, ldoff = 0;
    uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
    for (uint32_t unit = 0; unit < kUnits; unit++) {
        if (LOAD) {                                // 32 B of input per 128 B of output
            t0 = *reinterpret_cast<const uint4 *>(ip + ldoff);
            t1 = *reinterpret_cast<const uint4 *>(ip + ldoff + 16);
            ldoff += 32;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {              // store instruction k: rows 8k .. 8k+7, eight lanes per row
            const uint32_t row = k * 8 + lane / 8;
            uint32_t ph = 0;
            if (PHASE == 1) ph = (blockIdx.x * 2654435761u) >> 23;
            if (PHASE == 2) ph = ((static_cast<uint32_t>(s0) + row) * 2654435761u) >> 23;
            const uint32_t u = (unit + ph) % kUnits;
            uint8_t *p = out + (s0 + row) * kOutStride + static_cast<uint64_t>(u) * kUnit + (lane % 8) * 16;
            *reinterpret_cast<uint4 *>(p) = make_uint4(chain + k, unit, lane, 7);
        }
        for (uint32_t d = 0; d < delay; d++) chain = lds[chain & 1023u] + d;
        chain += (t0.x ^ t1.y) & 1u;
    }
    if (chain == 0x12345678u) sink[0] = chain;
}

template <int PHASE, bool LOAD>
static float run(const uint8_t *in, uint8_t *out, uint32_t n_waves, uint32_t delay, uint32_t lds_bytes, uint32_t *sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_fronts<PHASE, LOAD>), dim3(n_waves), dim3(64), lds_bytes, 0, in, out, delay, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return best;
}

int main(int argc, char **argv) {
    const uint32_t n_waves = 9537;
    const uint32_t lds = 20480;                    // eight waves per CU, as K1
    uint32_t *sink;
    CK(hipMalloc(&sink, 64));
    for (int alloc = 0; alloc < 6; alloc++) {
        uint8_t *in, *out;
        CK(hipMalloc(&in, size_t(n_waves) * 64 * kInStride + 4096));
        CK(hipMalloc(&out, size_t(n_waves) * 64 * kOutStride + 4096));
        CK(hipMemset(in, 1, size_t(n_waves) * 64 * kInStride));
        CK(hipMemset(out, 0, size_t(n_waves) * 64 * kOutStride));
        for (uint32_t delay : {8u, 24u}) {
            std::printf("alloc %d out %p delay %2u | store only: lockstep %6.2f  per-wave %6.2f  per-row %6.2f | load+store: lockstep %6.2f  per-wave %6.2f  per-row %6.2f ms\n",
                        alloc, static_cast<void *>(out), delay,
                        run<0, false>(in, out, n_waves, delay, lds, sink), run<1, false>(in, out, n_waves, delay, lds, sink),
                        run<2, false>(in, out, n_waves, delay, lds, sink), run<0, true>(in, out, n_waves, delay, lds, sink),
                        run<1, true>(in, out, n_waves, delay, lds, sink), run<2, true>(in, out, n_waves, delay, lds, sink));
            std::fflush(stdout);
        }
        CK(hipFree(in));
        CK(hipFree(out));
    }
    return 0;
}
