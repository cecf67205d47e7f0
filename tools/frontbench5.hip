// tools/frontbench5.hip -- fifth sandbox for K1's memory pattern: the floor of the product's OWN pattern.  frontbench2/3 fed
// the fronts with 32-byte pieces; the product (K1 v4) requests every 128-byte input line whole, once (8 x dwordx4 per lane),
// and writes 128-byte units, eight lines per store instruction.  This kernel does exactly that and nothing else (a short chain
// of dependent LDS reads stands in for the decode), on chunk-backed memory like the product's: load only, store only, both.
// Not part of the product.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr uint32_t kOutStride = 64u << 10, kInStride = 16u << 10, kUnit = 128, kUnits = kOutStride / kUnit;
constexpr uint32_t kWaves = 9537;

template <bool LOAD, bool STORE>
__global__ __launch_bounds__(64) void k_fronts(const uint8_t *in, uint8_t *out, uint32_t delay, uint32_t *sink) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 1024; i += 64) lds[i] = (i * 2654435761u) & 1023u;
    __syncthreads();
    const uint64_t s0 = static_cast<uint64_t>(blockIdx.x) * 64;
    const uint8_t *ip = in + (s0 + lane) * kInStride;
    uint32_t chain = lane;
    uint4 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t unit = 0; unit < kUnits; unit++) {
        if (LOAD && (unit & 3u) == 0) {            // one whole input line per four output units (4 : 1, as the product)
            chain += (t[0].x ^ t[3].y ^ t[7].z) & 1u;      // the line before is used up
#pragma unroll
            for (int i = 0; i < 8; i++) t[i] = *reinterpret_cast<const uint4 *>(ip + (unit >> 2) * 128 + 16 * i);
        }
        if (STORE) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t row = k * 8 + lane / 8;
                uint8_t *p = out + (s0 + row) * kOutStride + static_cast<uint64_t>(unit) * kUnit + (lane % 8) * 16;
                *reinterpret_cast<uint4 *>(p) = make_uint4(chain + k, unit, lane, 7);
            }
        }
        for (uint32_t d = 0; d < delay; d++) chain = lds[chain & 1023u] + d;
    }
    chain += (t[0].x ^ t[3].y ^ t[7].z) & 1u;
    if (chain == 0x12345678u) sink[0] = chain;
}

static void *mapped(size_t bytes) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t chunk = size_t(1) << 30, total = (bytes + chunk - 1) / chunk * chunk;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, size_t(2) << 20, nullptr, 0));
    for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap(static_cast<char *>(va) + off, chunk, 0, h, 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    return va;
}

template <bool LOAD, bool STORE>
static float run(const uint8_t *in, uint8_t *out, uint32_t delay, uint32_t lds_bytes, uint32_t *sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_fronts<LOAD, STORE>), dim3(kWaves), dim3(64), lds_bytes, 0, in, out, delay, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t *sink;
    CK(hipMalloc(&sink, 64));
    uint8_t *in = static_cast<uint8_t *>(mapped(size_t(kWaves) * 64 * kInStride + 4096));
    uint8_t *out = static_cast<uint8_t *>(mapped(size_t(kWaves) * 64 * kOutStride + 4096));
    CK(hipMemset(in, 1, size_t(kWaves) * 64 * kInStride));
    for (uint32_t lds : {20480u, 10240u}) {        // eight / sixteen waves per CU
        for (uint32_t delay : {0u, 8u, 24u, 48u}) {
            std::printf("lds %5u delay %2u | load only %6.2f  store only %6.2f  load + store %6.2f ms\n", lds, delay,
                        run<true, false>(in, out, delay, lds, sink), run<false, true>(in, out, delay, lds, sink),
                        run<true, true>(in, out, delay, lds, sink));
            std::fflush(stdout);
        }
    }
    return 0;
}
