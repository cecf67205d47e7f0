// tools/frontbench4.hip -- fourth sandbox for K1's memory pattern: frontbench3 showed that the phase of the write fronts does
// not matter and that the ALLOCATION does (store-only 6.65 or 7.0 ms for the same 40 GB at the same virtual address).  Here the
// same store-only kernel (610 k fronts, 128-byte units, eight lines per store instruction) runs on 40 GB windows
// (a) at six offsets of ONE 240 GB hipMalloc, (b) on six hipMallocs one after the other, (c) on hipExtMallocWithFlags(
// hipDeviceMallocContiguous), (d) on virtual ranges backed by hipMemCreate chunks of 2 MiB, 64 MiB and 1 GiB.
// Not part of the product.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr uint32_t kOutStride = 64u << 10, kUnit = 128, kUnits = kOutStride / kUnit;
constexpr uint32_t kWaves = 9537;
constexpr size_t kBytes = size_t(kWaves) * 64 * kOutStride;

__global__ __launch_bounds__(64) void k_fronts(uint8_t *out, uint32_t delay, uint32_t *sink) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 1024; i += 64) lds[i] = (i * 2654435761u) & 1023u;
    __syncthreads();
    const uint64_t s0 = static_cast<uint64_t>(blockIdx.x) * 64;
    uint32_t chain = lane;
    for (uint32_t unit = 0; unit < kUnits; unit++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t row = k * 8 + lane / 8;
            uint8_t *p = out + (s0 + row) * kOutStride + static_cast<uint64_t>(unit) * kUnit + (lane % 8) * 16;
            *reinterpret_cast<uint4 *>(p) = make_uint4(chain + k, unit, lane, 7);
        }
        for (uint32_t d = 0; d < delay; d++) chain = lds[chain & 1023u] + d;
    }
    if (chain == 0x12345678u) sink[0] = chain;
}

// a plain streaming fill of the same bytes, for comparison (one 16-byte store per thread and step, grid-stride)
__global__ __launch_bounds__(256) void k_stream(uint4 *out, size_t n16) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += size_t(gridDim.x) * 256) out[i] = make_uint4(1, 2, 3, 4);
}

static uint32_t *g_sink;

static void measure(const char *what, int idx, uint8_t *out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f, best_s = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        float ms;
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_fronts, dim3(kWaves), dim3(64), 20480, 0, out, 8u, g_sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<uint4 *>(out), kBytes / 16);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best_s) best_s = ms;
    }
    std::printf("%-28s %d  out %p  fronts %6.2f ms = %4.2f TB/s   streaming fill %6.2f ms = %4.2f TB/s\n", what, idx, static_cast<void *>(out),
                best, kBytes / best / 1e9, best_s, kBytes / best_s / 1e9);
    std::fflush(stdout);
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
}

static void vmm(size_t chunk, const char *name) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk < gran) chunk = gran;
    const size_t total = (kBytes + chunk - 1) / chunk * chunk;
    for (int rep = 0; rep < 3; rep++) {
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, total, size_t(2) << 20, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t off = 0; off < total; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, chunk, &prop, 0));
            CK(hipMemMap(static_cast<char *>(va) + off, chunk, 0, h, 0));
            hs.push_back(h);
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, total, &acc, 1));
        measure(name, rep, static_cast<uint8_t *>(va));
        CK(hipMemUnmap(va, total));
        for (auto h : hs) CK(hipMemRelease(h));
        CK(hipMemAddressFree(va, total));
    }
}

int main(int argc, char **argv) {
    CK(hipMalloc(&g_sink, 64));
    {
        uint8_t *pool;
        const size_t pool_bytes = size_t(240) << 30;
        CK(hipMalloc(&pool, pool_bytes));
        for (int i = 0; i < 6; i++) {
            uint8_t *out = pool + (size_t(i) * 40 << 30);
            if (size_t(out - pool) + kBytes > pool_bytes) out = pool + pool_bytes - kBytes;
            measure("one 240 GB hipMalloc, window", i, out);
        }
        CK(hipFree(pool));
    }
    for (int i = 0; i < 6; i++) {
        uint8_t *out;
        CK(hipMalloc(&out, kBytes));
        measure("hipMalloc", i, out);
        CK(hipFree(out));
    }
    for (int i = 0; i < 4; i++) {
        uint8_t *out;
        if (hipExtMallocWithFlags(reinterpret_cast<void **>(&out), kBytes, hipDeviceMallocContiguous) != hipSuccess) {
            std::printf("hipDeviceMallocContiguous: not available\n");
            (void)hipGetLastError();
            break;
        }
        measure("hipDeviceMallocContiguous", i, out);
        CK(hipFree(out));
    }
    vmm(size_t(2) << 20, "hipMemCreate 2 MiB chunks");
    vmm(size_t(64) << 20, "hipMemCreate 64 MiB chunks");
    vmm(size_t(1) << 30, "hipMemCreate 1 GiB chunks");
    return 0;
}
