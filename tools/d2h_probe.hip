// tools/d2h_probe.hip -- device-to-host through a 64 MiB pinned window, three ways: hipMemcpyAsync (the runtime's copy engines),
// a kernel that stores into the pinned buffer over PCIe, and hipMemcpyAsync with a window from hipHostMallocNumaUser-less
// plain flags.  Run it several times in a row: the iterator's D2H rate differs between the first GPU process on a box and
// later ones, and this tells whether the copy engines are what differs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void k_copy(const uint4 *__restrict__ src, uint4 *dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t win = size_t(64) << 20, total = size_t(16) << 30, dev_n = size_t(2) << 30;
    uint8_t *d = nullptr, *h = nullptr, *h2 = nullptr;
    // argv[1] == "vmm": the device buffer as the product's large buffers are made (an address range backed by 1 GiB hipMemCreate chunks)
    if (argc > 1 && !strcmp(argv[1], "vmm")) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        void *va = nullptr;
        if (hipMemAddressReserve(&va, dev_n, size_t(2) << 20, nullptr, 0) != hipSuccess) return 2;
        for (size_t off = 0; off < dev_n; off += size_t(1) << 30) {
            hipMemGenericAllocationHandle_t hnd;
            if (hipMemCreate(&hnd, size_t(1) << 30, &prop, 0) != hipSuccess || hipMemMap((char *)va + off, size_t(1) << 30, 0, hnd, 0) != hipSuccess) return 3;
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(va, dev_n, &acc, 1) != hipSuccess) return 4;
        d = (uint8_t *)va;
        printf("(device buffer: mapped chunks)\n");
    } else if (hipMalloc(&d, dev_n) != hipSuccess) return 1;
    if (hipHostMalloc(&h, win) != hipSuccess || hipHostMalloc(&h2, win, hipHostMallocNonCoherent) != hipSuccess) return 1;
    hipMemset(d, 1, dev_n);
    hipStream_t s;
    hipStreamCreate(&s);
    volatile uint64_t sink = 0;
    for (int mode = 0; mode < 4; mode++) {
        uint8_t *hw = mode == 3 ? h2 : h;
        const double t0 = now();
        for (size_t off = 0; off < total; off += win) {
            const uint8_t *src = d + (off % dev_n);
            if (mode == 0 || mode == 3) {
                hipMemcpyAsync(hw, src, win, hipMemcpyDeviceToHost, s);
            } else if (mode == 1) {
                hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s, (const uint4 *)src, (uint4 *)hw, win / 16);
            } else {
                hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, (const uint4 *)src, (uint4 *)hw, win / 16);
            }
            hipStreamSynchronize(s);
            sink += hw[off % win];
        }
        const double dt = now() - t0;
        printf("%s: %.1f GB/s\n", mode == 0 ? "hipMemcpyAsync D2H        " : mode == 1 ? "kernel copy, 256 workgroups" : mode == 2 ? "kernel copy, 1024 workgroups" : "hipMemcpyAsync, non-coherent window", total / dt / 1e9);
    }
    return 0;
}
