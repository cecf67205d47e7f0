// tools/duplex_probe.hip -- does this box move device-to-host and host-to-device bytes at the same time?  Pinned host buffers,
// copies by kernel (as the product's k_copy_out does both ways) and by hipMemcpyAsync (the runtime's copy engines): 8 GiB out
// alone, 2 GiB in alone, then both at once on two streams (the ratio of the record iterator: 40 GB out, 10 GB in), the two
// directions by the same means and by different ones.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_copy(const uint4 *__restrict__ src, uint4 *dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t win = size_t(64) << 20, out_total = size_t(8) << 30, in_total = size_t(2) << 30, dev_n = size_t(1) << 30;
    uint8_t *d_out = nullptr, *d_in = nullptr, *h_out[2] = {nullptr, nullptr}, *h_in[2] = {nullptr, nullptr};
    if (hipMalloc(&d_out, dev_n) != hipSuccess || hipMalloc(&d_in, dev_n) != hipSuccess) return 1;
    for (int k = 0; k < 2; k++)
        if (hipHostMalloc(&h_out[k], win) != hipSuccess || hipHostMalloc(&h_in[k], win) != hipSuccess) return 1;
    hipMemset(d_out, 1, dev_n);
    for (int k = 0; k < 2; k++)
        for (size_t i = 0; i < win; i += 4096) h_in[k][i] = 1;
    hipStream_t so, si;
    hipStreamCreateWithFlags(&so, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&si, hipStreamNonBlocking);
    auto run = [&](bool out, bool in, bool kernel_out, bool kernel_in) {
        const size_t n_out = out ? out_total / win : 0, n_in = in ? in_total / win : 0;
        hipDeviceSynchronize();
        const double t0 = now();
        double t_out = 0, t_in = 0;
        // windows of both directions enqueued in proportion, two buffers each way (a stream runs its copies in order)
        size_t io = 0, ii = 0;
        while (io < n_out || ii < n_in) {
            if (io < n_out) {
                uint8_t *src = d_out + (io * win) % dev_n;
                if (kernel_out) hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, so, (const uint4 *)src, (uint4 *)h_out[io & 1], win / 16);
                else hipMemcpyAsync(h_out[io & 1], src, win, hipMemcpyDeviceToHost, so);
                io++;
            }
            if (ii < n_in && (n_out == 0 || ii * n_out <= io * n_in)) {
                uint8_t *dst = d_in + (ii * win) % dev_n;
                if (kernel_in) hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, si, (const uint4 *)h_in[ii & 1], (uint4 *)dst, win / 16);
                else hipMemcpyAsync(dst, h_in[ii & 1], win, hipMemcpyHostToDevice, si);
                ii++;
            }
        }
        if (out) { hipStreamSynchronize(so); t_out = now() - t0; }
        if (in) { hipStreamSynchronize(si); t_in = now() - t0; }
        hipDeviceSynchronize();
        const double t = now() - t0;
        printf("out by %-7s in by %-7s %-12s", kernel_out ? "kernel" : "engines", kernel_in ? "kernel" : "engines", out && in ? "both at once" : out ? "out alone" : "in alone");
        if (out) printf("  out %5.1f GB/s (%.3f s)", out_total / t_out / 1e9, t_out);
        if (in) printf("  in %5.1f GB/s (%.3f s)", in_total / t_in / 1e9, t_in);
        if (out && in) printf("  together %5.1f GB/s over %.3f s", (out_total + in_total) / t / 1e9, t);
        printf("\n");
    };
    for (int rep = 0; rep < 2; rep++)
        for (int kernel = 1; kernel >= 0; kernel--) {
            run(true, false, kernel, kernel);
            run(false, true, kernel, kernel);
            run(true, true, kernel, kernel);
            run(true, true, kernel, !kernel);
        }
    return 0;
}
