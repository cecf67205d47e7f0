// tests/emu/hipemu.cpp -- TEST INFRASTRUCTURE: fiber-based executor behind tests/emu/hip/hip_runtime.h
#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <chrono>
#include <cstdio>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#define EMU_ASAN 1
#else
#define EMU_ASAN 0
#endif

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {
namespace {
constexpr size_t kStack = 256 << 10;
struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    int state = 0;  // 0 = runnable, 1 = at barrier, 2 = done
    dim3 tid;
    void *fake = nullptr;
};
std::vector<Fiber> g_fibers;
ucontext_t g_sched;
int g_cur = -1;
const std::function<void()> *g_body = nullptr;
alignas(64) unsigned char g_dyn[160 << 10];
const void *g_sched_bottom = nullptr;
size_t g_sched_size = 0;

void switch_to_fiber(Fiber &f) {
#if EMU_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, f.stack, kStack);
    swapcontext(&g_sched, &f.ctx);
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
    swapcontext(&g_sched, &f.ctx);
#endif
}

void switch_to_sched(Fiber &f, bool dying) {
#if EMU_ASAN
    __sanitizer_start_switch_fiber(dying ? nullptr : &f.fake, g_sched_bottom, g_sched_size);
    swapcontext(&f.ctx, &g_sched);
    __sanitizer_finish_switch_fiber(f.fake, &g_sched_bottom, &g_sched_size);
#else
    (void)dying;
    swapcontext(&f.ctx, &g_sched);
#endif
}

void trampoline() {
    Fiber &f = g_fibers[static_cast<size_t>(g_cur)];
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &g_sched_bottom, &g_sched_size);
#endif
    (*g_body)();
    f.state = 2;
    switch_to_sched(f, true);
}
}  // namespace

void *dyn_shared() { return g_dyn; }

void sync_threads() {
    Fiber &f = g_fibers[static_cast<size_t>(g_cur)];
    f.state = 1;
    switch_to_sched(f, false);
    threadIdx = f.tid;
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
    if (shmem > sizeof(g_dyn)) {
        std::fprintf(stderr, "hipemu: dynamic LDS request too large\n");
        std::abort();
    }
    const size_t nthreads = static_cast<size_t>(block.x) * block.y * block.z;
    if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
    for (size_t t = 0; t < nthreads; t++)
        if (!g_fibers[t].stack) g_fibers[t].stack = static_cast<char *>(std::malloc(kStack));
    gridDim = grid;
    blockDim = block;
    g_body = &body;
    for (uint32_t bz = 0; bz < grid.z; bz++)
        for (uint32_t by = 0; by < grid.y; by++)
            for (uint32_t bx = 0; bx < grid.x; bx++) {
                blockIdx = dim3(bx, by, bz);
                size_t t = 0;
                for (uint32_t tz = 0; tz < block.z; tz++)
                    for (uint32_t ty = 0; ty < block.y; ty++)
                        for (uint32_t tx = 0; tx < block.x; tx++, t++) {
                            Fiber &f = g_fibers[t];
                            f.state = 0;
                            f.tid = dim3(tx, ty, tz);
                            f.fake = nullptr;
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = f.stack;
                            f.ctx.uc_stack.ss_size = kStack;
                            f.ctx.uc_link = nullptr;
                            makecontext(&f.ctx, trampoline, 0);
                        }
                size_t done = 0;
                while (done < nthreads) {
                    done = 0;
                    for (size_t k = 0; k < nthreads; k++) {
                        Fiber &f = g_fibers[k];
                        if (f.state == 2) {
                            done++;
                            continue;
                        }
                        f.state = 0;
                        g_cur = static_cast<int>(k);
                        threadIdx = f.tid;
                        switch_to_fiber(f);
                        if (f.state == 2) done++;
                    }
                }
            }
    g_body = nullptr;
    g_cur = -1;
}
}  // namespace hipemu

// block-wide OR: accumulate, barrier, read, barrier (all work-items of the block must call it)
static int g_or_acc = 0, g_or_out = 0, g_or_count = 0;
int __syncthreads_or(int pred) {
    const int nthreads = static_cast<int>(blockDim.x * blockDim.y * blockDim.z);
    g_or_acc |= pred != 0;
    if (++g_or_count == nthreads) {
        g_or_out = g_or_acc;
        g_or_acc = 0;
        g_or_count = 0;
    }
    hipemu::sync_threads();
    const int r = g_or_out;
    hipemu::sync_threads();
    return r;
}

// wave ballot for one-wave (<= 64 work-item) workgroups: bit i = predicate of work-item i
static unsigned long long g_bal_acc = 0, g_bal_out = 0;
static int g_bal_count = 0;
unsigned long long __ballot(int pred) {
    const int nthreads = static_cast<int>(blockDim.x * blockDim.y * blockDim.z);
    if (pred) g_bal_acc |= 1ull << (threadIdx.x & 63);
    if (++g_bal_count == nthreads) {
        g_bal_out = g_bal_acc;
        g_bal_acc = 0;
        g_bal_count = 0;
    }
    hipemu::sync_threads();
    const unsigned long long r = g_bal_out;
    hipemu::sync_threads();
    return r;
}

// ---- runtime API ---------------------------------------------------------------------------
struct hipemuEvent {
    std::chrono::steady_clock::time_point t;
};
struct hipemuStream {
    int dummy;
};
static int g_device = 0;

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d != 0) return hipErrorInvalidValue; g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = g_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->name, "hipemu (CPU fibers, test only)");
    std::strcpy(p->gcnArchName, "emu");
    p->totalGlobalMem = size_t(8) << 30;
    p->multiProcessorCount = 1;
    return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n) {
    void *q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1) != 0) return hipErrorOutOfMemory;
    // device memory comes with whatever was there before (the product keeps small buffers from one decoder to the next):
    // nothing may count on zeros
    if (n <= (size_t(64) << 20)) std::memset(q, 0xA5, n);
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes) { *free_bytes = size_t(32) << 30; *total_bytes = size_t(32) << 30; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = new hipemuStream{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemuEvent{}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
