// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A minimal single-threaded HIP stand-in so that the device code in nafcodec_amd/csrc can be
// compiled with g++ and run under AddressSanitizer / UBSan on a machine without a GPU (GPU
// sanitizers are not available on the MI355X pool).  Each workgroup is executed by fibers
// (one per work-item, switched at __syncthreads()), workgroups run one after another.
// The product library (libnafgpu.so) is built with hipcc and never sees this header; the
// build that uses it is tests/emu/libnafgpu_emu.so, loaded only by tests/test_emu_*.py.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

// ---- qualifiers ---------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(hipemu::dyn_shared());

struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 {
    uint32_t x, y;
};
struct alignas(16) uint4 {
    uint32_t x, y, z, w;
};
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

extern dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {
void *dyn_shared();
void sync_threads();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
}  // namespace hipemu

inline void __syncthreads() { hipemu::sync_threads(); }
int __syncthreads_or(int pred);
// wave-level primitives: the harness runs one-wave workgroups as a block of fibers
inline void __builtin_amdgcn_fence(int, const char *) {}
inline void __threadfence() {}
inline void __builtin_amdgcn_wave_barrier() { hipemu::sync_threads(); }
inline int __any(int pred) { return __syncthreads_or(pred); }
unsigned long long __ballot(int pred);
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

// ---- device intrinsics used by kernels.hip ------------------------------------------------
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz(static_cast<unsigned>(v)); }
template <class T>
inline T atomicCAS(T *p, T cmp, T val) {
    T old = *p;
    if (old == cmp) *p = val;
    return old;
}
template <class T>
inline T atomicMax(T *p, T v) {
    T old = *p;
    if (v > old) *p = v;
    return old;
}
template <class T>
inline T atomicSub(T *p, T v) {
    T old = *p;
    *p = old - v;
    return old;
}
template <class T>
inline T atomicOr(T *p, T v) {
    T old = *p;
    *p = old | v;
    return old;
}
template <class T>
inline T atomicAdd(T *p, T v) {
    T old = *p;
    *p = old + v;
    return old;
}
// v_alignbit_b32: ({hi, lo} >> (shift & 31)) truncated to 32 bits
inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t shift) {
    return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (shift & 31));
}
// v_perm_b32: byte select from {hi (bytes 4-7), lo (bytes 0-3)}
// v_bfe_u32: `width` bits of v from bit `offset`
inline float __frcp_rn(float x) { return 1.0f / x; }
inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline uint32_t __builtin_amdgcn_ubfe(uint32_t v, uint32_t offset, uint32_t width) {
    offset &= 31u;
    width &= 31u;
    return width ? (v >> offset) & ((1u << width) - 1u) : 0u;
}
inline uint32_t __builtin_amdgcn_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t src = (static_cast<uint64_t>(hi) << 32) | lo;
    uint32_t r = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t s = (sel >> (8 * k)) & 0xFF;
        uint32_t b = 0;
        if (s < 8) b = static_cast<uint32_t>((src >> (8 * s)) & 0xFF);
        else if (s == 0x0C) b = 0;
        else if (s >= 0x0D) b = 0xFF;
        r |= b << (8 * k);
    }
    return r;
}

// buffer descriptors (raw, stride 0): base + 32-bit offset, accesses outside [0, num_records) are
// dropped (stores) or return 0 (loads) -- the range check k_huf_decode uses to switch lanes off
#define NAFGPU_EMU 1
typedef uint32_t u32x4 __attribute__((vector_size(16)));
struct __amdgpu_buffer_rsrc_t {
    uint8_t *base;
    uint32_t n;
};
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *p, short, int n, int) {
    return __amdgpu_buffer_rsrc_t{static_cast<uint8_t *>(p), static_cast<uint32_t>(n)};
}
inline u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, int) {
    u32x4 v = {0, 0, 0, 0};
    const uint64_t o = static_cast<uint64_t>(voff) + soff;
    if (o + 16 <= r.n) std::memcpy(&v, r.base + o, 16);
    return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b128(u32x4 v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, int) {
    const uint64_t o = static_cast<uint64_t>(voff) + soff;
    if (o + 16 <= r.n) std::memcpy(r.base + o, &v, 16);
}
inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t v) { return v; }   // only applied to wave-uniform values

// ---- runtime API subset ----------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
typedef struct hipemuStream *hipStream_t;
typedef struct hipemuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

struct hipDeviceProp_t {
    char name[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    char gcnArchName[256];
};

hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes);
hipError_t hipStreamCreate(hipStream_t *s);
enum { hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // the harness runs launches in order
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);

template <class... P, class... A>
inline void hipLaunchKernelGGL(void (*kernel)(P...), dim3 grid, dim3 block, size_t shmem, hipStream_t, A... args) {
    hipemu::launch(grid, block, shmem, [&]() { kernel(static_cast<P>(args)...); });
}
