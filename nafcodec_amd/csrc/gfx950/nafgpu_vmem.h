// nafgpu_vmem.h (gfx950) -- vector-memory helpers of k_huf_decode.
//
// On CDNA the VM counter retires loads AND stores in issue order.  k_huf_decode requests its next
// input pair, then issues the flush stores, then decodes, then needs the pair: the wait in front of
// that use must be `vmcnt(4)` (the two loads are back, the four stores may still be in flight) and
// not `vmcnt(0)`, or every round waits for its own stores to be acknowledged by a saturated HBM
// write path (21 ms vs 13 ms on the 10 GB config).  hipcc computes that count itself as long as the
// number of VM operations between the loads and their first use is the same on every path --
// which is why the flush issues its four stores unconditionally (rows with nothing to flush store
// to a sink).  These helpers are plain loads / stores; an inline-asm version is NOT safe here
// (the register allocator is free to reuse the destination registers of an asm load before its
// data has arrived).
#pragma once
#include <hip/hip_runtime.h>

namespace nafgpu {

typedef uint32_t vm_u32x4 __attribute__((ext_vector_type(4)));

__device__ inline vm_u32x4 vm_load16(const uint8_t *p) { return *reinterpret_cast<const vm_u32x4 *>(p); }

__device__ inline void vm_store16(uint8_t *p, vm_u32x4 v) { *reinterpret_cast<vm_u32x4 *>(p) = v; }

// documentation of the wait hipcc is expected to emit at this point (checked in the .s: `make asm`)
template <int N>
__device__ inline void vm_wait() {}

}  // namespace nafgpu
