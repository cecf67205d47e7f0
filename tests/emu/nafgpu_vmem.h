// tests/emu/nafgpu_vmem.h -- TEST INFRASTRUCTURE: plain C++ stand-in for
// nafcodec_amd/csrc/gfx950/nafgpu_vmem.h (the CPU harness has no VM counter).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>

namespace nafgpu {
struct vm_u32x4 {
    uint32_t x, y, z, w;
};
inline vm_u32x4 vm_load16(const uint8_t *p) {
    vm_u32x4 v;
    std::memcpy(&v, p, 16);
    return v;
}
inline void vm_store16(uint8_t *p, vm_u32x4 v) { std::memcpy(p, &v, 16); }
template <int N>
inline void vm_wait() {}
}  // namespace nafgpu
