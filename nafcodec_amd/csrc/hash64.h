// hash64.h -- order-sensitive 64-bit checksum used for full-size parity checks.
// H(buf) = sum over 4 KiB chunks c of mix64(c * K + sum_i (byte_i + 1) * (2 i + 1))   (mod 2^64)
// The per-chunk sums and the outer sum are commutative, so the GPU can compute it with one
// workgroup per chunk and a single atomic add, and the synthetic writer can compute the expected
// value block by block on the host.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace nafgpu {

constexpr uint64_t kHashChunk = 4096;

__host__ __device__ inline uint64_t hash_mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

__host__ __device__ inline uint64_t hash_chunk_final(uint64_t chunk_index, uint64_t weighted_sum) {
    return hash_mix64(chunk_index * 0x9E3779B97F4A7C15ull + weighted_sum);
}

inline uint64_t hash64_host(const uint8_t *p, uint64_t n, uint64_t first_chunk = 0) {
    uint64_t h = 0;
    for (uint64_t c = 0; c * kHashChunk < n; c++) {
        const uint64_t lo = c * kHashChunk, hi = lo + kHashChunk < n ? lo + kHashChunk : n;
        uint64_t s = 0;
        for (uint64_t i = lo; i < hi; i++) s += (uint64_t(p[i]) + 1) * (2 * (i - lo) + 1);
        h += hash_chunk_final(first_chunk + c, s);
    }
    return h;
}

}  // namespace nafgpu
