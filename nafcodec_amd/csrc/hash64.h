// hash64.h -- order-sensitive 64-bit checksum used for full-size parity checks.
//   H(buf) = sum over 8-byte little-endian words w_j of mix64(w_j ^ (j + 1) * K)      (mod 2^64)
// (the last word zero-padded; j counted from the start of the whole object).  Every word goes
// through a non-linear bijection keyed by its position before it is added, so two byte errors
// cannot cancel and swapped or shifted data changes the value; the outer sum is commutative, so
// the GPU computes it with any work split and one atomic add per workgroup, the synthetic writer
// block by block on the host, and the values of shards that start on a 4 KiB chunk boundary
// (first_chunk) add up to the value of the whole object.  Lengths are compared separately.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace nafgpu {

constexpr uint64_t kHashChunk = 4096;                  // granularity of `first_chunk`
constexpr uint64_t kHashKey = 0x9E3779B97F4A7C15ull;

__host__ __device__ inline uint64_t hash_mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

// contribution of word number `index` (0-based, in the whole object)
__host__ __device__ inline uint64_t hash_word(uint64_t word, uint64_t index) {
    return hash_mix64(word ^ ((index + 1) * kHashKey));
}

inline uint64_t hash64_host(const uint8_t *p, uint64_t n, uint64_t first_chunk = 0) {
    uint64_t h = 0;
    const uint64_t w0 = first_chunk * (kHashChunk / 8), n_full = n / 8;
    for (uint64_t j = 0; j < n_full; j++) {
        uint64_t w;
        std::memcpy(&w, p + 8 * j, 8);
        h += hash_word(w, w0 + j);
    }
    if (n & 7) {
        uint64_t w = 0;
        std::memcpy(&w, p + 8 * n_full, n & 7);
        h += hash_word(w, w0 + n_full);
    }
    return h;
}

}  // namespace nafgpu
