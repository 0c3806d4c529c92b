// philox.hpp -- Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), the counter-based generator
// behind every device-side draw of this library: one block of four 32-bit words per (key, counter) pair, no state.
#pragma once
#include <stdint.h>

__device__ __forceinline__ void philox4(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&o)[4])
{
    uint32_t k0 = uint32_t(seed), k1 = uint32_t(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = uint64_t(0xD2511F53u) * c0, p1 = uint64_t(0xCD9E8D57u) * c2;
        const uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0, n2 = uint32_t(p0 >> 32) ^ c3 ^ k1;
        c1 = uint32_t(p1); c3 = uint32_t(p0); c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
