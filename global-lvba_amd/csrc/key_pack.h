// key_pack.h -- re-packing 3 x 21-bit voxel keys onto the bits that vary (voxel_internal.h: why; voxelize.hip, window_ba.hip: where).
// Plain integer arithmetic, host and device: tests/key_pack_check.cpp compiles this file with g++ and checks that the re-packed
// keys order exactly like the packed ones and expand back to them.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define LVBA_KP_HD __host__ __device__ __forceinline__
#else
#define LVBA_KP_HD inline
#endif

namespace lvba {

// The range as six maxima, so that it starts from a plain memset to zero: rng[j] = max(KEY_MAXC - x_j), rng[3 + j] = max(x_j),
// x = the biased key components (0 .. 2^21 - 1).
constexpr int KEY_MAXC = (1 << 21) - 1;
struct KeyPack {
    int lo[3];  // biased minima
    int b[3];   // bits per component
    int total;  // >= 1
};
inline KeyPack key_pack_of(const int rng[6])
{
    KeyPack kp;
    kp.total = 0;
    for (int j = 0; j < 3; ++j) {
        kp.lo[j] = KEY_MAXC - rng[j];
        const unsigned span = rng[3 + j] >= kp.lo[j] ? (unsigned)(rng[3 + j] - kp.lo[j]) : 0u;
        kp.b[j] = span ? 32 - __builtin_clz(span) : 0;
        kp.total += kp.b[j];
    }
    if (kp.total == 0) kp.total = 1;
    return kp;
}
// key = x << 42 | y << 21 | z (pack_key)  ->  (x - x0) << (by + bz) | (y - y0) << bz | (z - z0): the same lexicographic order
template <class K> LVBA_KP_HD K key_compress(uint64_t key, const KeyPack kp)
{
    const int x = (int)(key >> 42), y = (int)((key >> 21) & 0x1FFFFF), z = (int)(key & 0x1FFFFF);
    // in 64 bits, narrowed afterwards: b[1] + b[2] may equal the width of K (b[0] = 0, 32 key bits on the uint32 path), and a
    // shift by the full width of its type is undefined
    const uint64_t v = ((uint64_t)(x - kp.lo[0]) << (kp.b[1] + kp.b[2])) | ((uint64_t)(y - kp.lo[1]) << kp.b[2]) | (uint64_t)(z - kp.lo[2]);
    return (K)v;
}
template <class K> LVBA_KP_HD uint64_t key_expand(K c, const KeyPack kp)
{
    const uint64_t v = (uint64_t)c;
    const uint64_t x = (kp.b[1] + kp.b[2] < 64 ? v >> (kp.b[1] + kp.b[2]) : 0) + (uint64_t)kp.lo[0];
    const uint64_t y = ((v >> kp.b[2]) & (((uint64_t)1 << kp.b[1]) - 1)) + (uint64_t)kp.lo[1];
    const uint64_t z = (v & (((uint64_t)1 << kp.b[2]) - 1)) + (uint64_t)kp.lo[2];
    return (x << 42) | (y << 21) | z;
}

} // namespace lvba
