// CPU check of global-lvba_amd/csrc/key_pack.h (test infrastructure; compiled by tests/test_ordering.py).
// The voxel map's root sort and the anchor down-sampling sort run on keys re-packed onto the bits that vary.  That is only a
// drop-in for the 63-bit sort if the re-packed keys ORDER exactly like the packed ones (then a stable sort gives the same
// permutation) and expand back to them.  Random boxes of every shape, both key widths, the degenerate ranges.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../global-lvba_amd/csrc/key_pack.h"

using lvba::KeyPack;
static int g_fail = 0;
#define CHECK(c, ...) do { if (!(c)) { if (g_fail < 20) { printf("FAILED line %d: %s  ", __LINE__, #c); printf(__VA_ARGS__); printf("\n"); } ++g_fail; } } while (0)

static uint64_t pack(const int k[3]) { return ((uint64_t)k[0] << 42) | ((uint64_t)k[1] << 21) | (uint64_t)k[2]; } // biased components

template <class K> static void run_box(std::mt19937_64 &rng, const int lo[3], const int hi[3], int n)
{
    std::vector<uint64_t> keys;
    int r[6] = {0, 0, 0, 0, 0, 0}; // the six maxima, as the kernels build them
    for (int i = 0; i < n; ++i) {
        int k[3];
        for (int j = 0; j < 3; ++j) {
            k[j] = lo[j] + (int)(rng() % (uint64_t)(hi[j] - lo[j] + 1));
            if (i < 2) k[j] = i ? hi[j] : lo[j]; // the corners are in
            r[j] = std::max(r[j], lvba::KEY_MAXC - k[j]);
            r[3 + j] = std::max(r[3 + j], k[j]);
        }
        keys.push_back(pack(k));
    }
    const KeyPack kp = lvba::key_pack_of(r);
    int total = 0;
    for (int j = 0; j < 3; ++j) {
        CHECK(kp.lo[j] == lo[j], "minimum %d: %d != %d", j, kp.lo[j], lo[j]);
        const unsigned span = (unsigned)(hi[j] - lo[j]);
        CHECK(span < (1u << kp.b[j]) || kp.b[j] == 32, "width %d too small", j);
        CHECK(kp.b[j] == 0 || span >= (1u << (kp.b[j] - 1)), "width %d not minimal", j);
        total += kp.b[j];
    }
    CHECK(kp.total == std::max(total, 1), "total");
    if (kp.total > (int)(8 * sizeof(K))) return; // the caller picks the 64-bit path then
    std::vector<K> c(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) {
        c[i] = lvba::key_compress<K>(keys[i], kp);
        CHECK(lvba::key_expand<K>(c[i], kp) == keys[i], "round trip of key %zu", i);
        CHECK(kp.total >= 64 || (uint64_t)c[i] < ((uint64_t)1 << kp.total), "key %zu outside its %d bits", i, kp.total);
    }
    for (size_t i = 0; i + 1 < keys.size(); ++i) { // random pairs: same order, same ties
        const size_t j = (size_t)(rng() % keys.size());
        CHECK((keys[i] < keys[j]) == (c[i] < c[j]) && (keys[i] == keys[j]) == (c[i] == c[j]), "order of keys %zu, %zu", i, j);
    }
}

int main()
{
    std::mt19937_64 rng(20250926);
    const int M = lvba::KEY_MAXC;
    int boxes = 0;
    for (int rep = 0; rep < 400; ++rep) {
        int lo[3], hi[3];
        for (int j = 0; j < 3; ++j) {
            const int w = (int)(rng() % 22);                          // 0 .. 21 bits of extent
            const int span = w == 0 ? 0 : (int)(rng() % ((uint64_t)1 << w));
            lo[j] = (int)(rng() % (uint64_t)(M - span + 1));
            hi[j] = lo[j] + span;
        }
        run_box<uint32_t>(rng, lo, hi, 300);
        run_box<uint64_t>(rng, lo, hi, 300);
        ++boxes;
    }
    { // the corners of the key space, a single voxel, a single column
        const int a0[3] = {0, 0, 0}, a1[3] = {M, M, M};
        run_box<uint64_t>(rng, a0, a1, 500);
        run_box<uint64_t>(rng, a0, a0, 5);
        run_box<uint32_t>(rng, a1, a1, 5);
        const int b0[3] = {1 << 20, (1 << 20) - 7, 0}, b1[3] = {1 << 20, (1 << 20) + 9, M};
        run_box<uint32_t>(rng, b0, b1, 300);
        // b[0] = 0 and b[1] + b[2] = 32: the x component's shift equals the width of the 32-bit key (computed in 64 bits)
        const int c0[3] = {77, 5, 0}, c1[3] = {77, 5 + (1 << 11) - 1, M};
        run_box<uint32_t>(rng, c0, c1, 300);
        run_box<uint64_t>(rng, c0, c1, 300);
    }
    if (g_fail) { printf("%d check(s) failed\n", g_fail); return 1; }
    printf("key pack ok (%d boxes)\n", boxes);
    return 0;
}
