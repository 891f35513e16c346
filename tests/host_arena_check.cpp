// CPU check of global-lvba_amd/csrc/host_arena.h (test infrastructure; compiled by tests/test_ordering.py::test_host_arena).
#include <cstdio>
#include <cstring>
#include <thread>
#include "../global-lvba_amd/csrc/host_arena.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main()
{
    using lvba::HostArena;
    HostArena &A = HostArena::get();
    CHECK(A.cached_bytes() == 0);
    // size classes: at most 12.5 % slack, monotone, idempotent
    for (size_t b = HostArena::kMin; b < ((size_t)1 << 30); b = b * 5 / 4 + 13) {
        const size_t c = HostArena::size_class(b);
        CHECK(c >= b && c <= b + b / 8 && HostArena::size_class(c) == c);
    }
    // small blocks go straight to malloc / free: nothing is cached
    void *s = A.alloc(1000);
    CHECK(s != nullptr);
    memset(s, 1, 1000);
    A.free(s, 1000);
    CHECK(A.cached_bytes() == 0);
    // a large block comes back from the free list of its class, and only of its class
    const size_t big = (size_t)3 << 20;
    void *p = A.alloc(big);
    CHECK(p != nullptr);
    memset(p, 2, big);
    A.free(p, big);
    CHECK(A.cached_bytes() == HostArena::size_class(big));
    void *q = A.alloc(big - 1000);                 // same class (within its slack)
    CHECK(q == p && A.cached_bytes() == 0);
    void *r = A.alloc(2 * big);                    // another class: a fresh block
    CHECK(r != nullptr && r != q);
    A.free(q, big - 1000);
    A.free(r, 2 * big);
    CHECK(A.cached_bytes() == HostArena::size_class(big) + HostArena::size_class(2 * big));
    CHECK(A.release() == HostArena::size_class(big) + HostArena::size_class(2 * big) && A.cached_bytes() == 0);
    // the vector type of the set-up tables: growth and destruction go through the arena
    {
        lvba::hvec<double> v;
        for (int i = 0; i < 300000; ++i) v.push_back(i);
        double sum = 0;
        for (double x : v) sum += x;
        CHECK(sum == 299999.0 * 300000.0 / 2.0);
        lvba::hvec<lvba::hvec<int>> vv(4, lvba::hvec<int>(50000, 7));
        CHECK(vv[3][49999] == 7);
    }
    CHECK(A.cached_bytes() > 0 && A.cached_bytes() <= A.cap());
    // concurrent use (the window stage's worker threads build tables at the same time)
    std::thread th[4];
    bool ok[4] = {false, false, false, false};
    for (int t = 0; t < 4; ++t)
        th[t] = std::thread([t, &ok] {
            bool good = true;
            for (int it = 0; it < 200; ++it) {
                lvba::hvec<int64_t> v((size_t)20000 + 1000 * t + it, t);
                good = good && v.front() == t && v.back() == t;
            }
            ok[t] = good;
        });
    for (auto &x : th) x.join();
    CHECK(ok[0] && ok[1] && ok[2] && ok[3]);
    A.release();
    CHECK(A.cached_bytes() == 0);
    // a miss in the request's own class takes the smallest cached block of up to twice the class, and the block goes back
    // under ITS class whatever size its user frees it with
    {
        const size_t c3 = HostArena::size_class((size_t)3 << 20), c9 = HostArena::size_class((size_t)9 << 20);
        void *a3 = A.alloc((size_t)3 << 20), *a9 = A.alloc((size_t)9 << 20);
        A.free(a3, (size_t)3 << 20);
        A.free(a9, (size_t)9 << 20);
        CHECK(A.cached_bytes() == c3 + c9);
        void *b = A.alloc((size_t)2 << 20);        // 2 MB class is empty: the 3 MB block (<= 2 x 2 MB), not the 9 MB one
        CHECK(b == a3 && A.cached_bytes() == c9);
        void *c = A.alloc((size_t)2 << 20);        // nothing within twice the class left: a fresh block
        CHECK(c != a9 && A.cached_bytes() == c9);
        A.free(b, (size_t)2 << 20);
        CHECK(A.cached_bytes() == c9 + c3);
        A.free(c, (size_t)2 << 20);
        A.release();
    }
    // the cap: nothing above it is kept (LVBA_HOST_CACHE_MB; 0 turns the cache off)
    {
        CHECK(A.cap() == HostArena::kDefaultCap);
        A.set_cap((size_t)4 << 20);
        void *a = A.alloc((size_t)3 << 20), *b = A.alloc((size_t)3 << 20);
        A.free(a, (size_t)3 << 20);
        A.free(b, (size_t)3 << 20);               // would exceed 4 MB: goes to free()
        CHECK(A.cached_bytes() == HostArena::size_class((size_t)3 << 20));
        A.set_cap(0);
        A.release();
        void *z = A.alloc((size_t)1 << 20);
        A.free(z, (size_t)1 << 20);
        CHECK(A.cached_bytes() == 0);
        A.set_cap(HostArena::kDefaultCap);
    }
    printf("host arena ok\n");
    return 0;
}
