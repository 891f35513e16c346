// CPU check of global-lvba_amd/csrc/host_tables.h: chunking of the voxel-major kernels and work items of the pair pass.
#include <cstdio>
#include <cstdlib>
#include "../global-lvba_amd/csrc/host_tables.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static int check_chunks(const lvba::hvec<int64_t> &k, int CF, int CV)
{
    lvba::hvec<int64_t> off(k.size() + 1, 0), chunk;
    for (size_t a = 0; a < k.size(); ++a) off[a + 1] = off[a] + k[a];
    int64_t Q = -1, Qw = 0;
    for (auto v : k) Qw += v * (v - 1) / 2;
    CHECK(lvba::chunk_voxels((int64_t)k.size(), off.data(), CF, CV, chunk, Q) == -1);
    CHECK(Q == Qw && chunk.front() == 0 && chunk.back() == (int64_t)k.size());
    for (size_t c = 0; c + 1 < chunk.size(); ++c) {
        const int64_t v0 = chunk[c], v1 = chunk[c + 1];
        CHECK(v1 > v0 || k.empty());
        const int64_t nf = off[v1] - off[v0];
        if (v1 - v0 == 1 && k[v0] > CF) continue;                    // a big voxel alone
        CHECK(nf <= CF && v1 - v0 <= CV);
        for (int64_t a = v0; a < v1; ++a) CHECK(k[a] <= CF);
        // greedy: the next voxel would not have fitted
        if (v1 < (int64_t)k.size() && k[v1] <= CF) CHECK(nf + k[v1] > CF || v1 - v0 == CV);
    }
    return 0;
}

int main()
{
    uint64_t rng = 99;
    auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); };
    for (int rep = 0; rep < 200; ++rep) {
        lvba::hvec<int64_t> k(1 + next() % 400);
        for (auto &v : k) v = 2 + next() % 7;
        if (rep % 3 == 0) k[next() % k.size()] = 257 + next() % 900;   // a voxel with more observers than lanes
        if (rep % 5 == 0) k[0] = 256;
        if (rep % 7 == 0) k.back() = 300;
        if (check_chunks(k, 256, 128)) return 1;
    }
    if (check_chunks({}, 256, 128)) return 1;
    if (check_chunks(lvba::hvec<int64_t>(1000, 2), 256, 128)) return 1;  // the voxel limit binds (128 voxels x 2 factors)
    if (check_chunks(lvba::hvec<int64_t>(10, 256), 256, 128)) return 1;
    { // a voxel with a single factor is rejected, and reported
        lvba::hvec<int64_t> off{0, 3, 4, 8}, chunk; int64_t Q;
        CHECK(lvba::chunk_voxels(3, off.data(), 256, 128, chunk, Q) == 1);
    }
    for (int rep = 0; rep < 200; ++rep) { // group boundaries (lvba_balm_set_groups): no chunk straddles a break, every break starts a chunk
        lvba::hvec<int64_t> k(20 + next() % 600);
        for (auto &v : k) v = 2 + next() % 9;
        if (rep % 4 == 0) k[next() % k.size()] = 300 + next() % 500;
        lvba::hvec<int64_t> off(k.size() + 1, 0), chunk, brk;
        for (size_t a = 0; a < k.size(); ++a) off[a + 1] = off[a] + k[a];
        for (int64_t v = 1 + next() % 40; v < (int64_t)k.size(); v += 1 + next() % 90) brk.push_back(v);
        int64_t Q = -1, Qw = 0;
        for (auto v : k) Qw += v * (v - 1) / 2;
        CHECK(lvba::chunk_voxels((int64_t)k.size(), off.data(), 256, 128, chunk, Q, brk.data(), (int64_t)brk.size()) == -1);
        CHECK(Q == Qw && chunk.front() == 0 && chunk.back() == (int64_t)k.size());
        for (size_t c = 0; c + 1 < chunk.size(); ++c) {
            CHECK(chunk[c + 1] > chunk[c]);
            const int64_t nf = off[chunk[c + 1]] - off[chunk[c]];
            CHECK((chunk[c + 1] - chunk[c] == 1 && k[chunk[c]] > 256) || (nf <= 256 && chunk[c + 1] - chunk[c] <= 128));
            for (int64_t b : brk) CHECK(!(chunk[c] < b && b < chunk[c + 1]));       // no chunk straddles a break
        }
        for (int64_t b : brk) { // every break is a chunk start
            bool found = false;
            for (int64_t c : chunk) found = found || c == b;
            CHECK(found);
        }
        // without breaks the table is the plain one
        lvba::hvec<int64_t> plain, none;
        int64_t Q2;
        CHECK(lvba::chunk_voxels((int64_t)k.size(), off.data(), 256, 128, plain, Q2) == -1);
        CHECK(lvba::chunk_voxels((int64_t)k.size(), off.data(), 256, 128, none, Q2, brk.data(), 0) == -1);
        CHECK(plain == none);
    }
    for (int rep = 0; rep < 200; ++rep) { // pair work items
        const int nb = 1 + next() % 300;
        lvba::hvec<int64_t> slot(nb), off(nb + 1, 0);
        for (int b = 0; b < nb; ++b) { slot[b] = 10 * b + 3; off[b + 1] = off[b] + 1 + (rep % 2 ? next() % 90 : next() % 5000); }
        const int64_t Q = off.back(), cut = lvba::pair_cut_length(Q);
        CHECK(cut >= 64 && cut <= 512 && cut % 16 == 0);
        lvba::hvec<int64_t> io, id, mo, ms; int64_t np;
        lvba::cut_pair_items(slot, off, Q, io, id, mo, ms, np);
        CHECK(io.front() == 0 && io.back() == Q && io.size() == id.size() + 1 && mo.size() == ms.size() + 1 && mo.back() == np);
        size_t it = 0, m = 0;
        for (int b = 0; b < nb; ++b) {
            const int64_t len = off[b + 1] - off[b];
            if (len <= cut) {
                CHECK(id[it] == slot[b] && io[it] == off[b] && io[it + 1] == off[b + 1]);
                ++it;
            } else {
                CHECK(ms[m] == slot[b]);
                int64_t q = off[b];
                for (int64_t pi = mo[m]; pi < mo[m + 1]; ++pi, ++it) {
                    CHECK(id[it] == -(1 + pi) && io[it] == q && io[it + 1] - io[it] <= cut && io[it + 1] > io[it]);
                    q = io[it + 1];
                }
                CHECK(q == off[b + 1]);
                ++m;
            }
        }
        CHECK(it == id.size() && m == ms.size());
    }
    for (int rep = 0; rep < 200; ++rep) { // grouped items: repeating slots (one run per voxel window) and a small cut
        const int nslots = 1 + next() % 60, nruns = 1 + next() % 400;
        const int64_t cut = 1 + next() % 40;
        lvba::hvec<int64_t> slot(nruns), off(nruns + 1, 0);
        for (int r = 0; r < nruns; ++r) { slot[r] = next() % nslots; off[r + 1] = off[r] + 1 + next() % 100; }
        lvba::hvec<int64_t> io, id, mo, ms, mi; int64_t np;
        lvba::group_pair_items(slot, off, cut, nslots, io, id, mo, ms, mi, np);
        CHECK(io.front() == 0 && io.back() == off.back() && io.size() == id.size() + 1 && mo.size() == ms.size() + 1);
        CHECK((int64_t)mi.size() == mo.back() && (int64_t)mi.size() == np);
        // every item lies inside one run, is at most `cut` long, and the items tile the pair array
        lvba::hvec<int64_t> item_slot(id.size());
        size_t r = 0;
        for (size_t i = 0; i < id.size(); ++i) {
            CHECK(io[i + 1] > io[i] && io[i + 1] - io[i] <= cut);
            while (io[i] >= off[r + 1]) ++r;
            CHECK(io[i + 1] <= off[r + 1]);
            item_slot[i] = slot[r];
        }
        // direct items: their slot occurs once; partial indices follow the item order; every summed block lists exactly its
        // items' partials, in item order
        lvba::hvec<int> cnt(nslots, 0);
        for (auto sl : item_slot) cnt[sl]++;
        int64_t next_partial = 0;
        lvba::hvec<lvba::hvec<int64_t>> want(nslots);
        for (size_t i = 0; i < id.size(); ++i) {
            if (cnt[item_slot[i]] == 1) CHECK(id[i] == item_slot[i]);
            else { CHECK(id[i] == -(1 + next_partial)); want[item_slot[i]].push_back(next_partial++); }
        }
        CHECK(next_partial == np);
        size_t m = 0;
        for (int sl = 0; sl < nslots; ++sl) {
            if (cnt[sl] <= 1) continue;
            CHECK(m < ms.size() && ms[m] == sl && mo[m + 1] - mo[m] == (int64_t)want[sl].size());
            for (size_t q = 0; q < want[sl].size(); ++q) CHECK(mi[mo[m] + q] == want[sl][q]);
            ++m;
        }
        CHECK(m == ms.size());
    }
    std::printf("host tables ok\n");
    return 0;
}
