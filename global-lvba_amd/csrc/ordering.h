// ordering.h -- pose ordering for the band solver (host code, header-only, no HIP: unit-tested on the CPU by
// tests/ordering_check.cpp).
#pragma once
#include "host_arena.h"
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <utility>
#include <vector>

namespace lvba {

// Pose ordering for the band solver: reverse Cuthill-McKee on the pose co-visibility graph (byte adjacency
// matrix adj[N*N]) from two start rules (pseudo-peripheral node, minimum-degree node), then a barycenter
// refinement: positions are repeatedly replaced by the mean position of the neighbours and re-ranked, which
// interleaves the two sides of ring-like trajectories (loop closures).  The candidate with the smallest
// pose-block bandwidth wins.  One-off host work at finalize().
inline int32_t bandwidth_of(const lvba::hvec<lvba::hvec<int32_t>> &nb, const lvba::hvec<int32_t> &perm)
{
    const int N = (int)perm.size();
    lvba::hvec<int32_t> ip(N);
    for (int i = 0; i < N; ++i) ip[perm[i]] = i;
    int32_t bw = 0;
    for (int i = 0; i < N; ++i)
        for (int j : nb[i]) bw = std::max(bw, std::abs(ip[i] - ip[j]));
    return bw;
}

inline void rcm_from(const lvba::hvec<lvba::hvec<int32_t>> &nb, const lvba::hvec<int32_t> &deg, bool peripheral,
                     lvba::hvec<int32_t> &perm)
{
    const int N = (int)nb.size();
    lvba::hvec<char> seen(N, 0), mark(N, 0);
    lvba::hvec<int32_t> order, level(N);
    order.reserve(N);
    auto bfs_far = [&](int start) { // farthest node (minimal degree among the last level) from start
        std::queue<int> q;
        lvba::hvec<int> touched;
        q.push(start); mark[start] = 1; touched.push_back(start); level[start] = 0;
        int last = start;
        while (!q.empty()) {
            int a = q.front(); q.pop();
            if (level[a] > level[last] || (level[a] == level[last] && deg[a] < deg[last])) last = a;
            for (int b : nb[a])
                if (!mark[b] && !seen[b]) { mark[b] = 1; level[b] = level[a] + 1; touched.push_back(b); q.push(b); }
        }
        for (int t : touched) mark[t] = 0;
        return last;
    };
    lvba::hvec<int32_t> by_deg(N);
    for (int i = 0; i < N; ++i) by_deg[i] = i;
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return deg[a] < deg[b]; });
    for (int root : by_deg) {
        if (seen[root]) continue;
        int s = root;
        if (peripheral)
            for (int pass = 0; pass < 3; ++pass) s = bfs_far(s);
        std::queue<int> q;
        q.push(s); seen[s] = 1;
        while (!q.empty()) {
            int a = q.front(); q.pop();
            order.push_back(a);
            for (int b : nb[a])
                if (!seen[b]) { seen[b] = 1; q.push(b); }
        }
    }
    perm.assign(order.rbegin(), order.rend());
}

// Local search on top of RCM + barycenter: pick one of the longest edges, slide one of its endpoints a few places towards
// the other (the nodes in between shift by one), keep the move if (bandwidth, number of edges within 8 of it) did not get
// worse.  Edge lengths live in a histogram, so a move costs O(degree x nodes moved).  Deterministic (own LCG): every rank
// of a multi-GPU job derives the same order from the same (all-reduced) graph.  On the C3 co-visibility graph (ring
// trajectory, +-50 pose band, antipodal loop closures) 10 N moves take the half-bandwidth from ~470 to ~435 pose blocks
// (~15 % fewer factorisation flops) for ~0.1 s of one-off host time; 8x more moves reach ~420 but cost 2.5 s, more than
// a whole refinement saves.
inline void hill_climb(const lvba::hvec<lvba::hvec<int32_t>> &nb, lvba::hvec<int32_t> &perm, int32_t &bw_io)
{
    const int N = (int)perm.size();
    if (N < 64) return;
    lvba::hvec<int32_t> pos(N), order(perm);
    for (int i = 0; i < N; ++i) pos[order[i]] = i;
    lvba::hvec<int64_t> hist((size_t)N + 1, 0);
    for (int i = 0; i < N; ++i)
        for (int j : nb[i])
            if (j > i) hist[std::abs(pos[i] - pos[j])]++;
    int32_t cur = N;
    while (cur > 0 && hist[cur] == 0) --cur;
    constexpr int32_t SLACK = 8, SMAX = 8;
    auto near_max = [&](int32_t m) { int64_t c = 0; for (int32_t l = std::max(1, m - SLACK); l <= m; ++l) c += hist[l]; return c; };
    uint64_t rng = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); };
    // candidate list of long edges, rebuilt lazily
    lvba::hvec<std::pair<int32_t, int32_t>> longe;
    auto rebuild = [&]() {
        longe.clear();
        for (int i = 0; i < N; ++i)
            for (int j : nb[i])
                if (j > i && std::abs(pos[i] - pos[j]) >= cur - SLACK) longe.emplace_back(i, j);
    };
    rebuild();
    const int64_t iters = std::min<int64_t>(100000, 10 * (int64_t)N); // fixed budget: every rank must derive the same order
    lvba::hvec<int32_t> seg, old;
    int64_t since_rebuild = 0;
    int64_t stale = 0;
    for (int64_t it = 0; it < iters && cur > 1 && stale < 5000; ++it, ++stale) {
        if (longe.empty() || ++since_rebuild > 2000) { rebuild(); since_rebuild = 0; if (longe.empty()) break; }
        const auto e = longe[next() % longe.size()];
        int32_t a = e.first, b = e.second;
        if (std::abs(pos[a] - pos[b]) < cur - SLACK) continue; // stale entry
        if (pos[a] > pos[b]) std::swap(a, b);
        const bool move_a = next() & 1;
        const int32_t node = move_a ? a : b;
        const int32_t s = 1 + (int32_t)(next() % SMAX);
        const int32_t p0 = pos[node];
        const int32_t p1 = std::min(std::max(p0 + (move_a ? s : -s), 0), N - 1);
        if (p0 == p1) continue;
        const int32_t lo = std::min(p0, p1), hi = std::max(p0, p1);
        seg.assign(order.begin() + lo, order.begin() + hi + 1);
        old.resize(seg.size());
        const int64_t before_near = near_max(cur);
        const int32_t before = cur;
        // remove the lengths of all edges incident to the moved nodes, move, add the new lengths
        auto touch = [&](int sign) {
            for (int32_t v : seg)
                for (int j : nb[v]) {
                    // an edge between two moved nodes is visited twice: count it once (from the smaller id)
                    const bool both = pos[j] >= lo && pos[j] <= hi;
                    if (both && j < v) continue;
                    hist[std::abs(pos[v] - pos[j])] += sign;
                }
        };
        touch(-1);
        for (size_t q = 0; q < seg.size(); ++q) old[q] = pos[seg[q]];
        if (p0 < p1) { for (int32_t q = lo; q < hi; ++q) order[q] = seg[q - lo + 1]; order[hi] = seg[0]; }
        else { for (int32_t q = lo + 1; q <= hi; ++q) order[q] = seg[q - lo - 1]; order[lo] = seg.back(); }
        for (int32_t q = lo; q <= hi; ++q) pos[order[q]] = q;
        touch(+1);
        int32_t now = std::max(cur, hi - lo + cur); // upper bound, then walk down
        if (now > N) now = N;
        while (now > 0 && hist[now] == 0) --now;
        const int64_t now_near = near_max(now);
        const bool better = now < before || (now == before && now_near <= before_near);
        if (better) {
            if (now < before || now_near < before_near) stale = 0;
            cur = now;
        } else { // undo
            touch(-1);
            for (size_t q = 0; q < seg.size(); ++q) { order[lo + q] = seg[q]; pos[seg[q]] = old[q]; }
            touch(+1);
        }
    }
    if (cur < bw_io) { perm = order; bw_io = cur; }
}

// the ordering from neighbour lists (nb is re-sorted by degree).  quick: RCM + barycenter sweeps only (nd_plan.h weighs many
// candidate partitions; the one it keeps is ordered in full)
inline int32_t rcm_order_nb(lvba::hvec<lvba::hvec<int32_t>> &nb, lvba::hvec<int32_t> &perm, bool quick = false)
{
    const int N = (int)nb.size();
    lvba::hvec<int32_t> deg(N, 0);
    for (int i = 0; i < N; ++i) deg[i] = (int32_t)nb[i].size();
    for (int i = 0; i < N; ++i)
        std::sort(nb[i].begin(), nb[i].end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
    lvba::hvec<int32_t> best, cand;
    int32_t best_bw = INT32_MAX;
    for (int variant = 0; variant < 2; ++variant) {
        rcm_from(nb, deg, variant == 0, cand);
        const int32_t bw = bandwidth_of(nb, cand);
        if (bw < best_bw) { best_bw = bw; best = cand; }
    }
    // barycenter refinement of the best candidate
    lvba::hvec<double> x(N), y(N);
    for (int i = 0; i < N; ++i) x[best[i]] = i;
    lvba::hvec<int32_t> idx(N);
    for (int it = 1; it <= (quick ? 10 : 40); ++it) {
        for (int i = 0; i < N; ++i) {
            if (nb[i].empty()) { y[i] = x[i]; continue; }
            double s = 0.0;
            for (int j : nb[i]) s += x[j];
            y[i] = s / (double)nb[i].size();
        }
        x.swap(y);
        if (it % 5 == 0) {
            for (int i = 0; i < N; ++i) idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return x[a] < x[b]; });
            const int32_t bw = bandwidth_of(nb, idx);
            if (bw < best_bw) { best_bw = bw; best = idx; }
            for (int i = 0; i < N; ++i) x[idx[i]] = i; // re-rank so the positions do not collapse
        }
    }
    perm = best;
    if (!quick) hill_climb(nb, perm, best_bw);
    return best_bw;
}


inline void rcm_order(const lvba::hvec<uint8_t> &adj, int N, lvba::hvec<int32_t> &perm)
{
    lvba::hvec<lvba::hvec<int32_t>> nb(N);
    for (int i = 0; i < N; ++i) { // the matrix is sparse (C3: 13 % non-zero): skip it eight bytes at a time
        const uint8_t *row = adj.data() + (size_t)i * N;
        int j = 0;
        for (; j + 8 <= N; j += 8) {
            uint64_t w;
            std::memcpy(&w, row + j, 8);
            if (!w) continue;
            for (int e = 0; e < 8; ++e)
                if (row[j + e] && j + e != i) nb[i].push_back(j + e);
        }
        for (; j < N; ++j)
            if (row[j] && j != i) nb[i].push_back(j);
    }
    rcm_order_nb(nb, perm);
}

} // namespace lvba
