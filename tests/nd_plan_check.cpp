// CPU check of the nested-dissection plan (global-lvba_amd/csrc/nd_plan.h): on three graph shapes the plan is a permutation,
// no edge joins two different arcs, every arc -- separator edge is in the arc's separator list, the bandwidths it reports hold
// (arcs on their own edges; the separator on direct edges + the fill cliques), the owner assignment uses every rank, it is
// deterministic -- and the choice is the expected one (band for the folded ring of C3, chunks for the long ring on 8 ranks, hubs
// for the parking lot).     usage: nd_plan_check <shape: ring|long|lot> [N] [reach] [ranks]
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include "../global-lvba_amd/csrc/nd_plan.h"

int main(int argc, char **argv)
{
    const std::string shape = argc > 1 ? argv[1] : "lot";
    const int N = argc > 2 ? atoi(argv[2]) : 2000, W = argc > 3 ? atoi(argv[3]) : 50, ranks = argc > 4 ? atoi(argv[4]) : 1;
    lvba::hvec<uint8_t> adj((size_t)N * N, 0);
    uint64_t rng = 777;
    auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); };
    // voxels seen from ~5 poses within +-W of a home pose; 5 % also from the antipodal point (ring, long) or -- voxels homed in one
    // of eight short stretches that are all the same place -- from another crossing of that place (lot)
    const int n_lot = 8, Ls = W / 2;
    int lot[8];
    for (int q = 0; q < n_lot; ++q) lot[q] = (int)((q + 0.37 * (q % 3)) * N / n_lot) + W;
    const int V = 200 * N;
    for (int v = 0; v < V; ++v) {
        const int home = next() % N, k = 2 + next() % 7;
        int obs[16];
        int in_lot = -1;
        for (int q = 0; q < n_lot; ++q)
            if (home >= lot[q] && home < lot[q] + Ls) in_lot = q;
        const bool far = shape == "lot" ? (in_lot >= 0 && next() % 2 == 0) : (next() % 1000 < 50);
        for (int i = 0; i < k; ++i) {
            int p = home + (int)(next() % (2 * W + 1)) - W;
            if (far && i >= k / 2) {
                if (shape == "lot") p = lot[(in_lot + 1 + next() % (n_lot - 1)) % n_lot] + (int)(next() % Ls);
                else p += N / 2;
            }
            obs[i] = ((p % N) + N) % N;
        }
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < i; ++j)
                if (obs[i] != obs[j]) { adj[(size_t)obs[i] * N + obs[j]] = 1; adj[(size_t)obs[j] * N + obs[i]] = 1; }
    }
    lvba::hvec<int32_t> perm_band;
    lvba::rcm_order(adj, N, perm_band);
    lvba::hvec<int32_t> ipb((size_t)N);
    for (int i = 0; i < N; ++i) ipb[(size_t)perm_band[(size_t)i]] = i;
    int32_t Bb = 0;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            if (adj[(size_t)i * N + j]) Bb = std::max(Bb, std::abs(ipb[(size_t)i] - ipb[(size_t)j]));
    const auto t0 = std::chrono::steady_clock::now();
    const lvba::NdPlan pl = lvba::nd_plan(adj.data(), N, perm_band, Bb, ranks);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const lvba::NdPlan pl2 = lvba::nd_plan(adj.data(), N, perm_band, Bb, ranks);
    std::printf("shape=%s N=%d band_Bb=%d kind=%s arcs=%d Ns=%d BbS=%d t_band_ms=%.3f t_nd_ms=%.3f plan_ms=%.0f\n", shape.c_str(), N, Bb,
                pl.kind, (int)pl.arcs.size(), pl.Ns, pl.BbS, 1e3 * pl.t_band, 1e3 * pl.t_nd, ms);
    if (pl.active != pl2.active || pl.perm != pl2.perm) { std::printf("not deterministic\n"); return 2; }
    if (!pl.active) return shape == "ring" ? 0 : 10; // the folded ring keeps its band; the other two shapes must be dissected
    if (shape == "ring") { std::printf("the C3 ring should keep the band\n"); return 11; }
    // ---- structural checks
    lvba::hvec<int32_t> pos((size_t)N, -1);
    for (int i = 0; i < N; ++i) {
        const int v = pl.perm[(size_t)i];
        if (v < 0 || v >= N || pos[(size_t)v] >= 0) { std::printf("not a permutation\n"); return 1; }
        pos[(size_t)v] = i;
    }
    lvba::hvec<int32_t> arc_of((size_t)N, -1);
    int covered = 0;
    lvba::hvec<int> used((size_t)ranks, 0);
    for (size_t a = 0; a < pl.arcs.size(); ++a) {
        const auto &A = pl.arcs[a];
        if (A.p0 != covered) { std::printf("arcs not contiguous\n"); return 3; }
        covered += A.Na;
        for (int q = A.p0; q < A.p0 + A.Na; ++q) arc_of[(size_t)q] = (int)a;
        if (A.owner < 0 || A.owner >= ranks) return 4;
        used[(size_t)A.owner] = 1;
        std::printf("  arc %zu: p0=%d Na=%d Bb=%d sep=%zu owner=%d\n", a, A.p0, A.Na, A.Bb, A.sep.size(), A.owner);
    }
    if (covered != pl.ps || pl.ps + pl.Ns != N) { std::printf("ranges do not add up\n"); return 5; }
    if ((int)pl.arcs.size() >= ranks)
        for (int r = 0; r < ranks; ++r)
            if (!used[(size_t)r]) { std::printf("rank %d owns nothing\n", r); return 6; }
    lvba::hvec<uint8_t> fill((size_t)pl.Ns * pl.Ns, 0);
    for (const auto &A : pl.arcs)
        for (int x : A.sep)
            for (int y : A.sep) fill[(size_t)x * pl.Ns + y] = 1;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            if (!adj[(size_t)i * N + j] || i == j) continue;
            const int pi = pos[(size_t)i], pj = pos[(size_t)j];
            const int ai = arc_of[(size_t)pi], aj = arc_of[(size_t)pj];
            if (ai >= 0 && aj >= 0) {
                if (ai != aj) { std::printf("an edge joins arcs %d and %d\n", ai, aj); return 7; }
                if (std::abs(pi - pj) > pl.arcs[(size_t)ai].Bb) { std::printf("arc %d: edge longer than its bandwidth\n", ai); return 8; }
            } else if (ai >= 0 || aj >= 0) {
                const int a = ai >= 0 ? ai : aj, q = (ai >= 0 ? pj : pi) - pl.ps;
                const auto &sp = pl.arcs[(size_t)a].sep;
                if (!std::binary_search(sp.begin(), sp.end(), q)) { std::printf("arc %d misses separator pose %d\n", a, q); return 9; }
            } else
                fill[(size_t)(pi - pl.ps) * pl.Ns + (pj - pl.ps)] = 1;
        }
    for (int x = 0; x < pl.Ns; ++x)
        for (int y = 0; y < pl.Ns; ++y)
            if (fill[(size_t)x * pl.Ns + y] && std::abs(x - y) > pl.BbS) { std::printf("separator: coupling (%d, %d) outside its band %d\n", x, y, pl.BbS); return 12; }
    if (shape == "lot" && std::strcmp(pl.kind, "hubs")) { std::printf("expected hubs\n"); return 13; }
    if (shape == "long" && std::strcmp(pl.kind, "chunks")) { std::printf("expected chunks\n"); return 14; }
    std::printf("nd plan ok\n");
    return 0;
}
