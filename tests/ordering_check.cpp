// CPU check of the pose ordering (global-lvba_amd/csrc/ordering.h): permutation, bandwidth against the natural order and
// plain RCM, determinism, timing.  usage: ordering_check [N] [band] [loop_frac_permille]
#include <chrono>
#include <cstdio>
#include <cstring>
#include <numeric>
#include "../global-lvba_amd/csrc/ordering.h"

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 2000, W = argc > 2 ? atoi(argv[2]) : 50, loop = argc > 3 ? atoi(argv[3]) : 50;
    // co-visibility of a closed trajectory: every voxel is seen by ~5 poses within +-W of a home pose (ring distance), and a
    // share of the voxels additionally by poses around the antipodal point (loop closures) -- the structure of bench config C3
    lvba::hvec<uint8_t> adj((size_t)N * N, 0);
    uint64_t rng = 12345;
    auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); };
    const int V = 200 * N;
    for (int v = 0; v < V; ++v) {
        const int home = next() % N, k = 2 + next() % 7;
        int obs[16];
        for (int i = 0; i < k; ++i) {
            int off = (int)(next() % (2 * W + 1)) - W;
            if ((int)(next() % 1000) < loop && i == k - 1) off += N / 2;
            obs[i] = ((home + off) % N + N) % N;
        }
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < i; ++j)
                if (obs[i] != obs[j]) { adj[(size_t)obs[i] * N + obs[j]] = 1; adj[(size_t)obs[j] * N + obs[i]] = 1; }
    }
    lvba::hvec<lvba::hvec<int32_t>> nb(N);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            if (adj[(size_t)i * N + j] && i != j) nb[i].push_back(j);
    lvba::hvec<int32_t> nat(N);
    std::iota(nat.begin(), nat.end(), 0);
    const int32_t bw_nat = lvba::bandwidth_of(nb, nat);
    lvba::hvec<int32_t> p1, p2;
    const auto t0 = std::chrono::steady_clock::now();
    lvba::rcm_order(adj, N, p1);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    lvba::rcm_order(adj, N, p2);
    lvba::hvec<char> seen(N, 0);
    for (int v : p1) { if (v < 0 || v >= N || seen[v]) { std::printf("not a permutation\n"); return 1; } seen[v] = 1; }
    if (p1 != p2) { std::printf("not deterministic\n"); return 2; }
    const int32_t bw = lvba::bandwidth_of(nb, p1);
    // plain RCM for comparison
    lvba::hvec<int32_t> deg(N), rcm;
    for (int i = 0; i < N; ++i) deg[i] = (int32_t)nb[i].size();
    lvba::rcm_from(nb, deg, true, rcm);
    const int32_t bw_rcm = lvba::bandwidth_of(nb, rcm);
    std::printf("N=%d natural=%d rcm=%d ordered=%d time_ms=%.1f\n", N, bw_nat, bw_rcm, bw, ms);
    if (bw > bw_rcm || bw > bw_nat) return 3;
    return 0;
}
