// nd_plan.h -- one level of nested dissection for pose graphs the band solver does not fit (host code, header-only, no HIP:
// unit-tested on the CPU by tests/nd_plan_check.cpp).  The solver that runs the plan: ldlt_nd.h.
//
// Input: the pose co-visibility graph (byte adjacency matrix, the same on every rank of a multi-GPU job), the band ordering
// ordering.h found for it and the number of ranks.  Output: either "keep the band" or a partition
//      poses = arc_0 | arc_1 | ... | arc_{P-1} | separator
// in which the arcs are the connected components of the graph without the separator poses (each with a band ordering of its
// own, usually much narrower than the whole graph's) and the separator poses are ordered for the band solver on THEIR graph
// (direct edges + the fill through every arc: the separator poses an arc touches become a clique).
// Two kinds of separator are tried, a cost model in the solver's own units (launches on the serial chain x microseconds per
// launch, flops / sustained rate) picks among them and the band:
//   hubs    poses whose degree is well above the median: a place crossed many times is a clique on top of the ring, and the
//           clique's poses see every crossing's neighbourhood -- taking them out leaves the ring's arcs;
//   chunks  K runs of consecutive poses of the BAND ordering, each as wide as the band: a band is a path, a chunk of width
//           >= bandwidth cuts it.  On a folded ring the pieces between two chunks fall apart into the two sides of the fold, each
//           with half the bandwidth: K chunks give 2 K arcs.  This is what lets G ranks share a long band (C4: n / bw = 24).
#pragma once
#include "host_arena.h"
#include "ordering.h"
#include <algorithm>
#include <cstdint>
#include <vector>

namespace lvba {

struct NdPlanArc {
    int32_t p0 = 0, Na = 0, Bb = 0, owner = 0;
    lvba::hvec<int32_t> sep; // separator-local indices (0 .. Ns-1, in the separator's own order) of the poses the arc touches, ascending
};
struct NdPlan {
    bool active = false;
    int32_t ps = 0, Ns = 0, BbS = 0; // the separator: solver positions [ps, ps + Ns), half-bandwidth of its system in pose blocks
    lvba::hvec<NdPlanArc> arcs;
    lvba::hvec<int32_t> perm;        // perm[solver position] = caller's pose
    const char *kind = "band";
    double t_band = 0.0, t_nd = 0.0; // the model's estimates, seconds per solve
};

// ---- the cost model: seconds per solve (measured constants of rounds 4 / 5: a chain-bound look-ahead launch is ~27 us with one
// problem in it and ~32 with the two ends of a band, wide bands sustain ~36 TFLOP/s of fp64 MFMA (n bw^2 flops: 9.3 ms at
// n = 12 000, bw = 5 243; 88 ms at n = 60 000, bw = 7 229), a forward-substitution launch of ldlt_nd.h ~12 us)
inline double nd_fact_seconds(double n, double bw, bool may_twist)
{
    const double t_launch = may_twist ? 32e-6 : 27e-6, rate = 36e12; // (two problems per launch: 32 - 36 us; one: 23 - 27 us)
    if (bw + 128.0 >= 0.6 * n) return std::max(n * n * n / 3.0 / rate, n / 64.0 * t_launch); // dense
    double chain = n / 64.0;
    const double P = std::floor((n - bw) / 128.0);
    if (may_twist && P >= 4) chain = P + (n - 128.0 * P) / 64.0;
    return std::max(n * bw * bw / rate, chain * t_launch);
}
// one arc alone on the device: the factorisation's chain with the border's forward substitution riding in its launches (two
// panels behind: ldlt_lookahead.h, FwdPassenger), then Y^T D^-1 Y.  work_out: the same as pure throughput -- what several arcs of
// one rank, side by side, add up to.  (measured, round 5: a step launch with its passengers 26 us; Y^T D^-1 Y 0.73 ms at
// n = 5 100, s = 1 818 = 23 TFLOP/s; whole solves 5.0 ms / 24.2 ms at 2 000 / 10 000 poses against 4.2 / 19.8 from this model)
inline double nd_arc_seconds(double n, double bw, double s, double *work_out)
{
    const double t_launch = 28e-6, rate = 36e12, rate_fwd = 20e12, rate_schur = 23e12;
    const double w = std::min(bw, n);
    const double f_fact = n * w * w / rate, f_fwd = 2.0 * n * w * s / rate_fwd, f_schur = n * s * s / rate_schur;
    *work_out = f_fact + f_fwd + f_schur;
    return std::max(f_fact + f_fwd, n / 64.0 * t_launch) + f_schur;
}

namespace nd_detail {
inline void neighbours(const uint8_t *adj, int N, lvba::hvec<lvba::hvec<int32_t>> &nb)
{
    nb.assign((size_t)N, lvba::hvec<int32_t>());
    for (int i = 0; i < N; ++i) {
        const uint8_t *row = adj + (size_t)i * N;
        for (int j = 0; j < N; ++j)
            if (row[j] && j != i) nb[(size_t)i].push_back(j);
    }
}
// band ordering of the sub-graph on `nodes` (caller indices) from the neighbour lists (+ `extra` lists, local indices):
// order_out = the nodes in solver order, returns its half-bandwidth.  local: scratch [N], -1 on entry and on exit.
inline int32_t sub_order(const lvba::hvec<lvba::hvec<int32_t>> &nb, const lvba::hvec<int32_t> &nodes, lvba::hvec<int32_t> &local,
                         lvba::hvec<int32_t> &order_out, bool quick, const lvba::hvec<lvba::hvec<int32_t>> *extra = nullptr)
{
    const int m = (int)nodes.size();
    order_out.assign(nodes.begin(), nodes.end());
    if (m <= 2) return m - 1 > 0 ? m - 1 : 0;
    for (int a = 0; a < m; ++a) local[(size_t)nodes[(size_t)a]] = a;
    lvba::hvec<lvba::hvec<int32_t>> sub((size_t)m);
    for (int a = 0; a < m; ++a) {
        for (int v : nb[(size_t)nodes[(size_t)a]])
            if (local[(size_t)v] >= 0) sub[(size_t)a].push_back(local[(size_t)v]);
        if (extra) {
            for (int b : (*extra)[(size_t)a])
                if (b != a) sub[(size_t)a].push_back(b);
            std::sort(sub[(size_t)a].begin(), sub[(size_t)a].end());
            sub[(size_t)a].erase(std::unique(sub[(size_t)a].begin(), sub[(size_t)a].end()), sub[(size_t)a].end());
        }
    }
    for (int a = 0; a < m; ++a) local[(size_t)nodes[(size_t)a]] = -1;
    lvba::hvec<int32_t> perm;
    const int32_t bw = rcm_order_nb(sub, perm, quick);
    for (int a = 0; a < m; ++a) order_out[(size_t)a] = nodes[(size_t)perm[(size_t)a]];
    return bw;
}
} // namespace nd_detail

// One candidate: the separator set `in_sep` (by caller index).  Builds the whole plan and its cost; false if it degenerates.
inline bool nd_build_candidate(int N, const lvba::hvec<lvba::hvec<int32_t>> &nb, lvba::hvec<uint8_t> in_sep,
                               int n_ranks, const char *kind, bool quick, NdPlan &out)
{
    lvba::hvec<int32_t> scratch((size_t)N, -1);
    // connected components of the rest; components too small to be worth a factorisation of their own join the separator
    lvba::hvec<int32_t> comp((size_t)N, -1);
    lvba::hvec<lvba::hvec<int32_t>> parts;
    for (int pass = 0; pass < 2; ++pass) {
        std::fill(comp.begin(), comp.end(), -1);
        parts.clear();
        for (int s0 = 0; s0 < N; ++s0) {
            if (in_sep[(size_t)s0] || comp[(size_t)s0] >= 0) continue;
            const int id = (int)parts.size();
            parts.emplace_back();
            lvba::hvec<int32_t> stack(1, s0);
            comp[(size_t)s0] = id;
            while (!stack.empty()) {
                const int a = stack.back();
                stack.pop_back();
                parts[(size_t)id].push_back(a);
                for (int b : nb[(size_t)a])
                    if (!in_sep[(size_t)b] && comp[(size_t)b] < 0) { comp[(size_t)b] = id; stack.push_back(b); }
            }
        }
        bool moved = false;
        for (auto &p : parts)
            if ((int)p.size() < 22) { // < 2 panels
                for (int a : p) in_sep[(size_t)a] = 1;
                moved = true;
            }
        if (!moved) break;
    }
    lvba::hvec<int32_t> sep_nodes;
    for (int i = 0; i < N; ++i)
        if (in_sep[(size_t)i]) sep_nodes.push_back(i);
    const int Ns = (int)sep_nodes.size(), P = (int)parts.size();
    if (P < 1 || Ns < 1 || Ns > N / 2) return false; // (one arc is a plan too: a ring with a hub taken out is still a ring, with a narrow band)
    for (auto &p : parts) std::sort(p.begin(), p.end());
    // the separator's graph: direct edges + one clique per arc over the separator poses it touches
    lvba::hvec<int32_t> sep_local((size_t)N, -1);
    for (int q = 0; q < Ns; ++q) sep_local[(size_t)sep_nodes[(size_t)q]] = q;
    lvba::hvec<lvba::hvec<int32_t>> touch((size_t)P); // separator poses (local index in sep_nodes) each arc touches
    lvba::hvec<lvba::hvec<int32_t>> fillg((size_t)Ns);
    for (int a = 0; a < P; ++a) {
        lvba::hvec<uint8_t> seen((size_t)Ns, 0);
        for (int v : parts[(size_t)a])
            for (int b : nb[(size_t)v])
                if (in_sep[(size_t)b] && !seen[(size_t)sep_local[(size_t)b]]) { seen[(size_t)sep_local[(size_t)b]] = 1; touch[(size_t)a].push_back(sep_local[(size_t)b]); }
        for (int x : touch[(size_t)a])
            for (int y : touch[(size_t)a]) fillg[(size_t)x].push_back(y);
    }
    lvba::hvec<int32_t> sep_order;
    out = NdPlan();
    out.BbS = nd_detail::sub_order(nb, sep_nodes, scratch, sep_order, quick, &fillg);
    lvba::hvec<int32_t> sep_pos((size_t)N, -1); // caller index -> position in the separator's order
    for (int q = 0; q < Ns; ++q) sep_pos[(size_t)sep_order[(size_t)q]] = q;
    // the arcs, largest first (the owner assignment below deals them out in that order)
    lvba::hvec<int32_t> by_size((size_t)P);
    for (int a = 0; a < P; ++a) by_size[(size_t)a] = a;
    std::stable_sort(by_size.begin(), by_size.end(), [&](int a, int b) { return parts[(size_t)a].size() > parts[(size_t)b].size(); });
    out.perm.clear();
    out.perm.reserve((size_t)N);
    lvba::hvec<double> load((size_t)std::max(1, n_ranks), 0.0), chain((size_t)std::max(1, n_ranks), 0.0);
    for (int idx = 0; idx < P; ++idx) {
        const int a = by_size[(size_t)idx];
        NdPlanArc arc;
        lvba::hvec<int32_t> order;
        arc.Bb = nd_detail::sub_order(nb, parts[(size_t)a], scratch, order, quick);
        arc.p0 = (int32_t)out.perm.size();
        arc.Na = (int32_t)order.size();
        for (int v : order) out.perm.push_back(v);
        for (int x : touch[(size_t)a]) arc.sep.push_back(sep_pos[(size_t)sep_nodes[(size_t)x]]);
        std::sort(arc.sep.begin(), arc.sep.end());
        double fl = 0.0;
        const double t = nd_arc_seconds(6.0 * arc.Na, 6.0 * arc.Bb + 5.0, 6.0 * (double)arc.sep.size(), &fl);
        int best = 0; // the rank with the least work so far (ties: the lowest rank -- every rank computes the same assignment)
        for (int r = 1; r < (int)load.size(); ++r)
            if (load[(size_t)r] < load[(size_t)best]) best = r;
        arc.owner = best;
        load[(size_t)best] += fl;
        chain[(size_t)best] = std::max(chain[(size_t)best], t);
        out.arcs.push_back(arc);
    }
    out.ps = (int32_t)out.perm.size();
    out.Ns = Ns;
    for (int v : sep_order) out.perm.push_back(v);
    double t_arcs = 0.0;
    for (size_t r = 0; r < load.size(); ++r) t_arcs = std::max(t_arcs, std::max(load[r], chain[r])); // a rank's arcs run side by side
    out.t_nd = 1.15 * (t_arcs + nd_fact_seconds(6.0 * Ns, 6.0 * out.BbS + 5.0, true)) + 0.4e-3; // (+ fills, merges, back-substitution: measured)
    out.kind = kind;
    out.active = true;
    return true;
}

// perm_band / Bb_band: ordering.h's result for the whole graph.  min_gain: the dissection must be this much faster by the model.
inline NdPlan nd_plan(const uint8_t *adj, int N, const lvba::hvec<int32_t> &perm_band, int32_t Bb_band, int n_ranks, double min_gain = 0.8)
{
    NdPlan best;
    best.t_band = nd_fact_seconds(6.0 * N, 6.0 * Bb_band + 5.0, true);
    if (N < 256) return best; // (1 536 unknowns: two dozen panels either way)
    lvba::hvec<lvba::hvec<int32_t>> nb;
    nd_detail::neighbours(adj, N, nb);
    double t_best = best.t_band * min_gain;
    auto consider = [&](const lvba::hvec<uint8_t> &in_sep, const char *kind) {
        NdPlan cand;
        if (!nd_build_candidate(N, nb, in_sep, n_ranks, kind, true, cand)) return;
        cand.t_band = best.t_band;
        if (cand.t_nd < t_best) { t_best = cand.t_nd; best = cand; }
        // Long arcs are serial chains of their own (an arc is factorised top-down: its factor is reused by the border's forward
        // substitution): cut them further with chunks of THEIR band ordering -- the pieces are independent, run side by side on
        // their streams, and the chunks join the separator.  Repeated while the model gains (a folded ring falls into its two
        // sides at the first cut, each with half the bandwidth: the next cut's chunks are half as wide).
        NdPlan base = cand;
        for (int round = 0; round < 3; ++round) {
            NdPlan round_best;
            double t_round = base.t_nd;
            for (int pieces : {2, 4}) {
                lvba::hvec<uint8_t> s2((size_t)N, 0);
                for (int q = base.ps; q < N; ++q) s2[(size_t)base.perm[(size_t)q]] = 1;
                bool any = false;
                for (const NdPlanArc &a : base.arcs) {
                    const int w = std::max(1, (int)a.Bb);
                    const double piece = ((double)a.Na - (double)(pieces - 1) * w) / pieces;
                    if (piece < 3.0 * w || piece < 22.0) continue;
                    for (int c = 0; c < pieces - 1; ++c) {
                        const int a0 = a.p0 + (int)((c + 1) * piece + c * w + 0.5);
                        for (int q = a0; q < a0 + w && q < a.p0 + a.Na; ++q) s2[(size_t)base.perm[(size_t)q]] = 1;
                    }
                    any = true;
                }
                if (!any) break;
                NdPlan c2;
                if (!nd_build_candidate(N, nb, s2, n_ranks, kind, true, c2)) continue;
                c2.t_band = best.t_band;
                if (c2.t_nd < t_round) { t_round = c2.t_nd; round_best = c2; }
            }
            if (!round_best.active) break;
            base = round_best;
            if (base.t_nd < t_best) { t_best = base.t_nd; best = base; }
        }
    };
    // ---- hubs
    {
        lvba::hvec<int32_t> deg((size_t)N);
        for (int i = 0; i < N; ++i) deg[(size_t)i] = (int32_t)nb[(size_t)i].size();
        lvba::hvec<int32_t> sorted(deg);
        std::sort(sorted.begin(), sorted.end());
        const double med = (double)sorted[(size_t)N / 2];
        for (double tau : {1.35, 1.6, 2.0}) {
            lvba::hvec<uint8_t> in_sep((size_t)N, 0);
            int cnt = 0;
            for (int i = 0; i < N; ++i)
                if ((double)deg[(size_t)i] > tau * med) { in_sep[(size_t)i] = 1; ++cnt; }
            if (cnt >= 2 && cnt <= N / 3) consider(in_sep, "hubs");
        }
    }
    // ---- chunks of the band ordering: what lets several RANKS share a long band (on one GPU the arcs would only share its
    // matrix pipes: nothing to gain over the two-ended band).  K chunks give K + 1 pieces of a path or 2 K arcs of a folded ring:
    // K = ranks / 2 and K = ranks - 1 are the two counts that can give every rank an arc.
    // LVBA_ND_CHUNK_RANKS=r (measurement only, with LVBA_SOLVER=nd): cut as for r ranks although fewer -- or one -- run the arcs
    int cr = n_ranks;
    if (const char *e = getenv("LVBA_ND_CHUNK_RANKS")) cr = std::max(cr, atoi(e));
    if (cr >= 2 && (int64_t)Bb_band * 6 <= N) {
        const int w = Bb_band; // a chunk of `Bb_band` consecutive positions cuts the band (edges reach at most Bb_band)
        for (int K : {std::max(1, cr / 2), cr - 1}) {
            if ((int64_t)K * w * 3 > N || (K == cr - 1 && K == std::max(1, cr / 2))) continue;
            for (int variant = 0; variant < 2; ++variant) {
                // pieces between the chunks: K + 1 of them.  variant 0: equal pieces; variant 1: the two END pieces half as long
                // (on a folded ring the inner pieces fall into two arcs each, the end pieces -- around the folds -- do not)
                const double units = variant == 0 ? (double)(K + 1) : (double)K;
                const double piece = ((double)N - (double)K * w) / units;
                if (piece < 2.0 * w) continue;
                lvba::hvec<uint8_t> in_sep((size_t)N, 0);
                double pos = variant == 0 ? piece : 0.5 * piece;
                for (int c = 0; c < K; ++c) {
                    const int a0 = (int)(pos + 0.5);
                    for (int q = a0; q < a0 + w && q < N; ++q) in_sep[(size_t)perm_band[(size_t)q]] = 1;
                    pos += piece + w;
                }
                consider(in_sep, "chunks");
            }
        }
    }
    if (best.active) { // the partition that won, ordered in full (hill-climbing on every arc and on the separator's graph)
        lvba::hvec<uint8_t> in_sep((size_t)N, 0);
        for (int q = best.ps; q < N; ++q) in_sep[(size_t)best.perm[(size_t)q]] = 1;
        NdPlan full;
        if (nd_build_candidate(N, nb, in_sep, n_ranks, best.kind, false, full) && full.t_nd <= best.t_nd * 1.02) {
            full.t_band = best.t_band;
            best = full;
        }
    }
    return best;
}

} // namespace lvba
