// ldlt_schedule.h -- which launch does what in the look-ahead form of the band LDL^T (host-only, no HIP: ldlt.hip turns the
// list into launches, tests/ldlt_schedule_check.cpp replays it on a tile-level model of the factorisation and checks that
// every tile receives every contribution exactly once, from one writer per launch, after its inputs exist).
//
// Launch X_p (ldlt_lookahead.h) carries the chain and row roles of panel p -- they turn block column p into L, apply panels
// p - 1 and p to block column p + 1 and factorise the diagonal block p + 1 -- and up to two BULK jobs: trailing updates of
// earlier panels on block columns >= p + 2.  A job is (panel o [, partner e = o - 1], tile columns [ca, cb)) in bulk
// coordinates: tile column tj' <-> block column o + 2 + tj'.  What a panel q owes:
//     block column q + 1   roles of X_q            block column q + 2   roles of X_{q+1}
//     block columns >= q + 3 (tj' >= 1)            a job, in a launch X_r with q + 1 <= r <= (block column) - 2
// Consecutive panels are PAIRED (e, o = e + 1) so that every C tile is read and written once for both (rank 128):
//     X_o      row roles   (q_extra)               = block column e + 3 from panel e alone, which X_{o+1}'s roles need complete:
//                                                    the row workgroups of X_o hold L(i, e) anyway (their q is e), one product more
//                                                    each -- as a bulk job of its own ("single job (e, tj' = 1)") it made the
//                                                    second-half launches of config C3 534 workgroups on 512 seats
//     X_{e+2}  pair job    (o + e, tj' in [1, cs)) = block columns e + 4 .. : the first half (at least tj' = 1, 2)
//     X_{e+3}  pair job    (o + e, tj' >= cs)      the second half, beside the next pair's q_extra products (block column e + 5 < e + 6)
// Panels left over (an odd count, windows too short to pair) run their whole update as one single job in X_{q+1}.
//
// DEFERRED form (round 6; the phases that close early, i.e. the two ends of the twisted factorisation): panel q's update of block
// column q + 1 moves from the END of X_q -- where every row workgroup had to recompute L(q+1, q), a product that 40 workgroups per
// problem did alike -- to the START of X_{q+1}, where L(i, q) and Z(q+1, q) are stored operands:
//     row t >= 2 of X_p ("lean")   A(i,p) -= L(i,q) Z(p,q)^T;  L(i,p) = A(i,p) G_p;  A(i,p+1) -= L(i,q) Z(p+1,q)^T     (3 products)
//     row 1 of X_p, and every row of the phase's closing launch: the same first product, then the full form (the chain workgroup
//     of X_{p+1} needs A(p+2, p+1) complete; the Schur complement must be complete when the phase ends)
// The chain role and the bulk jobs are unchanged.
#pragma once
#include <cstdint>
#include <vector>

namespace lvba {

struct SchedJob { int64_t o; int pair; int64_t ca, cb; };
struct SchedLaunch {
    int kind; // 0: the first diagonal block of a phase (ldlt_diag_blocked_kernel), 1: a step launch
    int64_t p;
    int roles, has_q, do_diag, njobs;
    int q_extra; // the row roles also apply panel q = p - 1 to block column p + 2 (q is the first panel of a pair: see below)
    int defer;   // the deferred form (above)
    SchedJob job[2];
};

inline int64_t sched_pair_items(int64_t tj, int64_t Tb) { return (Tb - tj + 1) / 2; } // 128 x 64 tiles of bulk tile column tj

// Panels [sa, sb) of one problem.  Tof(st) = tile rows below panel st (0: the last panel).  close: the phase ends with the
// Schur complement complete on the block columns >= sb and the diagonal block sb NOT factorised (the two ends of the twisted
// factorisation); else the phase runs to the last panel of the matrix.
template <class TF>
void ldlt_schedule_phase(int64_t sa, int64_t sb, bool close, bool rank128, TF Tof, std::vector<SchedLaunch> &out, bool defer = false)
{
    auto is_e = [&](int64_t st) { return rank128 && st >= sa && ((st - sa) % 2 == 0) && st + 1 < sb && Tof(st) >= 3 && Tof(st + 1) >= 2; };
    auto is_o = [&](int64_t st) { return st > sa && is_e(st - 1); };
    auto add = [&](SchedLaunch &L, int64_t o, int pair, int64_t ca, int64_t cb) {
        const int64_t Tb = Tof(o) - 1;
        if (cb > Tb) cb = Tb;
        if (ca < 0) ca = 0;
        if (ca >= cb) return;
        L.job[L.njobs++] = SchedJob{o, pair, ca, cb};
    };
    auto bulk_only = [&]() { SchedLaunch L{}; L.kind = 1; L.p = -1; return L; };
    struct { bool on; int64_t o, ca; } pend{false, 0, 0};
    auto flush = [&]() {
        if (!pend.on) return;
        SchedLaunch L = bulk_only();
        add(L, pend.o, 1, pend.ca, INT64_MAX);
        if (L.njobs) out.push_back(L);
        pend.on = false;
    };
    {
        SchedLaunch L{};
        L.kind = 0; L.p = sa;
        out.push_back(L);
    }
    for (int64_t p = sa; p < sb; ++p) {
        if (Tof(p) == 0) break;
        SchedLaunch L{};
        L.kind = 1; L.p = p; L.roles = 1; L.has_q = p > sa; L.do_diag = !(close && p == sb - 1); L.defer = defer ? 1 : 0;
        if (L.has_q) {
            const int64_t q = p - 1;
            if (is_o(q)) { // X_{e+2}: the first half of the pair's rank-128 update (at least its tile columns 1 and 2)
                const int64_t Tb = Tof(q) - 1;
                int64_t tot = 0, part = 0, cs = 1;
                for (int64_t c = 1; c < Tb; ++c) tot += sched_pair_items(c, Tb);
                while (cs < Tb && (cs < 3 || 2 * part < tot)) part += sched_pair_items(cs++, Tb);
                flush(); // (nothing is pending here in a regular sequence)
                add(L, q, 1, 1, cs);
                if (cs < Tb) { pend.on = true; pend.o = q; pend.ca = cs; }
            } else if (is_e(q)) { // X_o: block column e + 3 from panel e alone (row roles), beside the previous pair's second half
                if (pend.on) { add(L, pend.o, 1, pend.ca, INT64_MAX); pend.on = false; }
                L.q_extra = 1;
            } else { // a panel without a partner: its whole update, alone in the launch
                flush();
                add(L, q, 0, 1, INT64_MAX);
            }
        }
        out.push_back(L);
    }
    if (close) {
        const int64_t last = sb - 1;
        if (is_o(last)) { // the last pair's update in full, and block column sb + 1 from panel sb - 1 (no X_sb will do it)
            flush();
            SchedLaunch L = bulk_only();
            add(L, last, 1, 1, INT64_MAX);
            add(L, last, 0, 0, 1);
            if (L.njobs) out.push_back(L);
        } else {
            flush();
            SchedLaunch L = bulk_only();
            add(L, last, 0, 0, INT64_MAX);
            if (L.njobs) out.push_back(L);
        }
    }
    flush();
}

} // namespace lvba
