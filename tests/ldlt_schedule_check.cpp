// CPU check of global-lvba_amd/csrc/ldlt_schedule.h (test infrastructure; compiled by tests/test_ordering.py).
//
// The look-ahead LDL^T (csrc/ldlt_lookahead.h) has no synchronisation inside a launch: it is correct iff the launch list is.
// This program replays the list on a TILE-LEVEL MODEL of the band factorisation -- a tile (i, j) holds the set of panels whose
// contribution L(i,k) D_k L(j,k)^T it has received -- and checks, launch by launch:
//   * every tile has at most one writer per launch, and nothing written in a launch is read in it (as L, as Z, or as the side
//     copy of A(p+1, p));
//   * a block column is turned into L only when all its contributions are in; a diagonal block is factorised only then;
//   * the roles of X_p find block column p + 1 holding exactly the panels <= p - 2; no contribution is applied twice;
//   * every L / Z / G / side copy that is read was produced by an EARLIER launch;
//   * at the end every contribution has been applied (Schur complement complete when the phase closes early);
//   * BUFFER LIFETIMES of ldlt.hip's run_phase: Z of panel k lives in slot k % 4 of a ring that X_{k+4} overwrites -- every launch
//     that reads Z_k (the roles as Z_q, the bulk jobs as Z_o / Z_e) must come before X_{k+4}, and not be X_{k+4} itself; the side
//     copy of A(p+1, p) and panel q's ready-made share of the diagonal block alternate between two slots by panel parity -- a
//     reader must find the panel it expects in its slot.
// usage: ldlt_schedule_check   (runs a list of geometries incl. config C3's: n = 12000, half-bandwidth 2597)
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#include "../global-lvba_amd/csrc/ldlt_schedule.h"

using lvba::SchedJob;
using lvba::SchedLaunch;

typedef std::pair<int64_t, int64_t> TK;
static inline TK tk(int64_t i, int64_t j) { return TK(i, j); }
static int g_fail = 0;
#define CHECK(c, ...) do { if (!(c)) { if (g_fail < 20) { printf("FAILED line %d: %s  ", __LINE__, #c); printf(__VA_ARGS__); printf("\n"); } ++g_fail; } } while (0)

struct Model {
    int64_t nf, bw, nsteps, sa, sb;
    bool close;
    std::vector<int64_t> T;
    std::map<std::pair<int64_t, int64_t>, std::set<int64_t>> applied;
    std::map<std::pair<int64_t, int64_t>, int> isL; // tile (i, k) -> launch that turned it into L
    std::map<int64_t, int> diag_done, side_copy, dq_made; // dq_made[q]: row 1 of X_q left L(q+2, q) Z(q+2, q)^T for X_{q+1}'s chain
    std::set<std::pair<int64_t, int64_t>> written, readL;
    int now = 0;
    int64_t cur_p = -1;                    // panel of the last roles launch enqueued (the one that owns Z slot cur_p % 4 now)
    int64_t side_slot[2] = {-1, -1}, dq_slot[2] = {-1, -1}; // which panel's side copy / diagonal share each of the two slots holds
    void needZ(int64_t k)
    {
#ifndef LVBA_ZRING
#define LVBA_ZRING 4 // (ldlt.hip: Zbuf[st % 4]; -DLVBA_ZRING=3 must FAIL: the schedule needs all four slots)
#endif
        CHECK(cur_p - k <= LVBA_ZRING - 1, "Z_%lld read in launch %d after X_%lld took its ring slot (current roles panel %lld)", (long long)k, now,
              (long long)(k + 4), (long long)cur_p);
    }

    Model(int64_t nf_, int64_t bw_) : nf(nf_), bw(bw_)
    {
        nsteps = (nf + 63) / 64;
        for (int64_t st = 0; st < nsteps; ++st) {
            const int64_t k = 64 * st, nbe = std::min<int64_t>(64, nf - k), w0 = k + nbe, rend = std::min(w0 + bw, nf);
            T.push_back(w0 < rend ? (rend - w0 + 63) / 64 : 0);
        }
    }
    int64_t Tof(int64_t st) const { return st < nsteps ? T[(size_t)st] : 0; }
    // panels of this phase that owe tile (i, j) a contribution
    std::set<int64_t> eligible(int64_t i, int64_t j, int64_t below) const
    {
        std::set<int64_t> e;
        for (int64_t k = sa; k < std::min(std::min(j, sb), below); ++k)
            if (i <= k + Tof(k)) e.insert(k);
        return e;
    }
    bool complete(int64_t i, int64_t j) { return applied[tk(i, j)] == eligible(i, j, INT64_MAX); }
    void write(int64_t i, int64_t j)
    {
        CHECK(!written.count(tk(i, j)), "tile (%lld,%lld) has two writers in launch %d", (long long)i, (long long)j, now);
        written.insert(tk(i, j));
    }
    void needL(int64_t i, int64_t k)
    {
        auto it = isL.find(tk(i, k));
        CHECK(it != isL.end() && it->second < now, "L(%lld,%lld) read in launch %d before it exists", (long long)i, (long long)k, now);
        readL.insert(tk(i, k));
    }
    void apply(int64_t i, int64_t j, int64_t k)
    {
        CHECK(k < j && i <= k + Tof(k), "panel %lld does not reach tile (%lld,%lld)", (long long)k, (long long)i, (long long)j);
        CHECK(!applied[tk(i, j)].count(k), "panel %lld applied twice to tile (%lld,%lld), launch %d", (long long)k, (long long)i, (long long)j, now);
        applied[tk(i, j)].insert(k);
    }
    void run(const std::vector<SchedLaunch> &sched)
    {
        for (const SchedLaunch &L : sched) {
            ++now;
            written.clear();
            readL.clear();
            if (L.kind == 0) {
                CHECK(complete(L.p, L.p), "first diagonal block %lld incomplete", (long long)L.p);
                diag_done[L.p] = now;
                if (Tof(L.p) > 0) { CHECK(complete(L.p + 1, L.p), "side tile incomplete"); side_copy[L.p] = now; side_slot[L.p % 2] = L.p; }
                continue;
            }
            if (L.roles) {
                const int64_t p = L.p, q = p - 1;
                cur_p = p; // (this launch writes Z slot p % 4: a job of the same launch must not read Z_{p-4})
                if (Tof(p) >= 2) CHECK(side_slot[p % 2] == p, "launch X_%lld finds the side copy of panel %lld in its slot", (long long)p, (long long)side_slot[p % 2]);
                if (L.has_q) needZ(q);
                CHECK(diag_done.count(p) && diag_done[p] < now, "G_%lld not there for launch %d", (long long)p, now);
                CHECK(L.has_q == (p > sa), "has_q");
                for (int64_t t = 0; t < Tof(p); ++t) {
                    const int64_t i = p + 1 + t;
                    if (L.defer && L.has_q && t >= 1 && i <= q + Tof(q)) { // deferred form: panel q's update of the row's own tile comes first
                        CHECK(applied[tk(i, p)] == eligible(i, p, q), "tile (%lld,%lld) before its deferred update: wrong set", (long long)i, (long long)p);
                        needL(i, q);
                        needL(p, q);
                        apply(i, p, q);
                    }
                    CHECK(complete(i, p), "tile (%lld,%lld) becomes L before it is complete", (long long)i, (long long)p);
                    write(i, p);
                    const bool lean = L.defer && L.do_diag && t >= 2;
                    // block column p + 1 in this row: exactly the panels <= p - 2 so far
                    CHECK(applied[tk(i, p + 1)] == eligible(i, p + 1, p - 1), "tile (%lld,%lld) before the roles of X_%lld: wrong set", (long long)i,
                          (long long)(p + 1), (long long)p);
                    if (L.has_q && i <= q + Tof(q)) {
                        needL(i, q);
                        needL(p + 1, q);
                        apply(i, p + 1, q);
                        // the chain workgroup (t = 0) takes panel q's share of the diagonal block from what row 1 of X_q left
                        if (t == 0) {
                            CHECK(dq_made.count(q) && dq_made[q] < now, "launch %d: no ready-made share of panel %lld for block %lld", now, (long long)q, (long long)(p + 1));
                            CHECK(dq_slot[q % 2] == q, "X_%lld's chain finds panel %lld's share in the slot of panel %lld's", (long long)p, (long long)dq_slot[q % 2], (long long)q);
                        }
                    }
                    if (!lean) {
                        if (t >= 1) CHECK(side_copy.count(p) && side_copy[p] < now, "side copy of A(%lld,%lld) missing", (long long)(p + 1), (long long)p);
                        apply(i, p + 1, p);
                    }
                    if (!lean || (L.has_q && i <= q + Tof(q))) write(i, p + 1);
                    if (t == 1) {
                        CHECK(complete(i, p + 1), "side copy taken of an incomplete tile");
                        side_copy[p + 1] = now; dq_made[p] = now;
                        side_slot[(p + 1) % 2] = p + 1; dq_slot[p % 2] = p; // (written at the END of the launch: the readers above have passed)
                    }
                    if (L.q_extra && t >= 1 && i <= q + Tof(q)) { // block column p + 2 from panel q, by the row that holds L(i, q)
                        CHECK(L.has_q, "q_extra without q");
                        needL(i, q);
                        needL(p + 2, q);
                        apply(i, p + 2, q);
                        write(i, p + 2);
                    }
                }
                for (int64_t t = 0; t < Tof(p); ++t) isL[tk(p + 1 + t, p)] = now;
                if (L.do_diag) {
                    CHECK(complete(p + 1, p + 1), "diagonal block %lld factorised incomplete", (long long)(p + 1));
                    diag_done[p + 1] = now;
                }
            }
            for (int jn = 0; jn < L.njobs; ++jn) {
                const SchedJob &J = L.job[jn];
                const int64_t o = J.o, e = o - 1, Tb = Tof(o) - 1;
                CHECK(J.ca >= 0 && J.cb <= Tb && J.ca < J.cb, "job columns [%lld,%lld) of %lld", (long long)J.ca, (long long)J.cb, (long long)Tb);
                needZ(o);
                CHECK(!(L.roles && o == L.p), "a job reads Z of the panel its own launch is writing");
                if (J.pair) needZ(e);
                for (int64_t tj = J.ca; tj < J.cb; ++tj) {
                    const int64_t j = o + 2 + tj;
                    for (int64_t i = j; i <= o + Tof(o); ++i) {
                        needL(i, o);
                        needL(j, o);
                        apply(i, j, o);
                        if (J.pair && i <= e + Tof(e)) {
                            CHECK(e >= sa, "partner below the phase");
                            needL(i, e);
                            needL(j, e);
                            apply(i, j, e);
                        }
                        write(i, j);
                    }
                }
            }
            for (const auto &rd : readL) CHECK(!written.count(rd), "tile (%lld,%lld) read and written in launch %d", (long long)rd.first, (long long)rd.second, now);
        }
        // the end of the phase
        const int64_t last = close ? sb : nsteps;
        for (int64_t k = sa; k < last; ++k) {
            if (!(close && k == sb)) CHECK(diag_done.count(k), "diagonal block %lld never factorised", (long long)k);
            for (int64_t t = 0; t < Tof(k); ++t) CHECK(isL.count(tk(k + 1 + t, k)), "tile (%lld,%lld) never became L", (long long)(k + 1 + t), (long long)k);
        }
        if (close) {
            CHECK(!diag_done.count(sb), "the block after the phase must not be factorised");
            for (int64_t j = sb; j < nsteps; ++j)
                for (int64_t i = j; i < nsteps; ++i)
                    if (!eligible(i, j, INT64_MAX).empty()) CHECK(complete(i, j), "Schur complement incomplete at (%lld,%lld)", (long long)i, (long long)j);
        }
    }
};

static int launches_of(const std::vector<SchedLaunch> &s) { return (int)s.size(); }

static void run_case(int64_t n, int64_t bw, bool twist, bool rank128, bool defer)
{
    // as ldlt_solve: the two ends eliminate P1 panels each when the band allows it, then the middle runs on matrix 1
    int64_t P1 = twist ? (n - bw) / 128 : 0;
    if (P1 < 4) P1 = 0;
    const int64_t nf = n - 64 * P1;
    int total = 0;
    if (P1 > 0) {
        Model m(nf, bw);
        m.sa = 0; m.sb = P1; m.close = true;
        std::vector<SchedLaunch> s;
        lvba::ldlt_schedule_phase(m.sa, m.sb, true, rank128, [&](int64_t st) { return m.Tof(st); }, s, defer);
        m.run(s);
        total += launches_of(s);
    }
    {
        Model m(nf, bw);
        m.sa = P1; m.sb = m.nsteps; m.close = false;
        std::vector<SchedLaunch> s;
        lvba::ldlt_schedule_phase(m.sa, m.sb, false, rank128, [&](int64_t st) { return m.Tof(st); }, s);
        m.run(s);
        total += launches_of(s);
    }
    printf("n=%lld bw=%lld twist=%d rank128=%d defer=%d: %d launches, %s\n", (long long)n, (long long)bw, (int)twist, (int)rank128, (int)defer, total,
           g_fail ? "FAILED" : "ok");
}

int main()
{
    const int64_t cases[][2] = {{12000, 2597}, {12000, 143}, {4200, 155}, {4200, 1000}, {640, 600}, {700, 23}, {64, 5}, {65, 64}, {128, 127},
                                {1000, 999}, {6000, 767}, {6000, 768}, {6000, 769}, {3001, 333}, {60000, 2549}};
    for (auto &c : cases)
        for (int tw = 0; tw < 2; ++tw)
            for (int r = 0; r < 2; ++r)
                for (int df = 0; df <= tw; ++df) { // (the deferred form exists for the phases that close early)
                    run_case(c[0], c[1], tw != 0, r != 0, df != 0);
                    if (g_fail) { printf("%d check(s) failed\n", g_fail); return 1; }
                }
    printf("ldlt schedule ok\n");
    return 0;
}
