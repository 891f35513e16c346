// tests/host_transport.cpp -- TEST-ONLY all-reduce transport for liblvba_hip.so's external-transport entry points
// (lvba_balm_dist_init_external / lvba_visual_dist_init_external, include/lvba_hip.h).
//
// Purpose: run the library's multi-rank code paths (max-reduced band width, all-reduced adjacency and the common pose order,
// the packed [H | g | cost] all-reduce, the global voxel count, the partitioned solve) with N > 1 ranks on a box with ONE GPU,
// where RCCL refuses two ranks on the same device.  Ranks are HOST THREADS of one process, each with its own library handle
// and stream on the same device; an all-reduce copies every rank's buffer to the host, meets the others at a barrier, reduces
// in RANK ORDER (so every rank computes bitwise the same result, as RCCL guarantees for its own reductions) and copies the
// result back.  It moves data only; every arithmetic operation on problem data stays in the library's kernels.  Not part of
// the product: built by __graft_entry__.build() into tests/_build/libhost_transport.so and loaded by the GPU tests.
//
// A rank that fails elsewhere never reaches the barrier: waits time out (HT_TIMEOUT_S, default 300 s) and poison the
// communicator, so the other ranks return an error instead of hanging.
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct Comm {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool poisoned = false;
    std::vector<std::vector<unsigned char>> stage;
    uint64_t calls = 0, bytes = 0;
    double timeout_s = 300.0;
    // returns false when the communicator is (or becomes) poisoned
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (poisoned) return false;
        const uint64_t g = gen;
        if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); return true; }
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return gen != g || poisoned; });
        if (!ok || poisoned) { poisoned = true; cv.notify_all(); return false; }
        return true;
    }
};
struct RankCtx { Comm *c; int rank; };

template <typename T>
void reduce(const std::vector<std::vector<unsigned char>> &stage, size_t count, bool is_max, T *out)
{
    const int n = (int)stage.size();
    const T *r0 = reinterpret_cast<const T *>(stage[0].data());
    for (size_t e = 0; e < count; ++e) out[e] = r0[e];
    for (int r = 1; r < n; ++r) {
        const T *p = reinterpret_cast<const T *>(stage[(size_t)r].data());
        if (is_max) { for (size_t e = 0; e < count; ++e) out[e] = p[e] > out[e] ? p[e] : out[e]; }
        else { for (size_t e = 0; e < count; ++e) out[e] = (T)(out[e] + p[e]); }
    }
}

} // namespace

extern "C" {

void *ht_create(int n_ranks)
{
    Comm *c = new Comm();
    c->n = n_ranks;
    c->stage.resize((size_t)n_ranks);
    if (const char *e = getenv("HT_TIMEOUT_S")) c->timeout_s = atof(e);
    return c;
}
void *ht_rank(void *comm, int rank) { return new RankCtx{static_cast<Comm *>(comm), rank}; }
void ht_rank_free(void *ctx) { delete static_cast<RankCtx *>(ctx); }
void ht_destroy(void *comm) { delete static_cast<Comm *>(comm); }
void ht_poison(void *comm) // a rank that failed elsewhere releases the others
{
    Comm *c = static_cast<Comm *>(comm);
    std::lock_guard<std::mutex> g(c->mu);
    c->poisoned = true;
    c->cv.notify_all();
}
void ht_stats(void *comm, uint64_t *calls, uint64_t *bytes)
{
    Comm *c = static_cast<Comm *>(comm);
    std::lock_guard<std::mutex> g(c->mu);
    *calls = c->calls; *bytes = c->bytes;
}

// lvba_allreduce_fn (dtype: 0 f64, 1 i64, 2 i32, 3 u8; op: 0 sum, 1 max)
int32_t ht_allreduce(void *ctx, void *dbuf, size_t count, int32_t dtype, int32_t op, void *stream)
{
    RankCtx *rc = static_cast<RankCtx *>(ctx);
    Comm &hc = *rc->c;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t esz = dtype <= 1 ? 8 : dtype == 2 ? 4 : 1;
    if (dtype < 0 || dtype > 3 || op < 0 || op > 1) return 1;
    std::vector<unsigned char> &mine = hc.stage[(size_t)rc->rank];
    mine.resize(count * esz);
    if (hipMemcpyAsync(mine.data(), dbuf, count * esz, hipMemcpyDeviceToHost, s) != hipSuccess) { ht_poison(&hc); return 2; }
    if (hipStreamSynchronize(s) != hipSuccess) { ht_poison(&hc); return 2; }
    if (!hc.barrier()) return 3;
    std::vector<unsigned char> res(count * esz);
    const bool is_max = op == 1;
    if (dtype == 0) reduce<double>(hc.stage, count, is_max, reinterpret_cast<double *>(res.data()));
    else if (dtype == 1) reduce<int64_t>(hc.stage, count, is_max, reinterpret_cast<int64_t *>(res.data()));
    else if (dtype == 2) reduce<int32_t>(hc.stage, count, is_max, reinterpret_cast<int32_t *>(res.data()));
    else reduce<uint8_t>(hc.stage, count, is_max, res.data());
    if (rc->rank == 0) { std::lock_guard<std::mutex> g(hc.mu); hc.calls += 1; hc.bytes += count * esz; }
    if (!hc.barrier()) return 3; // nobody refills its stage before everybody has read it
    if (hipMemcpyAsync(dbuf, res.data(), count * esz, hipMemcpyHostToDevice, s) != hipSuccess) { ht_poison(&hc); return 2; }
    if (hipStreamSynchronize(s) != hipSuccess) { ht_poison(&hc); return 2; } // res is a local
    return 0;
}

} // extern "C"
