// mempool.h -- caching device allocator for short-lived workspaces.
// hipMalloc / hipFree of 100 MB-class buffers cost 0.2-0.5 ms each and hipFree synchronises the device; the voxel
// front-end needs ~40 temporaries per map.  Blocks are cached per device in size classes (3 mantissa bits: <= 12.5 %
// slack) and handed back out; lvba_release_cached_memory() returns them to the driver.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <condition_variable>
#include <map>
#include <mutex>
#include <vector>

namespace lvba {

class DevicePool {
  public:
    static DevicePool &get()
    {
        static DevicePool p;
        return p;
    }
    static size_t size_class(size_t bytes)
    {
        if (bytes < 512) return 512;
        int hb = 63 - __builtin_clzll((unsigned long long)bytes);
        const size_t step = (size_t)1 << (hb - 3);
        return (bytes + step - 1) / step * step;
    }
    hipError_t alloc(void **p, size_t bytes)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const size_t cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(mu_);
            auto &v = free_[key(dev, cls)];
            if (!v.empty()) {
                *p = v.back();
                v.pop_back();
                cached_ -= cls;
                owner_[*p] = key(dev, cls);
                return hipSuccess;
            }
        }
        hipError_t e = hipMalloc(p, cls);
        if (e == hipErrorOutOfMemory) { // give the cache back and retry once
            (void)hipGetLastError(); // the failed hipMalloc stays latched in the thread's last-error slot: a later
                                     // HIPCHK(hipGetLastError()) would report NOMEM although the retry succeeded
            release();
            e = hipMalloc(p, cls);
        }
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> g(mu_);
            owner_[*p] = key(dev, cls);
        }
        return e;
    }
    // The caller guarantees no work that touches p is still in flight.
    void free(void *p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> g(mu_);
        auto it = owner_.find(p);
        if (it == owner_.end()) { (void)hipFree(p); return; }
        free_[it->second].push_back(p);
        cached_ += it->second.second;
        owner_.erase(it);
    }
    size_t release()
    {
        std::lock_guard<std::mutex> g(mu_);
        size_t n = cached_;
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto &kv : free_) {
            if (kv.second.empty()) continue;
            (void)hipSetDevice(kv.first.first);
            for (void *p : kv.second) (void)hipFree(p);
            kv.second.clear();
        }
        (void)hipSetDevice(cur);
        cached_ = 0;
        return n;
    }
    size_t cached_bytes()
    {
        std::lock_guard<std::mutex> g(mu_);
        return cached_;
    }

  private:
    typedef std::pair<int, size_t> Key;
    static Key key(int dev, size_t cls) { return Key(dev, cls); }
    std::mutex mu_;
    std::map<Key, std::vector<void *>> free_;
    std::map<void *, Key> owner_;
    size_t cached_ = 0;
};

// Streams are cached too: hipStreamCreate costs ~0.1 ms and hipStreamDestroy ~0.6 ms (measured, round 2: it waits for the
// device's queues), and a window stage used to create and destroy seven of them per call (one per worker thread, one per handle):
// 2.8 of the 16 ms of bench.py's window leg.  acquire() hands out an idle non-blocking stream of the current device, release()
// takes it back (draining it first if the caller has not); lvba_release_cached_memory() destroys what is cached.
class StreamCache {
  public:
    static StreamCache &get()
    {
        static StreamCache c;
        return c;
    }
    hipError_t acquire(hipStream_t *s)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(mu_);
            auto &v = free_[dev];
            if (!v.empty()) {
                *s = v.back();
                v.pop_back();
                owner_[*s] = dev;
                return hipSuccess;
            }
        }
        const hipError_t e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> g(mu_);
            owner_[*s] = dev;
        }
        return e;
    }
    void release(hipStream_t s)
    {
        if (!s) return;
        if (hipStreamQuery(s) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); }
        std::lock_guard<std::mutex> g(mu_);
        auto it = owner_.find(s);
        if (it == owner_.end()) { (void)hipStreamDestroy(s); return; }
        auto &v = free_[it->second];
        owner_.erase(it);
        if (v.size() < kKeep) v.push_back(s);
        else (void)hipStreamDestroy(s);
    }
    void release_all()
    {
        std::lock_guard<std::mutex> g(mu_);
        for (auto &kv : free_) {
            for (hipStream_t q : kv.second) (void)hipStreamDestroy(q);
            kv.second.clear();
        }
    }

  private:
    static constexpr size_t kKeep = 16; // per device
    std::mutex mu_;
    std::map<int, std::vector<hipStream_t>> free_;
    std::map<hipStream_t, int> owner_;
};

// Pinned staging memory for the scan upload (voxelize.hip: 24 MB per call): hipHostMalloc / hipHostFree of that size cost
// milliseconds and the unpinning stalls the device's queues (host_arena.h); a refinement handle's three small zero-copy blocks come
// from here too.  Blocks are allocated portable + mapped: a block pinned under one device is reused as zero-copy / staging memory
// by handles on another (lvba_window_ba_multi, lvba_lidar_ba_multi), which must not rest on what the default flags happen to
// give.  Small (handle) blocks and large (upload) blocks are cached in lists of their own, so that a burst of 4 KB handle blocks
// cannot evict the 24 MB upload block: at most kKeep of each stay cached.
class PinnedCache {
  public:
    static PinnedCache &get()
    {
        static PinnedCache c;
        return c;
    }
    hipError_t acquire(void **p, size_t bytes)
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            auto &fl = list_of(bytes);
            for (size_t i = 0; i < fl.size(); ++i)
                if (fl[i].second >= bytes && fl[i].second <= 2 * bytes + ((size_t)1 << 20)) {
                    *p = fl[i].first;
                    owner_[*p] = fl[i].second;
                    fl.erase(fl.begin() + (ptrdiff_t)i);
                    return hipSuccess;
                }
        }
        const hipError_t e = hipHostMalloc(p, bytes, hipHostMallocPortable | hipHostMallocMapped);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> g(mu_);
            owner_[*p] = bytes;
        }
        return e;
    }
    void release(void *p) // (no copy out of p is in flight any more)
    {
        if (!p) return;
        std::lock_guard<std::mutex> g(mu_);
        auto it = owner_.find(p);
        if (it == owner_.end()) { (void)hipHostFree(p); return; }
        const size_t bytes = it->second;
        owner_.erase(it);
        auto &fl = list_of(bytes);
        if (fl.size() < kKeep) fl.emplace_back(p, bytes);
        else (void)hipHostFree(p);
    }
    size_t release_all()
    {
        std::lock_guard<std::mutex> g(mu_);
        size_t n = 0;
        for (auto *fl : {&free_small_, &free_large_}) {
            for (auto &b : *fl) { n += b.second; (void)hipHostFree(b.first); }
            fl->clear();
        }
        return n;
    }

  private:
    static constexpr size_t kKeep = 8, kSmall = (size_t)1 << 20;
    std::vector<std::pair<void *, size_t>> &list_of(size_t bytes) { return bytes <= kSmall ? free_small_ : free_large_; }
    std::mutex mu_;
    std::vector<std::pair<void *, size_t>> free_small_, free_large_;
    std::map<void *, size_t> owner_;
};

// Host <-> device copies of set-up tables from / to the caller's or the library's PAGEABLE memory.  The runtime moves small
// transfers through its own staging buffers, but pins the user pages for transfers from 4 MB up (GPU_PINNED_MIN_XFER_SIZE) and
// unpins them afterwards; every set-up table is a temporary std::vector that is freed right after.  Unmapping host memory the
// driver knows about is what stalls the GPU (host_arena.h: 14-25 ms with no stream making progress), so
// medium-sized copies go in pieces below the pinning threshold and never register the vector at all; huge ones (the set-up of a
// 10 M-factor problem takes 0.3 s anyway) are left to the runtime.
inline hipError_t copy_chunked(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    const size_t kPin = (size_t)4 << 20, kPiece = (size_t)2 << 20, kHuge = (size_t)256 << 20;
    if (bytes < kPin || bytes > kHuge) return hipMemcpy(dst, src, bytes, kind);
    for (size_t o = 0; o < bytes; o += kPiece) {
        const hipError_t e = hipMemcpy((char *)dst + o, (const char *)src + o, bytes - o < kPiece ? bytes - o : kPiece, kind);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
inline hipError_t copy_h2d(void *dst, const void *src, size_t bytes) { return copy_chunked(dst, src, bytes, hipMemcpyHostToDevice); }
inline hipError_t copy_d2h(void *dst, const void *src, size_t bytes) { return copy_chunked(dst, src, bytes, hipMemcpyDeviceToHost); }

// One process-wide piece of pinned, device-visible host memory for SMALL transfers, read / written by kernels directly
// (zero-copy) instead of by a copy: the pose array of a refinement (30 KB for a window stage) goes up once and comes down once
// per call, on handles that live for a few milliseconds; through this stage that is a memcpy and a kernel reading over the host
// link -- no copy-engine submission, no pinning of the caller's pages, nothing to unpin.  Pinned and mapped once per process.
// The stage is cut into kSlots slots: lock(bytes) hands out a free one (several host threads drive windows concurrently, each
// holds its slot across its own stream synchronise without stopping the others; a caller waits only when all slots are taken),
// unlock(p) gives it back.  Requests larger than a slot get nullptr (the caller takes the plain copy).
class HostStage {
  public:
    static constexpr size_t kSlots = 8, kSlotBytes = (size_t)256 << 10, kBytes = kSlotBytes;
    static HostStage &get()
    {
        static HostStage s;
        return s;
    }
    // nullptr: no staging memory for this request (the caller takes the plain copy)
    void *lock(size_t bytes)
    {
        if (bytes > kSlotBytes) return nullptr;
        std::unique_lock<std::mutex> lk(mu_);
        if (!p_ && !failed_) {
            if (hipHostMalloc(&p_, kSlots * kSlotBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p_ = nullptr; failed_ = true; }
        }
        if (!p_) return nullptr;
        cv_.wait(lk, [&] { return busy_ != (1u << kSlots) - 1u; });
        unsigned k = 0;
        while (busy_ & (1u << k)) ++k;
        busy_ |= 1u << k;
        return static_cast<char *>(p_) + k * kSlotBytes;
    }
    void unlock(void *slot)
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            busy_ &= ~(1u << (unsigned)((static_cast<char *>(slot) - static_cast<char *>(p_)) / kSlotBytes));
        }
        cv_.notify_one();
    }

  private:
    std::mutex mu_;
    std::condition_variable cv_;
    unsigned busy_ = 0;
    void *p_ = nullptr;
    bool failed_ = false;
};

// Scoped workspace buffer bound to the stream that uses it: the destructor drains the stream before the block can be
// handed to anybody else.
struct DevBuf {
    void *p = nullptr;
    hipStream_t stream = nullptr;
    DevBuf() {}
    explicit DevBuf(hipStream_t s) : stream(s) {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf()
    {
        if (!p) return;
        (void)hipStreamSynchronize(stream);
        DevicePool::get().free(p);
    }
    hipError_t alloc(size_t bytes) { return DevicePool::get().alloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() const { return (T *)p; }
    void *release() { void *q = p; p = nullptr; return q; }
};

} // namespace lvba
