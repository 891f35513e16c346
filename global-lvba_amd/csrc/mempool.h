// mempool.h -- caching device allocator for short-lived workspaces.
// hipMalloc / hipFree of 100 MB-class buffers cost 0.2-0.5 ms each and hipFree synchronises the device; the voxel
// front-end needs ~40 temporaries per map.  Blocks are cached per device in size classes (3 mantissa bits: <= 12.5 %
// slack) and handed back out; lvba_release_cached_memory() returns them to the driver.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
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
