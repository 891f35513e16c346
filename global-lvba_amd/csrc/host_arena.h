// host_arena.h -- library-owned host memory for the set-up tables (header-only, no HIP).
//
// Host memory that goes back to the operating system while GPU queues are running stalls them: measured on the window stage's
// joint problem, freeing the set-up's host tables (std::vectors of 2-9 MB, which glibc serves with mmap and returns with munmap)
// was followed by 14-25 ms in which NO stream of the process got anything done -- the unmapping runs the kernel driver's MMU
// notifier, which evicts the process's queues and restores them a moment later -- six times the three LM iterations the set-up
// was for (47 -> 19 ms for 16 windows with the frees skipped).  Round 2 answered that by retuning the PROCESS's allocator
// (mallopt), which a drop-in library inside somebody's ROS node has no business doing.  Now the tables live in blocks the
// library owns: HostArena hands out malloc'ed blocks in size classes (<= 12.5 % slack) and takes them back into its own free
// lists instead of giving them to free() -- the pages stay mapped and are reused by the next set-up; the embedding
// application's allocator is left alone.  lvba_release_cached_memory() returns the cached blocks; at most cap() bytes are kept
// (512 MB; LVBA_HOST_CACHE_MB).
// Small blocks (< kMin) are plain malloc / free: they come from the heap, not from mappings of their own.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <new>
#include <vector>

namespace lvba {

class HostArena {
  public:
    // kMin: below it a block is plain malloc / free.  The cache keeps at most cap() bytes: 512 MB unless LVBA_HOST_CACHE_MB says
    // otherwise (read once; 0 turns the cache off).  What is kept stays mapped until lvba_release_cached_memory() -- a long-lived
    // process that has finished its refinements should call it.
    static constexpr size_t kMin = (size_t)64 << 10, kDefaultCap = (size_t)512 << 20, kHdr = 64;
    static HostArena &get()
    {
        static HostArena *a = new HostArena(); // never destroyed: blocks may be handed back during static destruction
        return *a;
    }
    static size_t size_class(size_t bytes)
    {
        int hb = 63 - __builtin_clzll((unsigned long long)bytes);
        const size_t step = (size_t)1 << (hb - 3);
        return (bytes + step - 1) / step * step;
    }
    size_t cap() const { return cap_; }
    void set_cap(size_t bytes) // tests
    {
        std::lock_guard<std::mutex> g(mu_);
        cap_ = bytes;
    }
    void *alloc(size_t bytes)
    {
        if (bytes < kMin) return std::malloc(bytes ? bytes : 1);
        const size_t cls = size_class(bytes);
        {
            // the smallest cached block that holds the request, with bounded slack (a block of up to twice the class): a process
            // whose problem sizes vary reuses what it has instead of keeping one set of blocks per class
            std::lock_guard<std::mutex> g(mu_);
            for (auto it = free_.lower_bound(cls); it != free_.end() && it->first <= 2 * cls; ++it) {
                if (it->second.empty()) continue;
                void *raw = it->second.back();
                it->second.pop_back();
                cached_ -= it->first;
                return static_cast<char *>(raw) + kHdr;
            }
        }
        void *raw = std::malloc(cls + kHdr);
        if (!raw) return nullptr;
        *static_cast<size_t *>(raw) = cls; // the block's real class, whatever size its user asks to free
        return static_cast<char *>(raw) + kHdr;
    }
    void free(void *p, size_t bytes)
    {
        if (!p) return;
        if (bytes < kMin) { std::free(p); return; }
        void *raw = static_cast<char *>(p) - kHdr;
        const size_t cls = *static_cast<size_t *>(raw);
        {
            std::lock_guard<std::mutex> g(mu_);
            if (cached_ + cls <= cap_) {
                free_[cls].push_back(raw);
                cached_ += cls;
                return;
            }
        }
        std::free(raw);
    }
    size_t release()
    {
        std::lock_guard<std::mutex> g(mu_);
        const size_t n = cached_;
        for (auto &kv : free_) {
            for (void *p : kv.second) std::free(p);
            kv.second.clear();
        }
        free_.clear();
        cached_ = 0;
        return n;
    }
    size_t cached_bytes()
    {
        std::lock_guard<std::mutex> g(mu_);
        return cached_;
    }

  private:
    HostArena()
    {
        if (const char *e = std::getenv("LVBA_HOST_CACHE_MB")) {
            char *end = nullptr;
            const long long mb = std::strtoll(e, &end, 10);
            if (end != e && mb >= 0) cap_ = (size_t)mb << 20;
        }
    }
    std::mutex mu_;
    std::map<size_t, std::vector<void *>> free_;
    size_t cached_ = 0, cap_ = kDefaultCap;
};

template <class T>
struct ArenaAlloc {
    typedef T value_type;
    ArenaAlloc() noexcept {}
    template <class U> ArenaAlloc(const ArenaAlloc<U> &) noexcept {}
    T *allocate(size_t n)
    {
        void *p = HostArena::get().alloc(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t n) noexcept { HostArena::get().free(p, n * sizeof(T)); }
    template <class U> bool operator==(const ArenaAlloc<U> &) const noexcept { return true; }
    template <class U> bool operator!=(const ArenaAlloc<U> &) const noexcept { return false; }
};

// the vector type of every set-up table of the library
template <class T> using hvec = std::vector<T, ArenaAlloc<T>>;

} // namespace lvba
