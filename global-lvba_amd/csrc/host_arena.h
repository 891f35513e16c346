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
// application's allocator is left alone.  lvba_release_cached_memory() returns the cached blocks; at most kCap bytes are kept.
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
    static constexpr size_t kMin = (size_t)64 << 10, kCap = (size_t)2 << 30;
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
    void *alloc(size_t bytes)
    {
        if (bytes < kMin) return std::malloc(bytes ? bytes : 1);
        const size_t cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = free_.find(cls);
            if (it != free_.end() && !it->second.empty()) {
                void *p = it->second.back();
                it->second.pop_back();
                cached_ -= cls;
                return p;
            }
        }
        return std::malloc(cls);
    }
    void free(void *p, size_t bytes)
    {
        if (!p) return;
        if (bytes < kMin) { std::free(p); return; }
        const size_t cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(mu_);
            if (cached_ + cls <= kCap) {
                free_[cls].push_back(p);
                cached_ += cls;
                return;
            }
        }
        std::free(p);
    }
    size_t release()
    {
        std::lock_guard<std::mutex> g(mu_);
        const size_t n = cached_;
        for (auto &kv : free_) {
            for (void *p : kv.second) std::free(p);
            kv.second.clear();
        }
        cached_ = 0;
        return n;
    }
    size_t cached_bytes()
    {
        std::lock_guard<std::mutex> g(mu_);
        return cached_;
    }

  private:
    std::mutex mu_;
    std::map<size_t, std::vector<void *>> free_;
    size_t cached_ = 0;
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
