// parallel.h -- a parallel loop over independent items for the host-side scene build.
//
// The reference builds its scene on one thread (model.cpp, main.cpp:268-343).  The steps that dominate that
// build -- the sort at every node of the hierarchy, the world-space box of every instance -- are independent
// of each other item by item, so they run on several threads here; what each item computes, and therefore
// every emitted byte, is the same as on one thread.
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace rt {

// Threads the scene build may use: $RT_HOST_THREADS, else the machine's hardware threads, at most 16 (a GPU
// box hands every GPU's process a share of its cores, and eight ranks build their scenes side by side).
inline unsigned host_threads()
{
    if (const char *env = std::getenv("RT_HOST_THREADS")) {
        const long n = std::strtol(env, nullptr, 10);
        if (n >= 1) return unsigned(n > 256 ? 256 : n);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw < 1 ? 1 : hw > 16 ? 16 : hw;
}

// body(i) for every i in [0, count), items handed out one at a time to up to host_threads() threads (in
// place when there is one item or one thread).  The items must not depend on each other.  The first
// exception an item throws is re-thrown here after all threads have finished.
template <typename Body>
void parallel_for(size_t count, Body body)
{
    const size_t threads = count < host_threads() ? count : host_threads();
    if (threads <= 1) {
        for (size_t i = 0; i < count; i++) body(i);
        return;
    }
    std::atomic<size_t> next(0);
    std::exception_ptr failure;
    std::mutex failure_lock;
    auto work = [&]() {
        for (size_t i = next.fetch_add(1); i < count; i = next.fetch_add(1)) {
            try {
                body(i);
            } catch (...) {
                std::lock_guard<std::mutex> guard(failure_lock);
                if (!failure) failure = std::current_exception();
                next.store(count);  // no further items
            }
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    try {
        for (size_t t = 1; t < threads; t++) pool.emplace_back(work);
    } catch (...) {  // no more threads to be had: the ones there are, and this one, do the work
    }
    work();
    for (std::thread &t : pool) t.join();
    if (failure) std::rethrow_exception(failure);
}

}  // namespace rt
