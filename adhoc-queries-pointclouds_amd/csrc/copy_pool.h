// copy_pool.h — a few helper threads that fill a pinned staging buffer in parallel slices.
//
// The host-block path (pcq_scan_host / pcq_scan_fd) is bound by how fast host bytes reach the pinned
// buffer the DMA engine reads from: one thread's memcpy / pread from the page cache runs at about half
// the PCIe Gen5 x16 rate (profiles/r01_host_path_rate.json), so the copy is split over the calling
// thread plus `helpers` workers.  Fork-join per staging chunk; no work is queued across calls.
#pragma once

#include <errno.h>
#include <pthread.h>
#include <sched.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

class CopyPool {
public:
    // result of run(): 0, or 1 = short read (end of file), or -errno of a failed pread
    // `cpus` (may be null): the CPUs the helpers are pinned to — the GPU's NUMA node, so that the pinned
    // staging buffer they fill is written by local cores
    explicit CopyPool(int helpers, const cpu_set_t *cpus = nullptr) {
        for (int i = 0; i < helpers; i++) {
            workers_.emplace_back([this] { loop(); });
            if (cpus && CPU_COUNT(cpus) > 0) (void)pthread_setaffinity_np(workers_.back().native_handle(), sizeof(cpu_set_t), cpus);
        }
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto &t : workers_) t.join();
    }
    CopyPool(const CopyPool &) = delete;
    CopyPool &operator=(const CopyPool &) = delete;

    int helpers() const { return (int)workers_.size(); }

    int run(int fd, uint8_t *dst, const uint8_t *src, size_t bytes) {
        constexpr size_t kMinSlice = 1u << 20;
        size_t parts = bytes / kMinSlice;
        if (parts > workers_.size() + 1) parts = workers_.size() + 1;
        if (parts <= 1) return copy(fd, dst, src, bytes);
        Job job;
        job.fd = fd, job.dst = dst, job.src = src, job.bytes = bytes;
        job.slice = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        job.nslices = (bytes + job.slice - 1) / job.slice;
        {
            // A job is published only while no helper is inside work(): a helper still spinning on the tickets of
            // the previous job would otherwise draw a ticket of this one and copy it with the old job's geometry.
            std::unique_lock<std::mutex> g(m_);
            idle_.wait(g, [this] { return active_ == 0; });
            job_ = job;
            next_.store(0, std::memory_order_relaxed);
            pending_ = job.nslices;
            result_ = 0;
            generation_++;
        }
        wake_.notify_all();
        work(job);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        return result_;
    }

private:
    struct Job {
        int fd = -1;
        uint8_t *dst = nullptr;
        const uint8_t *src = nullptr;
        size_t bytes = 0, slice = 0, nslices = 0;
    };
    static int copy(int fd, uint8_t *dst, const uint8_t *src, size_t bytes) {
        if (fd < 0) {
            memcpy(dst, src, bytes);
            return 0;
        }
        off_t off = (off_t)(uintptr_t)src;
        while (bytes) {
            const ssize_t r = pread(fd, dst, bytes, off);
            if (r < 0) {
                if (errno == EINTR) continue;
                return -errno;
            }
            if (r == 0) return 1;
            dst += r;
            off += r;
            bytes -= (size_t)r;
        }
        return 0;
    }
    // `job` is the caller's own copy of the descriptor, taken under the mutex in the generation it belongs to
    void work(const Job &job) {
        for (;;) {
            const size_t k = next_.fetch_add(1, std::memory_order_relaxed);
            if (k >= job.nslices) return;
            const size_t at = k * job.slice;
            const size_t len = job.bytes - at < job.slice ? job.bytes - at : job.slice;
            const int r = copy(job.fd, job.dst + at, job.src + at, len);
            std::lock_guard<std::mutex> g(m_);
            if (r && !result_) result_ = r;
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> g(m_);
                wake_.wait(g, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                job = job_;
                active_++;
            }
            work(job);
            std::lock_guard<std::mutex> g(m_);
            if (--active_ == 0) idle_.notify_all();
        }
    }

    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_, idle_;
    bool stop_ = false;
    uint64_t generation_ = 0;
    Job job_;
    size_t pending_ = 0;
    int active_ = 0;  // helpers inside work()
    std::atomic<size_t> next_{0};
    int result_ = 0;
};
