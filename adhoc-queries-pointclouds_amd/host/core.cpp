// core.cpp — Status, AABB, LAS header parsing, mmap, per-thread GPU contexts.
#include "pcq_host.hpp"

#include <atomic>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <map>
#include <memory>

namespace pcq {

Status Status::FromLib(int code) {
    if (code == PCQ_OK) return Ok();
    Status s = Err(code, pcq_last_error());
    s.panic = code == PCQ_ERR_PANIC;
    return s;
}

// ---- AABB (pasture-core 0.1.0 [recalled]) ----------------------------------------------------------
Status AABB::from_min_max(const double mn[3], const double mx[3], AABB *out) {
    for (int a = 0; a < 3; a++)
        if (mn[a] > mx[a]) return Status::Panic("AABB::from_min_max: Minimum position must be <= maximum position!");
    *out = from_min_max_unchecked(mn, mx);
    return Status::Ok();
}
AABB AABB::from_min_max_unchecked(const double mn[3], const double mx[3]) {
    AABB b;
    for (int a = 0; a < 3; a++) b.min[a] = mn[a], b.max[a] = mx[a];
    return b;
}
bool AABB::intersects(const AABB &o) const {
    for (int a = 0; a < 3; a++)
        if (!(min[a] <= o.max[a] && max[a] >= o.min[a])) return false;
    return true;
}
AABB AABB::union_of(const AABB &a, const AABB &b) {
    AABB r;
    for (int i = 0; i < 3; i++) {
        r.min[i] = a.min[i] < b.min[i] ? a.min[i] : b.min[i];
        r.max[i] = a.max[i] > b.max[i] ? a.max[i] : b.max[i];
    }
    return r;
}

// ---- LAS header: las 0.7.4 raw::Header::read_from + Header::from_raw [recalled]; byte layout as in
//      query/src/las.rs:7-40 --------------------------------------------------------------------------
namespace {
struct Reader {
    const uint8_t *p;
    size_t len, pos = 0;
    bool ok = true;
    template <typename T>
    T get() {
        T v{};
        if (len - pos < sizeof(T) || pos > len) {
            ok = false;
            pos = len;
            return v;
        }
        memcpy(&v, p + pos, sizeof(T));  // little-endian host
        pos += sizeof(T);
        return v;
    }
    void skip(size_t n) {
        if (len - pos < n || pos > len) {
            ok = false;
            pos = len;
        } else {
            pos += n;
        }
    }
};
constexpr uint16_t kFormatLen[11] = {20, 28, 26, 34, 57, 63, 30, 36, 38, 59, 67};
}  // namespace

Status parse_las_header(const uint8_t *data, size_t len, bool mask_format, LasHeader *h) {
    *h = LasHeader{};
    Reader r{data, len};
    char sig[4] = {0, 0, 0, 0};
    for (char &c : sig) c = (char)r.get<uint8_t>();
    if (!r.ok) return Status::Err(PCQ_ERR_HEADER, "failed to fill whole buffer");
    if (memcmp(sig, "LASF", 4) != 0) return Status::Err(PCQ_ERR_HEADER, "invalid file signature");
    r.skip(2 + 2 + 16);  // file_source_id, global_encoding, guid
    h->version_major = r.get<uint8_t>();
    h->version_minor = r.get<uint8_t>();
    r.skip(32 + 32 + 2 + 2);  // system id, generating software, creation day/year
    h->header_size = r.get<uint16_t>();
    h->offset_to_point_data = r.get<uint32_t>();
    r.skip(4);  // number of VLRs
    uint8_t fmt = r.get<uint8_t>();
    h->point_data_record_length = r.get<uint16_t>();
    const uint32_t legacy_count = r.get<uint32_t>();
    r.skip(20);  // points by return
    for (int a = 0; a < 3; a++) h->scale[a] = r.get<double>();
    for (int a = 0; a < 3; a++) h->offset[a] = r.get<double>();
    for (int a = 0; a < 3; a++) {  // max_x, min_x, max_y, min_y, max_z, min_z
        h->bounds.max[a] = r.get<double>();
        h->bounds.min[a] = r.get<double>();
    }
    const bool v13 = h->version_major > 1 || (h->version_major == 1 && h->version_minor >= 3);
    const bool v14 = h->version_major > 1 || (h->version_major == 1 && h->version_minor >= 4);
    uint64_t large_count = 0;
    if (v13) r.skip(8);  // start of waveform data packet record
    if (v14) {
        r.skip(8 + 4);  // first EVLR, number of EVLRs
        large_count = r.get<uint64_t>();
        r.skip(15 * 8);
    }
    if (!r.ok) return Status::Err(PCQ_ERR_HEADER, "failed to fill whole buffer");
    if (h->header_size > r.pos) {
        r.skip(h->header_size - r.pos);  // padding
        if (!r.ok) return Status::Err(PCQ_ERR_HEADER, "failed to fill whole buffer");
    }
    if (mask_format) fmt &= 0x0F;
    h->point_data_record_format = fmt;
    if (fmt > 10) return Status::Err(PCQ_ERR_HEADER, "invalid point format number: " + std::to_string(fmt));
    if (h->point_data_record_length < kFormatLen[fmt])
        return Status::Err(PCQ_ERR_HEADER, "point data record length " + std::to_string(h->point_data_record_length) +
                                               " too small for format " + std::to_string(fmt));
    if (fmt >= 6 && !v14)
        return Status::Err(PCQ_ERR_HEADER, "version " + std::to_string(h->version_major) + "." + std::to_string(h->version_minor) +
                                               " does not support point format " + std::to_string(fmt));
    h->number_of_points = legacy_count > 0 ? (uint64_t)legacy_count : large_count;
    return Status::Ok();
}

// ---- mmap ---------------------------------------------------------------------------------------------
MappedFile::~MappedFile() {
    if (data_) munmap(const_cast<uint8_t *>(data_), size_);
    if (fd_ >= 0) close(fd_);
}

Status MappedFile::open(const std::string &path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return Status::Err(PCQ_ERR_IO, path + ": " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) {
        const int e = errno;
        close(fd);
        return Status::Err(PCQ_ERR_IO, path + ": " + strerror(e));
    }
    size_ = (size_t)st.st_size;
    dev_ = (uint64_t)st.st_dev, ino_ = (uint64_t)st.st_ino;
    mtime_ns_ = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
    if (size_ > 0) {
        void *p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) {
            const int e = errno;
            close(fd);
            size_ = 0;
            return Status::Err(PCQ_ERR_IO, path + ": mmap: " + strerror(e));
        }
        data_ = (const uint8_t *)p;
    }
    fd_ = fd;
    return Status::Ok();
}

// ---- per-thread contexts ------------------------------------------------------------------------------
namespace {
std::atomic<bool> g_process_is_ending{false};
struct ThreadContexts {
    std::map<int, pcq_ctx *> by_device;
    ~ThreadContexts() {
        if (g_process_is_ending.load()) return;  // the `query` binary: the process ends right behind the query (main.cpp)
        for (auto &kv : by_device) pcq_shutdown(kv.second);
    }
};
}  // namespace

void contexts_die_with_the_process(bool yes) { g_process_is_ending.store(yes); }

namespace {
ThreadContexts &this_threads_contexts() {
    static thread_local ThreadContexts tc;
    return tc;
}
}  // namespace

// The calling thread's contexts, released NOW — from ordinary code, while every library the release calls into is whole.
// Left to the thread_local's destructor the same calls run in the thread-exit (or process-exit) phase: behind the
// destructors of thread-locals constructed later, which is where a preloaded profiler keeps the per-thread state its HIP
// interception uses (a `query` under rocprofv3 --memory-copy-trace printed its answer, wrote its traces and did not end:
// DESIGN.md section 9, profiles/r04_rocprof_query.log).  The drivers call this at the end of every worker and before main() returns; the destructor stays
// as the net under a library user's threads.
void release_thread_contexts() {
    if (g_process_is_ending.load()) return;
    ThreadContexts &tc = this_threads_contexts();
    for (auto &kv : tc.by_device) pcq_shutdown(kv.second);
    tc.by_device.clear();
}

Status thread_context(int device, pcq_ctx **out) {
    ThreadContexts &tc = this_threads_contexts();
    auto it = tc.by_device.find(device);
    if (it != tc.by_device.end()) {
        *out = it->second;
        return Status::Ok();
    }
    pcq_ctx *ctx = nullptr;
    int rc;
    {
        // One at a time PER DEVICE: several threads bringing up contexts on ONE GPU were measured to take longer in total than
        // back to back (runtime init, first queue creation: profiles/r01_cli_fixed_cost.log) — that is all the measurement says.
        // Different devices start together: with one process-wide mutex the eight contexts of `--gpus 8` came up one after
        // the other (8 x 50-230 ms) while the first GPU's worker was already scanning.
        static std::mutex table_mutex;
        static std::map<int, std::unique_ptr<std::mutex>> per_device;
        std::mutex *dev_mutex;
        {
            std::lock_guard<std::mutex> g(table_mutex);
            auto &slot = per_device[device];
            if (!slot) slot = std::make_unique<std::mutex>();
            dev_mutex = slot.get();
        }
        std::lock_guard<std::mutex> g(*dev_mutex);
        rc = pcq_init(device, &ctx);
    }
    if (rc) return Status::FromLib(rc);
    (void)pcq_prepare_host_scans(ctx);  // (the staging ring is pinned while the worker opens its first file: 8 ms off that file)
    tc.by_device[device] = ctx;
    *out = ctx;
    return Status::Ok();
}

}  // namespace pcq
