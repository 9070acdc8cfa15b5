// pcq_host.hpp — C++ host layer above the C ABI of include/pcq.h.
//
// The reference's host code is Rust; no Rust toolchain exists in the build image, so the host side
// is written in C++ and mirrors the reference's operator interface for this path — same names,
// argument meaning and error behaviour — so that tests read like tests of the reference:
//
//   readers::Point                                   readers/src/lib.rs:10-19
//   trait ResultCollector + Count/Buffer/GridSampled query/src/collect_points.rs:7-127
//   trait Searcher, BoundsSearcher, ClassSearcher,
//   enum SearchImplementation                        query/src/search/searcher.rs:19-152
//   search_{last,las}_file_by_{bounds,classification}_optimized
//                                                    query/src/search/last.rs:46-166, 213-293
//                                                    query/src/search/las.rs:52-148, 192-261
//   search_lazer_file_by_{bounds,classification}     query/src/search/lazer.rs:34-116
//   trait PointDumper, IgnoreDumper, FileDumper      query/src/dump_points.rs:13-121
//   get_all_input_files, parse_aabb, get_total_bounds, run_search_sequential,
//   run_search_parallel, is_valid_file               query/src/main.rs:29-189
//
// Everything that touches point data goes through libpcq.so (pcq_scan_host & co.); this layer only
// opens/mmaps files, parses LAS headers, converts the query box and drains collectors.  There is no
// CPU implementation of the scans here.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "pcq.h"

namespace pcq {

// anyhow::Result<()> stand-in.  `panic` marks conditions where the reference panics (exit 101).
struct Status {
    int code = PCQ_OK;
    std::string message;
    bool panic = false;
    bool ok() const { return code == PCQ_OK; }
    static Status Ok() { return {}; }
    static Status Err(int code, std::string msg) { return {code, std::move(msg), false}; }
    static Status Panic(std::string msg) { return {PCQ_ERR_PANIC, std::move(msg), true}; }
    static Status FromLib(int code);  // picks up pcq_last_error()
};

using Point = pcq_point;  // readers::Point, 31 bytes packed

// pasture_core::math::AABB<f64>
struct AABB {
    double min[3];
    double max[3];
    // AABB::from_min_max panics when min > max on an axis [recalled, pasture-core 0.1.0]
    static Status from_min_max(const double mn[3], const double mx[3], AABB *out);
    static AABB from_min_max_unchecked(const double mn[3], const double mx[3]);
    bool intersects(const AABB &o) const;           // inclusive
    static AABB union_of(const AABB &a, const AABB &b);
};

// las::raw::Header fields consumed by the path (layout: query/src/las.rs:7-40).
struct LasHeader {
    uint8_t version_major = 0, version_minor = 0;
    uint16_t header_size = 0;
    uint32_t offset_to_point_data = 0;
    uint8_t point_data_record_format = 0;  // after optional masking
    uint16_t point_data_record_length = 0;
    uint64_t number_of_points = 0;         // Header::number_of_points()
    double scale[3] = {0, 0, 0}, offset[3] = {0, 0, 0};
    AABB bounds{};                          // Header::bounds()
};
// raw::Header::read_from + Header::from_raw; mask_format applies `&= 0b1111` first (last.rs:222).
Status parse_las_header(const uint8_t *data, size_t len, bool mask_format, LasHeader *out);

// Read-only mmap of a whole file (last.rs:27-34).
class MappedFile {
public:
    MappedFile() = default;
    ~MappedFile();
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    Status open(const std::string &path);
    const uint8_t *data() const { return data_; }
    size_t size() const { return size_; }
    int fd() const { return fd_; }  // kept open: column blocks are streamed with pread (pcq_scan_fd)
    // which file this was when it was opened: device, inode, size, modification time (st_mtim)
    uint64_t dev() const { return dev_; }
    uint64_t ino() const { return ino_; }
    int64_t mtime_ns() const { return mtime_ns_; }

private:
    const uint8_t *data_ = nullptr;
    size_t size_ = 0;
    int fd_ = -1;
    uint64_t dev_ = 0, ino_ = 0;
    int64_t mtime_ns_ = 0;
};

// One GPU context per (thread, device); created on first use, destroyed with the thread.
Status thread_context(int device, pcq_ctx **out);
// The `query` binary ends with its query: its threads then leave their contexts to the end of the process instead of releasing
// streams, events, pinned and device memory one by one (main.cpp).  Off by default: a library user's threads release what
// they created.
void contexts_die_with_the_process(bool yes);
// Releases the calling thread's contexts now (no-op once contexts_die_with_the_process(true)): see core.cpp.
void release_thread_contexts();

// ---- collect_points.rs ------------------------------------------------------------------------
// The per-match `collect_one(Point)` callback of the reference does not exist here: matches are
// pushed into the device-resident collector by the scan kernels (a per-match callback across the
// boundary is the CPU bottleneck this path removes).  The drain side is identical.
class ResultCollector {
public:
    virtual ~ResultCollector();
    // ResultCollector::points  (None for the count collector)
    virtual std::optional<std::vector<Point>> points();
    // ResultCollector::points_ref (Some only for the buffer collector)
    virtual const std::vector<Point> *points_ref();
    virtual Status point_count(size_t *out);
    // A collector that is kept while other files are searched takes its compact form now (run_search_parallel keeps one per
    // file until all are done, main.rs:153-161).  Nothing to do except for the grid collector.
    virtual Status file_done() { return Status::Ok(); }
    bool has_points() const { return pcq_collector_has_points(handle_) != 0; }
    pcq_collector *handle() const { return handle_; }
    pcq_ctx *context() const { return ctx_; }
    uint64_t next_index = 0;  // file-order index of the next scanned point (first-seen-wins bookkeeping)

protected:
    pcq_ctx *ctx_ = nullptr;
    pcq_collector *handle_ = nullptr;
    std::vector<Point> cache_;
    bool cached_ = false;
    Status fetch();
};

class CountCollector : public ResultCollector {  // collect_points.rs:72-98
public:
    // device_counter (optional): 8 bytes in the context's HBM that this collector ADDS to instead of owning a
    // counter — the per-GPU counter all files of that GPU accumulate into (main.rs:164-180 as one all-reduce)
    static Status create(pcq_ctx *ctx, std::unique_ptr<ResultCollector> *out, uint64_t *device_counter = nullptr);
};
class BufferCollector : public ResultCollector {  // collect_points.rs:14-44
public:
    static Status create(pcq_ctx *ctx, std::unique_ptr<ResultCollector> *out);
    std::optional<std::vector<Point>> points() override;
    const std::vector<Point> *points_ref() override;
};
class GridSampledCollector : public ResultCollector {  // collect_points.rs:100-127
public:
    static Status create(pcq_ctx *ctx, const AABB &bounds, double cell_size, std::unique_ptr<ResultCollector> *out);
    std::optional<std::vector<Point>> points() override;
    // folds the file's matches into per-cell winners: what the reference's HashMap holds at this point (grid_sampling.rs:72-103);
    // unfolded, the device keeps a tuple per scanned point
    Status file_done() override { return Status::FromLib(pcq_collector_flush(handle_)); }
};

// ---- search/last.rs, search/las.rs ---------------------------------------------------------------
struct SearchLog {  // side output the reference prints from inside the scans
    int las_record_size = -1;  // las.rs:73 `println!("Point record size: {}")`
};
// What the host knows about one input file before any GPU work — the prologue of the four optimized searches:
// open + mmap, header parse, block offsets, the header-AABB early-out (last.rs:92-94), the f64 -> local integer box.
// A query whose files are all resolved here (skipped, empty, or in error) never wakes the GPU.
struct FilePlan {
    Status status;               // an error the reference raises before its per-point loop
    bool needs_gpu = false;      // false: resolved on the host
    int las_record_size = -1;    // las.rs:73 (printed even for a file that is then skipped)
    std::string path;            // opened again by execute_plan, for the duration of the scan
    uint64_t file_size = 0;      // as the prologue saw it; and WHICH file it saw — execute_plan opens the path again and the
    uint64_t file_dev = 0, file_ino = 0;  // offsets, scale and point count of the plan hold for that file only: one that was
    int64_t file_mtime_ns = 0;            // replaced or rewritten in between is an error, not a scan with the old header
    pcq_columns cols{};          // the column "pointers" are byte offsets into the file (pcq_scan_fd)
    pcq_predicate pred{};
};
FilePlan plan_last_file_by_bounds_optimized(const std::string &path, const AABB &bounds);
FilePlan plan_last_file_by_classification_optimized(const std::string &path, uint8_t cls);
FilePlan plan_las_file_by_bounds_optimized(const std::string &path, const AABB &bounds);
FilePlan plan_las_file_by_classification_optimized(const std::string &path, uint8_t cls);
Status execute_plan(FilePlan &plan, ResultCollector &rc);  // the per-point loop: pcq_scan_fd into the collector
Status search_last_file_by_bounds_optimized(const std::string &path, const AABB &bounds, ResultCollector &rc);
Status search_last_file_by_classification_optimized(const std::string &path, uint8_t cls, ResultCollector &rc);
Status search_las_file_by_bounds_optimized(const std::string &path, const AABB &bounds, ResultCollector &rc, SearchLog *log);
Status search_las_file_by_classification_optimized(const std::string &path, uint8_t cls, ResultCollector &rc);
// search/lazer.rs:34-116 over readers/src/lazer_reader.rs (one implementation for both --optimized settings)
Status search_lazer_file_by_bounds(const std::string &path, const AABB &bounds, ResultCollector &rc);
Status search_lazer_file_by_classification(const std::string &path, uint8_t cls, ResultCollector &rc);
Status lazer_file_bounds(const std::string &path, AABB *out);  // LAZERSource::from(..).get_metadata().bounds()

// ---- search/searcher.rs ----------------------------------------------------------------------------
enum class SearchImplementation { Regular, Optimized };

class Searcher {
public:
    virtual ~Searcher() = default;
    virtual Status search_file(const std::string &path, SearchImplementation impl, ResultCollector &collector,
                               SearchLog *log = nullptr) const = 0;
    // the host-only part of search_file where there is one (nullopt: search_file does everything)
    virtual std::optional<FilePlan> plan_file(const std::string &path, SearchImplementation impl) const = 0;
};
class BoundsSearcher : public Searcher {
public:
    explicit BoundsSearcher(const AABB &bounds) : bounds_(bounds) {}
    Status search_file(const std::string &path, SearchImplementation impl, ResultCollector &collector,
                       SearchLog *log = nullptr) const override;
    std::optional<FilePlan> plan_file(const std::string &path, SearchImplementation impl) const override;

private:
    AABB bounds_;
};
class ClassSearcher : public Searcher {
public:
    explicit ClassSearcher(uint8_t cls) : class_(cls) {}
    Status search_file(const std::string &path, SearchImplementation impl, ResultCollector &collector,
                       SearchLog *log = nullptr) const override;
    std::optional<FilePlan> plan_file(const std::string &path, SearchImplementation impl) const override;

private:
    uint8_t class_;
};

// ---- dump_points.rs ----------------------------------------------------------------------------------
class PointDumper {
public:
    virtual ~PointDumper() = default;
    virtual Status dump_points(const Point *points, size_t n) = 0;
    virtual size_t num_dumped_points() const = 0;
    // false: only the NUMBER of points is used (IgnoreDumper, dump_points.rs:28-33) — the driver then
    // asks the collector for its count instead of copying every record out of HBM
    virtual bool wants_points() const { return true; }
    // true when points()/points_ref() is Some for this collector kind (main.rs:135-141)
};
class IgnoreDumper : public PointDumper {
public:
    bool wants_points() const override { return false; }
    Status dump_points(const Point *, size_t n) override {
        dumped_ += n;
        return Status::Ok();
    }
    size_t num_dumped_points() const override { return dumped_; }

private:
    size_t dumped_ = 0;
};
class FileDumper : public PointDumper {
public:
    // `print` receives the "Writing N points" line (dump_points.rs:108); default: stdout
    static Status create(const std::string &root_dir, std::unique_ptr<PointDumper> *out,
                         std::function<void(const std::string &)> print = nullptr);
    Status dump_points(const Point *points, size_t n) override;
    size_t num_dumped_points() const override { return dumped_; }

private:
    std::string root_;
    std::function<void(const std::string &)> print_;
    size_t file_index_ = 0, dumped_ = 0;
};

// ---- main.rs -----------------------------------------------------------------------------------------
Status get_all_input_files(const std::string &input, std::vector<std::string> *out);
bool is_valid_file(const std::string &path);
Status parse_aabb(const std::string &s, AABB *out);
Status get_total_bounds(const std::vector<std::string> &files, AABB *out);

// Creates the collector for one worker: Result<Box<dyn ResultCollector>> of main.rs:24.
// `shared_counter` (may be null): the per-GPU device counter a count collector should add to.
using CollectorFactoryFn = std::function<Status(pcq_ctx *, uint64_t *shared_counter, std::unique_ptr<ResultCollector> *)>;

struct FileStat {  // filled by the drivers when RunOptions::stats is set (extra flag --stats-json)
    std::string path;
    int device = 0;
    double search_ms = 0;
};
struct RunOptions {
    std::vector<int> devices = {0};  // GPUs to use; files are the independent units (main.rs:153-161)
    // host threads feeding each GPU in --parallel mode; 0 = chosen by the driver.  One is the measured optimum for count
    // queries and for large files: the library splits the staging copy over its own helper threads and reaches the PCIe rate
    // from a single caller, while a second context on the same GPU costs another 20-40 ms of start-up.  A collector that
    // yields points ends every file on a synchronisation (a grid folds to its winners), which drains a single thread's
    // pipeline for ~1 ms per file: with many small files a second thread per GPU fills those gaps (64 files x 2 M points,
    // --density: 229 -> 200 ms; 16 x 20 M: no difference; 4 files: 20 ms worse — profiles/r03_density_threads.log).
    int threads_per_device = 0;
    // false: the factory makes count collectors (points() is None, main.rs:171-179) — the parallel driver then gives
    // every GPU one device counter and merges the counters with one all-reduce
    bool collectors_yield_points = true;
    // the factory makes grid collectors (--density): every file ends on a fold, i.e. on a synchronisation (see threads_per_device)
    bool collectors_fold_per_file = false;
    std::vector<FileStat> *stats = nullptr;
    // Test hooks, set only by the test entry of the C view (capi.cpp: pcq_query_main_with_hooks) — nothing in the `query`
    // binary reads the environment for them.  device_slots: the device list as given, repeats allowed ("0,0" = two device
    // SLOTS on one physical GPU, each with its own workers, contexts and counter block, merged like two GPUs; RCCL refuses
    // a communicator over a repeated device, which is the failure the merge's fallback exists for).  allreduce_fail: make
    // the count merge's collective fail through the real RCCL calls, 1 = before anything is touched, 2 = after the reduction.
    std::vector<int> test_device_slots;
    int test_allreduce_fail = 0;
};

// Files of a parallel query over device slots.  main.rs:153-161 hands files to whichever rayon thread is free; with a GPU
// behind every slot, a slot whose context is still coming up (50-230 ms of HIP start-up) must not find the queue drained by
// the slot that was ready first — and a slot that is ready must not idle next to files nobody has started.  So every slot
// starts with its OWN share, longest-processing-time first by planned points (the rule sharding.py applies across
// processes: equal files give file i -> slot i % N), takes its own files largest first, and once its share is gone takes
// the smallest file left on the slot with the most work left.  Thread-safe.
class FileScheduler {
  public:
    static constexpr size_t npos = (size_t)-1;
    FileScheduler(const std::vector<uint64_t> &cost, size_t nslots);
    size_t next(size_t slot);                                   // a file for a worker of `slot`, or npos when every file is taken
    size_t home_slot(size_t file) const { return home_[file]; }  // the slot the file was assigned to at the start
  private:
    struct Impl;
    std::shared_ptr<Impl> impl_;
    std::vector<size_t> home_;
};
// The schedule `nslots` workers (one per slot) produce when slot k's context is ready at ready_ms[k] and a file costs
// ms_per_unit x cost: slot_of_file[i] = the slot that scanned file i; returns the time the last worker finishes.
double simulate_schedule(const std::vector<uint64_t> &cost, const std::vector<double> &ready_ms, double ms_per_unit, std::vector<int> *slot_of_file);

// stdout lines go through `print` (so tests can capture them).
using PrintFn = std::function<void(const std::string &)>;
Status run_search_sequential(const std::vector<std::string> &files, const Searcher &searcher, SearchImplementation impl,
                             const CollectorFactoryFn &factory, PointDumper &dumper, const RunOptions &opt, const PrintFn &print);
Status run_search_parallel(const std::vector<std::string> &files, const Searcher &searcher, SearchImplementation impl,
                           const CollectorFactoryFn &factory, PointDumper &dumper, const RunOptions &opt, const PrintFn &print);

// ---- a dataset resident in HBM (resident.cpp) -----------------------------------------------------------
// Not in the reference (it re-reads the files for every query): the positions and classification blocks of a set of
// LAST files are loaded into one GPU's HBM once, and every count query over them is ONE batched launch.
struct ResidentFile {
    std::string path;
    LasHeader header;
    void *xyz = nullptr, *cls = nullptr;  // device blocks
};
class ResidentDataset {
public:
    ~ResidentDataset();
    ResidentDataset(const ResidentDataset &) = delete;
    ResidentDataset &operator=(const ResidentDataset &) = delete;
    static Status load(pcq_ctx *ctx, const std::vector<std::string> &paths, std::unique_ptr<ResidentDataset> *out);
    Status count_bounds(const AABB &bounds, uint64_t *matches, uint64_t *points_scanned = nullptr);
    Status count_class(uint8_t cls, uint64_t *matches, uint64_t *points_scanned = nullptr);
    size_t files() const { return files_.size(); }
    uint64_t points() const { return points_; }

private:
    ResidentDataset() = default;
    Status run(const std::vector<pcq_columns> &cols, const std::vector<pcq_predicate> &preds, uint64_t *matches);
    pcq_ctx *ctx_ = nullptr;
    std::vector<ResidentFile> files_;
    uint64_t *counter_ = nullptr;
    uint64_t points_ = 0;
};

// The whole CLI (main.rs:191-319): returns the process exit code; stdout/stderr text via callbacks.
int query_main(int argc, const char *const *argv, const PrintFn &out, const PrintFn &err, const RunOptions *test_hooks = nullptr);

}  // namespace pcq
