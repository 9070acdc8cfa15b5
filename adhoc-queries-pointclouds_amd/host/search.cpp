// search.cpp — collectors, the four optimized per-file searches and the Searcher dispatch.
//
// Each search function does on the host exactly what the reference does before its per-point loop
// (open + mmap, header parse, block offsets, file-level early-out, query box -> local integer box)
// and then hands the column blocks to libpcq.so, which replaces the loop and the collector pushes.
#include "pcq_host.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>

namespace pcq {

// ---- collectors (collect_points.rs) ---------------------------------------------------------------
ResultCollector::~ResultCollector() {
    if (handle_) pcq_collector_free(handle_);
}
std::optional<std::vector<Point>> ResultCollector::points() { return std::nullopt; }  // collect_points.rs:87-89
const std::vector<Point> *ResultCollector::points_ref() { return nullptr; }           // :91-93
Status ResultCollector::point_count(size_t *out) {
    uint64_t n = 0;
    const int rc = pcq_collector_point_count(handle_, &n);
    if (rc) return Status::FromLib(rc);
    *out = (size_t)n;
    return Status::Ok();
}
Status ResultCollector::fetch() {
    if (cached_) return Status::Ok();
    uint64_t n = 0;
    int rc = pcq_collector_points(handle_, nullptr, 0, &n);
    if (rc) return Status::FromLib(rc);
    cache_.resize((size_t)n);
    if (n) {
        rc = pcq_collector_points(handle_, cache_.data(), n, &n);
        if (rc) return Status::FromLib(rc);
    }
    cached_ = true;
    return Status::Ok();
}

Status CountCollector::create(pcq_ctx *ctx, std::unique_ptr<ResultCollector> *out, uint64_t *device_counter) {
    auto c = std::make_unique<CountCollector>();
    c->ctx_ = ctx;
    const int rc = device_counter ? pcq_collector_new_count_at(ctx, device_counter, &c->handle_) : pcq_collector_new_count(ctx, &c->handle_);
    if (rc) return Status::FromLib(rc);
    *out = std::move(c);
    return Status::Ok();
}
Status BufferCollector::create(pcq_ctx *ctx, std::unique_ptr<ResultCollector> *out) {
    auto c = std::make_unique<BufferCollector>();
    c->ctx_ = ctx;
    const int rc = pcq_collector_new_buffer(ctx, &c->handle_);
    if (rc) return Status::FromLib(rc);
    *out = std::move(c);
    return Status::Ok();
}
std::optional<std::vector<Point>> BufferCollector::points() {  // :33-35
    if (!fetch().ok()) return std::vector<Point>{};
    return cache_;
}
const std::vector<Point> *BufferCollector::points_ref() {  // :37-39
    if (!fetch().ok()) return nullptr;
    return &cache_;
}
Status GridSampledCollector::create(pcq_ctx *ctx, const AABB &bounds, double cell_size, std::unique_ptr<ResultCollector> *out) {
    auto c = std::make_unique<GridSampledCollector>();
    c->ctx_ = ctx;
    const int rc = pcq_collector_new_grid(ctx, bounds.min, bounds.max, cell_size, &c->handle_);
    if (rc) return Status::FromLib(rc);
    *out = std::move(c);
    return Status::Ok();
}
std::optional<std::vector<Point>> GridSampledCollector::points() {  // :116-118
    if (!fetch().ok()) return std::vector<Point>{};
    return cache_;
}

// ---- shared pieces of the four searches -------------------------------------------------------------
namespace {

std::optional<uint64_t> las_offset_to_color(uint8_t fmt) {  // las.rs:38-45, last.rs:83-88
    switch (fmt) {
    case 2: return 20;
    case 3: return 28;
    case 5: return 28;
    default: return std::nullopt;
    }
}

Status invalid_format(uint8_t fmt, const std::string &path) {
    return Status::Err(PCQ_ERR_FORMAT, "Invalid LAS format " + std::to_string(fmt) + " in file " + path);
}

Status eof() { return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer"); }

// A block [off, off + bytes) must lie inside the mapped file.  The reference's Cursor reads fail
// lazily with UnexpectedEof when a truncated byte is touched; here the blocks a scan may touch are
// validated up front ("deviations").
bool block_ok(const MappedFile &f, uint64_t off, uint64_t bytes) { return off <= f.size() && bytes <= f.size() - off; }

}  // namespace

// The column blocks were located in the mapped file (header parse, offsets: exactly as the reference does); the
// bytes themselves are streamed by the library with pread, so a plan carries FILE OFFSETS instead of addresses inside a
// mapping — and nothing else of the file: the mapping and the descriptor of the prologue are gone when the plan is
// returned, and the file is opened again here, for the duration of its scan.  (run_search_parallel plans every file before
// the first worker starts; a plan that kept its file open put a query over a thousand files past RLIMIT_NOFILE, where the
// reference — which opens inside the rayon task — has at most one file per thread open.)
Status execute_plan(FilePlan &plan, ResultCollector &rc) {
    if (!plan.status.ok() || !plan.needs_gpu) return plan.status;
    pcq_columns cols = plan.cols;
    cols.first_index = rc.next_index;
    const int fd = ::open(plan.path.c_str(), O_RDONLY);
    if (fd < 0) return Status::Err(PCQ_ERR_IO, plan.path + ": " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || (uint64_t)st.st_size < plan.file_size) {  // (truncated between the prologue and the scan)
        ::close(fd);
        return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
    }
    // The reference opens a file once, inside its task (last.rs:51); here the header was read by the prologue and the path
    // is opened again.  Another file under the same name — replaced, or rewritten in place — must not be scanned with the
    // first one's offsets, scale and point count.
    if ((uint64_t)st.st_dev != plan.file_dev || (uint64_t)st.st_ino != plan.file_ino || (uint64_t)st.st_size != plan.file_size ||
        (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec != plan.file_mtime_ns) {
        ::close(fd);
        return Status::Err(PCQ_ERR_IO, plan.path + ": the file changed while the query was running");
    }
    // (not waited for: the bytes have been read when this returns, the collector's accessors — or the driver's
    // synchronisation of the context before it merges — wait for the kernels; the next file's first read overlaps them)
    const int r = pcq_scan_fd_nowait(rc.context(), fd, &cols, &plan.pred, rc.handle());
    ::close(fd);
    rc.next_index += cols.n;
    return Status::FromLib(r);
}

namespace {
// plan helpers: `done` = resolved on the host (no GPU work), `gpu` = the scan is ready to be issued
FilePlan done(Status st = Status::Ok()) {
    FilePlan p;
    p.status = std::move(st);
    return p;
}
// `cols` points into the mapping of `file`: the plan keeps the offsets (a NULL column stays NULL; offset 0 never is a
// column, the header lives there)
FilePlan gpu(const std::string &path, const MappedFile &file, pcq_columns cols, const pcq_predicate &pred) {
    auto to_offset = [&](const void *p) -> const void * { return p ? (const void *)(uintptr_t)((const uint8_t *)p - file.data()) : nullptr; };
    cols.xyz = to_offset(cols.xyz);
    cols.cls = to_offset(cols.cls);
    cols.rgb = to_offset(cols.rgb);
    FilePlan plan;
    plan.needs_gpu = true;
    plan.path = path;
    plan.file_size = file.size();
    plan.file_dev = file.dev(), plan.file_ino = file.ino(), plan.file_mtime_ns = file.mtime_ns();
    plan.cols = cols;
    plan.pred = pred;
    return plan;
}
}  // namespace

// ---- last.rs:46-166 -------------------------------------------------------------------------------------
FilePlan plan_last_file_by_bounds_optimized(const std::string &path, const AABB &bounds) {
    auto holder = std::make_unique<MappedFile>();
    MappedFile &file = *holder;
    Status st = file.open(path);  // :51
    if (!st.ok()) return done(st);
    LasHeader h;
    st = parse_las_header(file.data(), file.size(), /*mask_format=*/false, &h);  // :53-54
    if (!st.ok()) return done(st);
    const uint8_t fmt = h.point_data_record_format;  // :68
    uint64_t cls_in_point;
    if (fmt <= 5) cls_in_point = 15;  // :69-79
    else if (fmt <= 10) cls_in_point = 16;
    else return done(invalid_format(fmt, path));
    const uint64_t n = h.number_of_points;
    const uint64_t otp = h.offset_to_point_data;
    const uint64_t cls_block = otp + n * cls_in_point;  // :80-81
    const auto col_in_point = las_offset_to_color(fmt);  // :83-88
    const std::optional<uint64_t> col_block = col_in_point ? std::optional<uint64_t>(otp + n * *col_in_point) : std::nullopt;  // :89-90

    if (!h.bounds.intersects(bounds)) return done();  // :92-94: resolved from the header alone

    pcq_predicate pred{};
    pred.kind = PCQ_PRED_BOUNDS;
    const int brc = pcq_box_to_local(bounds.min, bounds.max, h.scale, h.offset, pred.lmin, pred.lmax);  // :98-109
    if (brc) return done(Status::FromLib(brc));
    if (n == 0) return done();

    if (!block_ok(file, otp, n * 12) || !block_ok(file, cls_block, n) || (col_block && !block_ok(file, *col_block, n * 6)))
        return done(eof());
    pcq_columns cols{};
    cols.xyz = file.data() + otp;  // :114-121
    cols.xyz_stride = 12;
    cols.cls = file.data() + cls_block;  // :138-142
    cols.cls_stride = 1;
    cols.rgb = col_block ? file.data() + *col_block : nullptr;  // :145-153
    cols.rgb_stride = 6;
    cols.n = n;
    for (int a = 0; a < 3; a++) cols.scale[a] = h.scale[a], cols.offset[a] = h.offset[a];  // :156-160
    return gpu(path, file, cols, pred);
}
Status search_last_file_by_bounds_optimized(const std::string &path, const AABB &bounds, ResultCollector &rc) {
    FilePlan plan = plan_last_file_by_bounds_optimized(path, bounds);
    return execute_plan(plan, rc);
}

// ---- last.rs:213-293 -------------------------------------------------------------------------------------
FilePlan plan_last_file_by_classification_optimized(const std::string &path, uint8_t cls) {
    auto holder = std::make_unique<MappedFile>();
    MappedFile &file = *holder;
    Status st = file.open(path);  // :218
    if (!st.ok()) return done(st);
    LasHeader h;
    st = parse_las_header(file.data(), file.size(), /*mask_format=*/true, &h);  // :220-223
    if (!st.ok()) return done(st);
    const uint8_t fmt = h.point_data_record_format;  // :225
    uint64_t cls_in_point;
    if (fmt <= 5) cls_in_point = 15;  // :226-236
    else if (fmt <= 10) cls_in_point = 16;
    else return done(invalid_format(fmt, path));
    const auto col_in_point = las_offset_to_color(fmt);  // :238-243
    const uint64_t n = h.number_of_points;
    const uint64_t otp = h.offset_to_point_data;
    const uint64_t cls_block = cls_in_point * n + otp;  // :245-246, :254-256
    const std::optional<uint64_t> col_block = col_in_point ? std::optional<uint64_t>(otp + n * *col_in_point) : std::nullopt;  // :249-250
    if (n == 0) return done();

    if (!block_ok(file, cls_block, n) || !block_ok(file, otp, n * 12) || (col_block && !block_ok(file, *col_block, n * 6)))
        return done(eof());
    pcq_predicate pred{};
    pred.kind = PCQ_PRED_CLASS;
    pred.cls = cls;  // :259-262 whole byte
    pcq_columns cols{};
    cols.xyz = file.data() + otp;  // :265
    cols.xyz_stride = 12;
    cols.cls = file.data() + cls_block;
    cols.cls_stride = 1;
    cols.rgb = col_block ? file.data() + *col_block : nullptr;  // :272-280
    cols.rgb_stride = 6;
    cols.n = n;
    for (int a = 0; a < 3; a++) cols.scale[a] = h.scale[a], cols.offset[a] = h.offset[a];  // :283-287
    return gpu(path, file, cols, pred);
}
Status search_last_file_by_classification_optimized(const std::string &path, uint8_t cls, ResultCollector &rc) {
    FilePlan plan = plan_last_file_by_classification_optimized(path, cls);
    return execute_plan(plan, rc);
}

// ---- las.rs:52-148 ---------------------------------------------------------------------------------------
FilePlan plan_las_file_by_bounds_optimized(const std::string &path, const AABB &bounds) {
    auto holder = std::make_unique<MappedFile>();
    MappedFile &file = *holder;
    Status st = file.open(path);  // :57
    if (!st.ok()) return done(st);
    LasHeader h;
    st = parse_las_header(file.data(), file.size(), /*mask_format=*/false, &h);  // :59-60
    if (!st.ok()) return done(st);
    const int rec = h.point_data_record_length;  // :73 — printed before the early-out below
    auto with_rec = [rec](FilePlan p) {
        p.las_record_size = rec;
        return p;
    };
    const auto color_offset = las_offset_to_color(h.point_data_record_format);  // :74-80

    if (!h.bounds.intersects(bounds)) return with_rec(done());  // :82-84: resolved from the header alone

    pcq_predicate pred{};
    pred.kind = PCQ_PRED_BOUNDS;
    const int brc = pcq_box_to_local(bounds.min, bounds.max, h.scale, h.offset, pred.lmin, pred.lmax);  // :88-99
    if (brc) return with_rec(done(Status::FromLib(brc)));
    const uint64_t n = h.number_of_points, rl = h.point_data_record_length, otp = h.offset_to_point_data;
    if (n == 0) return with_rec(done());
    // every record up to its last needed byte: XYZ +0..12, class +15, colour +off..off+6
    const uint64_t last_needed = color_offset ? *color_offset + 6 : 16;
    if (!block_ok(file, otp, (n - 1) * rl + last_needed)) return with_rec(done(eof()));
    pcq_columns cols{};
    cols.xyz = file.data() + otp;  // :102-104
    cols.cls = file.data() + otp + 15;  // :121-124 — seek(Current(3)): always +15 on this path
    cols.rgb = color_offset ? file.data() + otp + *color_offset : nullptr;  // :127-135
    cols.xyz_stride = cols.cls_stride = cols.rgb_stride = rl;
    cols.n = n;
    for (int a = 0; a < 3; a++) cols.scale[a] = h.scale[a], cols.offset[a] = h.offset[a];  // :138-142
    FilePlan plan = gpu(path, file, cols, pred);
    plan.las_record_size = rec;
    return plan;
}
Status search_las_file_by_bounds_optimized(const std::string &path, const AABB &bounds, ResultCollector &rc, SearchLog *log) {
    FilePlan plan = plan_las_file_by_bounds_optimized(path, bounds);
    if (log) log->las_record_size = plan.las_record_size;
    return execute_plan(plan, rc);
}

// ---- las.rs:192-261 --------------------------------------------------------------------------------------
FilePlan plan_las_file_by_classification_optimized(const std::string &path, uint8_t cls) {
    auto holder = std::make_unique<MappedFile>();
    MappedFile &file = *holder;
    Status st = file.open(path);  // :197
    if (!st.ok()) return done(st);
    LasHeader h;
    st = parse_las_header(file.data(), file.size(), /*mask_format=*/false, &h);  // :199-200
    if (!st.ok()) return done(st);
    const uint8_t fmt = h.point_data_record_format;  // raw, unmasked (:202)
    uint64_t cls_in_point;
    if (fmt <= 5) cls_in_point = 15;
    else if (fmt <= 10) cls_in_point = 16;
    else return done(invalid_format(fmt, path));
    const auto color_offset = las_offset_to_color(fmt);  // :214-219
    const uint64_t n = h.number_of_points, rl = h.point_data_record_length, otp = h.offset_to_point_data;
    if (n == 0) return done();
    const uint64_t last_needed = color_offset ? *color_offset + 6 : cls_in_point + 1;
    if (!block_ok(file, otp, (n - 1) * rl + last_needed)) return done(eof());
    pcq_predicate pred{};
    pred.kind = PCQ_PRED_CLASS;
    pred.cls = cls;
    pcq_columns cols{};
    cols.xyz = file.data() + otp;  // :233-237
    cols.cls = file.data() + otp + cls_in_point;  // :224-228
    cols.rgb = color_offset ? file.data() + otp + *color_offset : nullptr;  // :240-248
    cols.xyz_stride = cols.cls_stride = cols.rgb_stride = rl;
    cols.n = n;
    for (int a = 0; a < 3; a++) cols.scale[a] = h.scale[a], cols.offset[a] = h.offset[a];  // :251-255
    return gpu(path, file, cols, pred);
}
Status search_las_file_by_classification_optimized(const std::string &path, uint8_t cls, ResultCollector &rc) {
    FilePlan plan = plan_las_file_by_classification_optimized(path, cls);
    return execute_plan(plan, rc);
}

// ---- searcher.rs ---------------------------------------------------------------------------------------------
namespace {
std::optional<std::string> extension_of(const std::string &path) {  // Path::extension().and_then(OsStr::to_str)
    const size_t slash = path.find_last_of('/');
    const std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = base.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return std::nullopt;
    return base.substr(dot + 1);
}
Status out_of_scope(const std::string &what, const std::string &path) {
    return Status::Err(PCQ_ERR_UNSUPPORTED, what + " is outside the MI355X hot path (SURVEY.md §2): " + path);
}
}  // namespace

// The host-only prologue of search_file for the formats whose prologue needs no GPU (LAS / LAST --optimized): everything
// the reference does before its per-point loop.  Other formats (and errors of the dispatch itself) are left to search_file.
std::optional<FilePlan> BoundsSearcher::plan_file(const std::string &path, SearchImplementation impl) const {
    const auto ext = extension_of(path);
    if (!ext || impl != SearchImplementation::Optimized) return std::nullopt;
    if (*ext == "las") return plan_las_file_by_bounds_optimized(path, bounds_);
    if (*ext == "last") return plan_last_file_by_bounds_optimized(path, bounds_);
    return std::nullopt;
}
std::optional<FilePlan> ClassSearcher::plan_file(const std::string &path, SearchImplementation impl) const {
    const auto ext = extension_of(path);
    if (!ext || impl != SearchImplementation::Optimized) return std::nullopt;
    if (*ext == "las") return plan_las_file_by_classification_optimized(path, class_);
    if (*ext == "last") return plan_last_file_by_classification_optimized(path, class_);
    return std::nullopt;
}

Status BoundsSearcher::search_file(const std::string &path, SearchImplementation impl, ResultCollector &collector,
                                   SearchLog *log) const {  // searcher.rs:43-90
    const auto ext = extension_of(path);
    if (!ext) return Status::Err(PCQ_ERR_EXTENSION, "Invalid extension on file " + path);
    if (*ext == "las" || *ext == "last") {
        if (impl == SearchImplementation::Regular) return out_of_scope("the Regular (non --optimized) search implementation", path);
        return *ext == "las" ? search_las_file_by_bounds_optimized(path, bounds_, collector, log)
                             : search_last_file_by_bounds_optimized(path, bounds_, collector);
    }
    if (*ext == "lazer") return search_lazer_file_by_bounds(path, bounds_, collector);  // searcher.rs:83, either implementation
    if (*ext == "laz") return out_of_scope("compressed format .laz", path);
    return Status::Err(PCQ_ERR_EXTENSION, "Unsupported file extension in file " + path);
}

Status ClassSearcher::search_file(const std::string &path, SearchImplementation impl, ResultCollector &collector,
                                  SearchLog *) const {  // searcher.rs:104-151
    const auto ext = extension_of(path);
    if (!ext) return Status::Err(PCQ_ERR_EXTENSION, "Invalid extension on file " + path);
    if (*ext == "las" || *ext == "last") {
        if (impl == SearchImplementation::Regular) return out_of_scope("the Regular (non --optimized) search implementation", path);
        return *ext == "las" ? search_las_file_by_classification_optimized(path, class_, collector)
                             : search_last_file_by_classification_optimized(path, class_, collector);
    }
    if (*ext == "lazer") return search_lazer_file_by_classification(path, class_, collector);  // searcher.rs:144
    if (*ext == "laz") return out_of_scope("compressed format .laz", path);
    return Status::Err(PCQ_ERR_EXTENSION, "Unsupported file extension in file " + path);
}

}  // namespace pcq
