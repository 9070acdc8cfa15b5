// run_search.cpp — file enumeration, the sequential / parallel drivers, the point dumpers and the CLI
// (query/src/main.rs, query/src/dump_points.rs).
#include <dirent.h>
#include <sys/stat.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <limits>
#include <mutex>
#include <thread>

#include "pcq_host.hpp"

namespace pcq {

// ---- main.rs:29-57 ---------------------------------------------------------------------------------------
Status get_all_input_files(const std::string &input, std::vector<std::string> *out) {
    out->clear();
    struct stat st;
    if (stat(input.c_str(), &st) != 0) return Status::Err(PCQ_ERR_IO, "Input path " + input + " does not exist!");
    if (S_ISREG(st.st_mode)) {
        out->push_back(input);
        return Status::Ok();
    }
    if (S_ISDIR(st.st_mode)) {
        DIR *d = opendir(input.c_str());
        if (!d) return Status::Err(PCQ_ERR_IO, strerror(errno));
        while (struct dirent *e = readdir(d)) {  // unsorted read_dir order, not recursive (main.rs:42-47)
            if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
            std::string p = input;
            if (!p.empty() && p.back() != '/') p += '/';
            out->push_back(p + e->d_name);
        }
        closedir(d);
        return Status::Ok();
    }
    return Status::Err(PCQ_ERR_IO, "Input path " + input + " is neither file nor directory!");
}

// main.rs:185-189
bool is_valid_file(const std::string &path) {
    const size_t slash = path.find_last_of('/');
    const std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = base.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return false;
    const std::string ext = base.substr(dot + 1);
    return ext == "las" || ext == "laz" || ext == "last" || ext == "lazer";
}

// Rust's str::parse::<f64>: no leading/trailing whitespace, no hex; inf/nan/exponents accepted.
static bool parse_f64(const std::string &s, double *v) {
    if (s.empty() || s[0] == ' ' || s[0] == '\t' || s[0] == '\n') return false;
    size_t q = (s[0] == '+' || s[0] == '-') ? 1 : 0;
    if (s.size() > q + 1 && s[q] == '0' && (s[q + 1] == 'x' || s[q + 1] == 'X')) return false;
    char *end = nullptr;
    errno = 0;
    *v = strtod(s.c_str(), &end);
    return end != s.c_str() && *end == 0;
}

// main.rs:59-92
Status parse_aabb(const std::string &s, AABB *out) {
    std::vector<std::string> parts;
    size_t start = 0;
    for (;;) {
        const size_t semi = s.find(';', start);
        parts.push_back(s.substr(start, semi == std::string::npos ? std::string::npos : semi - start));
        if (semi == std::string::npos) break;
        start = semi + 1;
    }
    if (parts.size() != 6) return Status::Err(PCQ_ERR_ARG, "Could not parse AABB from string \"" + s + "\"");
    double c[6];
    for (int i = 0; i < 6; i++)
        if (!parse_f64(parts[i], &c[i]))
            return Status::Err(PCQ_ERR_ARG, "Could not parse AABB from string \"" + s + "\": invalid float literal");
    return AABB::from_min_max(c, c + 3, out);
}

// main.rs:94-120
Status get_total_bounds(const std::vector<std::string> &files, AABB *out) {
    const double mx = std::numeric_limits<double>::max();
    const double lo[3] = {mx, mx, mx}, hi[3] = {-mx, -mx, -mx};
    AABB total = AABB::from_min_max_unchecked(lo, hi);  // :114
    for (const auto &f : files) {
        if (f.size() >= 6 && f.compare(f.size() - 6, 6, ".lazer") == 0) {  // :102-107
            AABB b;
            Status st = lazer_file_bounds(f, &b);
            if (!st.ok()) return st;
            total = AABB::union_of(total, b);
            continue;
        }
        MappedFile mf;
        Status st = mf.open(f);
        if (!st.ok()) return st;
        const bool is_last = f.size() >= 5 && f.compare(f.size() - 5, 5, ".last") == 0;
        LasHeader h;
        st = parse_las_header(mf.data(), mf.size(), /*mask_format=*/is_last, &h);  // LASTReader::from masks (last_reader.rs:79)
        if (!st.ok()) return st;
        total = AABB::union_of(total, h.bounds);  // :115-117
    }
    *out = total;
    return Status::Ok();
}

// ---- dump_points.rs ------------------------------------------------------------------------------------------
Status FileDumper::create(const std::string &root_dir, std::unique_ptr<PointDumper> *out,
                          std::function<void(const std::string &)> print) {  // :45-60
    struct stat st;
    if (stat(root_dir.c_str(), &st) != 0) return Status::Err(PCQ_ERR_IO, "Path " + root_dir + " does not exist!");
    if (!S_ISDIR(st.st_mode)) return Status::Err(PCQ_ERR_IO, "Path " + root_dir + " is no directory!");
    auto d = std::make_unique<FileDumper>();
    d->root_ = root_dir;
    d->print_ = std::move(print);
    *out = std::move(d);
    return Status::Ok();
}

// LAS 1.2, point format 2 writer (dump_points.rs:63-116).  The reference delegates to las::Builder and
// pasture-io's LASWriter, neither of which is in the container, so byte-identical output is not
// defined (las stamps creation date / software id); parity is at the decoded-record level:
// header version 1.2 / format 2, offset = min position, scale by the rule at :80-88, records
// (X,Y,Z,class,RGB) with X = round((x - offset) / scale).
Status FileDumper::dump_points(const Point *points, size_t n) {
    if (n == 0) return Status::Ok();  // :65-67
    std::string path = root_;
    if (!path.empty() && path.back() != '/') path += '/';
    path += "matching_points_" + std::to_string(file_index_) + ".las";  // :68-70
    file_index_ += 1;
    const double big = std::numeric_limits<double>::max();
    double mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};  // :74-79
    for (size_t i = 0; i < n; i++) {
        const double p[3] = {points[i].x, points[i].y, points[i].z};
        for (int a = 0; a < 3; a++) {
            if (p[a] < mn[a]) mn[a] = p[a];
            if (p[a] > mx[a]) mx[a] = p[a];
        }
    }
    double max_extent = mx[0] - mn[0];  // :80-81
    for (int a = 1; a < 3; a++) max_extent = std::fmax(max_extent, mx[a] - mn[a]);
    const double min_scale = max_extent / (double)std::numeric_limits<int32_t>::max();  // :82
    double scale = std::pow(10.0, std::ceil(std::log10(min_scale)));                      // :84
    if (scale < 0.001) scale = 0.001;                                                     // :86-88
    if (!(scale >= 0.001)) scale = 0.001;  // NaN (single point: log10(0) = -inf -> 10^-inf = 0 -> clamped)

    if (print_) {  // :108 — through the same channel as the other stdout lines, so their order is the program order
        print_("Writing " + std::to_string(n) + " points");
    } else {
        printf("Writing %zu points\n", n);
        fflush(stdout);
    }
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return Status::Err(PCQ_ERR_IO, path + ": " + strerror(errno));
    uint8_t hdr[227];
    memset(hdr, 0, sizeof hdr);
    memcpy(hdr, "LASF", 4);
    hdr[24] = 1;
    hdr[25] = 2;
    memcpy(hdr + 26, "pcq", 3);
    memcpy(hdr + 58, "pcq query (MI355X)", 18);
    const time_t now = time(nullptr);
    struct tm tmv;
    gmtime_r(&now, &tmv);
    const uint16_t doy = (uint16_t)(tmv.tm_yday + 1), year = (uint16_t)(tmv.tm_year + 1900);
    memcpy(hdr + 90, &doy, 2);
    memcpy(hdr + 92, &year, 2);
    const uint16_t hs = 227, rl = 26;
    const uint32_t otp = 227, nvlr = 0, cnt = (uint32_t)n;
    memcpy(hdr + 94, &hs, 2);
    memcpy(hdr + 96, &otp, 4);
    memcpy(hdr + 100, &nvlr, 4);
    hdr[104] = 2;
    memcpy(hdr + 105, &rl, 2);
    memcpy(hdr + 107, &cnt, 4);
    memcpy(hdr + 111, &cnt, 4);  // points by return [0]
    for (int a = 0; a < 3; a++) {
        memcpy(hdr + 131 + 8 * a, &scale, 8);
        memcpy(hdr + 155 + 8 * a, &mn[a], 8);  // offset = min position (:92-104)
        memcpy(hdr + 179 + 16 * a, &mx[a], 8);
        memcpy(hdr + 187 + 16 * a, &mn[a], 8);
    }
    bool ok = fwrite(hdr, 1, sizeof hdr, f) == sizeof hdr;
    std::vector<uint8_t> buf;
    buf.reserve(26 * 65536);
    for (size_t i = 0; i < n && ok; i++) {
        uint8_t rec[26];
        memset(rec, 0, sizeof rec);
        const double p[3] = {points[i].x, points[i].y, points[i].z};
        for (int a = 0; a < 3; a++) {
            const int32_t v = (int32_t)std::llround((p[a] - mn[a]) / scale);
            memcpy(rec + 4 * a, &v, 4);
        }
        rec[15] = points[i].classification;
        memcpy(rec + 20, &points[i].r, 2);
        memcpy(rec + 22, &points[i].g, 2);
        memcpy(rec + 24, &points[i].b, 2);
        buf.insert(buf.end(), rec, rec + 26);
        if (buf.size() >= 26 * 65536) {
            ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
            buf.clear();
        }
    }
    if (ok && !buf.empty()) ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    if (fclose(f) != 0) ok = false;
    if (!ok) return Status::Err(PCQ_ERR_IO, path + ": write failed");
    dumped_ += n;  // :113
    return Status::Ok();
}

// The fallback of the count merge: the per-GPU counts, read one by one (main.rs:164-180 as a host loop).
static uint64_t merge_counts_on_host(const std::vector<pcq_ctx *> &ctxs, const std::vector<const uint64_t *> &counts, Status *status) {
    uint64_t total = 0;
    for (size_t k = 0; k < ctxs.size() && status->ok(); k++) {
        uint64_t part = 0;
        *status = Status::FromLib(pcq_copy_to_host(ctxs[k], &part, counts[k], 8));
        total += part;
    }
    return total;
}

// ---- drain shared by both drivers (main.rs:135-141 / :164-180) ---------------------------------------------
static Status drain(ResultCollector &c, PointDumper &dumper, std::optional<size_t> *num_matches) {
    if (c.has_points() && !dumper.wants_points()) {  // same observable behaviour, no device-to-host copy
        size_t n = 0;
        Status st = c.point_count(&n);
        if (!st.ok()) return st;
        return dumper.dump_points(nullptr, n);
    }
    if (const std::vector<Point> *ref = c.points_ref()) return dumper.dump_points(ref->data(), ref->size());
    if (auto pts = c.points()) return dumper.dump_points(pts->data(), pts->size());
    size_t n = 0;
    Status st = c.point_count(&n);
    if (!st.ok()) return st;
    *num_matches = num_matches->has_value() ? **num_matches + n : n;
    return Status::Ok();
}

// main.rs:122-144 — one collector across all files, in file order, on the first device.
Status run_search_sequential(const std::vector<std::string> &files, const Searcher &searcher, SearchImplementation impl,
                             const CollectorFactoryFn &factory, PointDumper &dumper, const RunOptions &opt, const PrintFn &print) {
    pcq_ctx *ctx = nullptr;
    Status st = thread_context(opt.devices.empty() ? 0 : opt.devices[0], &ctx);
    if (!st.ok()) return st;
    std::unique_ptr<ResultCollector> collector;
    st = factory(ctx, nullptr, &collector);  // :129
    if (!st.ok()) return st;
    for (const auto &f : files) {  // :131-133
        SearchLog log;
        const auto t_f = std::chrono::steady_clock::now();
        st = searcher.search_file(f, impl, *collector, &log);
        if (opt.stats)
            opt.stats->push_back({f, opt.devices.empty() ? 0 : opt.devices[0],
                                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_f).count()});
        if (log.las_record_size >= 0) print("Point record size: " + std::to_string(log.las_record_size));  // las.rs:73
        if (!st.ok()) return st;
    }
    std::optional<size_t> matches;
    st = drain(*collector, dumper, &matches);  // :135-141
    if (!st.ok()) return st;
    if (matches) print("Found " + std::to_string(*matches) + " matching points");
    return Status::Ok();
}

// ---- file -> device slot (pcq_host.hpp) -----------------------------------------------------------------
struct FileScheduler::Impl {
    std::mutex mu;
    std::vector<std::deque<size_t>> q;  // per slot: its files, largest first
    std::vector<uint64_t> left;         // per slot: cost not yet taken
    std::vector<uint64_t> cost;
};
FileScheduler::FileScheduler(const std::vector<uint64_t> &cost, size_t nslots) : impl_(std::make_shared<Impl>()), home_(cost.size(), 0) {
    if (nslots < 1) nslots = 1;
    impl_->q.resize(nslots);
    impl_->left.assign(nslots, 0);
    impl_->cost = cost;
    std::vector<size_t> order(cost.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });  // (equal files keep their input order)
    std::vector<uint64_t> load(nslots, 0);
    for (size_t i : order) {
        size_t best = 0;
        for (size_t k = 1; k < nslots; k++)
            if (load[k] < load[best]) best = k;
        load[best] += cost[i] ? cost[i] : 1;
        impl_->q[best].push_back(i);
        impl_->left[best] += cost[i];
        home_[i] = best;
    }
}
size_t FileScheduler::next(size_t slot) {
    Impl &m = *impl_;
    std::lock_guard<std::mutex> g(m.mu);
    if (slot >= m.q.size()) slot = 0;
    size_t from = slot;
    bool own = !m.q[slot].empty();
    if (!own) {  // the slot with the most work left gives up its smallest file
        from = m.q.size();
        for (size_t k = 0; k < m.q.size(); k++)
            if (!m.q[k].empty() && (from == m.q.size() || m.left[k] > m.left[from])) from = k;
        if (from == m.q.size()) return npos;
    }
    size_t f;
    if (own) f = m.q[from].front(), m.q[from].pop_front();
    else f = m.q[from].back(), m.q[from].pop_back();
    m.left[from] -= m.cost[f];
    return f;
}
double simulate_schedule(const std::vector<uint64_t> &cost, const std::vector<double> &ready_ms, double ms_per_unit, std::vector<int> *slot_of_file) {
    FileScheduler sched(cost, ready_ms.size());
    std::vector<double> t = ready_ms;
    std::vector<bool> done(ready_ms.size(), false);
    slot_of_file->assign(cost.size(), -1);
    double end = 0;
    for (;;) {
        size_t w = ready_ms.size();
        for (size_t k = 0; k < ready_ms.size(); k++)
            if (!done[k] && (w == ready_ms.size() || t[k] < t[w])) w = k;  // the worker that asks next
        if (w == ready_ms.size()) break;
        const size_t f = sched.next(w);
        if (f == FileScheduler::npos) {
            done[w] = true;
            continue;
        }
        (*slot_of_file)[f] = (int)w;
        t[w] += ms_per_unit * (double)cost[f];
        if (t[w] > end) end = t[w];
    }
    return end;
}

// main.rs:146-183 — files are independent units: one collector per file.  rayon's par_iter becomes
// host threads pulling file indices from a shared queue, `threads_per_device` per GPU; each thread
// owns a GPU context (stream + pinned staging), so host staging copies of one file overlap the
// kernels of another.  Results are merged in input-file order (rayon's collect preserves order).
//
// Host first: every rayon task of the reference starts with the same host-only prologue (open, header,
// block offsets, the header-AABB early-out of last.rs:92-94, the box conversion).  That prologue runs here
// for ALL files before any GPU context exists; a file it resolves (skipped, empty, in error) never reaches a
// worker, and a query it resolves completely never wakes the GPU (HIP start-up is 0.13-0.3 s,
// profiles/r01_hip_init_probe.log).
//
// Count queries: every GPU owns ONE device counter; all files of that GPU add to it (the kernels' finishing
// step is an atomic add), and the global count is the all-reduce of the per-GPU counters — main.rs:164-180
// as a single RCCL all-reduce(sum, u64) — read back once.  With one GPU the all-reduce is the identity.
Status run_search_parallel(const std::vector<std::string> &files, const Searcher &searcher, SearchImplementation impl,
                           const CollectorFactoryFn &factory, PointDumper &dumper, const RunOptions &opt, const PrintFn &print) {
    const size_t nfiles = files.size();
    const bool timing = getenv("PCQ_TIMING") != nullptr;
    const auto t_run = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run).count(); };
    std::vector<std::optional<FilePlan>> plans(nfiles);
    std::vector<Status> results(nfiles);
    std::vector<SearchLog> logs(nfiles);
    std::vector<size_t> work;  // files that need a GPU
    for (size_t i = 0; i < nfiles; i++) {
        plans[i] = searcher.plan_file(files[i], impl);
        if (plans[i]) {
            logs[i].las_record_size = plans[i]->las_record_size;
            results[i] = plans[i]->status;
            if (plans[i]->status.ok() && plans[i]->needs_gpu) work.push_back(i);
        } else {
            work.push_back(i);  // no host-only prologue for this format: search_file does everything
        }
    }
    if (timing) fprintf(stderr, "[pcq] %zu of %zu files need the GPU (host prologue of all files: %.1f ms)\n", work.size(), nfiles, since());

    std::vector<std::unique_ptr<ResultCollector>> collectors(nfiles);
    std::vector<int> file_device(nfiles, -1);
    std::vector<double> file_ms(nfiles, 0.0);
    std::vector<int> devices = opt.devices.empty() ? std::vector<int>{0} : opt.devices;
    if (!opt.test_device_slots.empty()) devices = opt.test_device_slots;  // (tests: device slots, repeats allowed — pcq_host.hpp)
    // file -> device slot: every slot starts with its own longest-processing-time share (by planned points: 12 B or 1 B a
    // point all the same, the files of one query have one predicate), so a GPU whose context is still coming up keeps its
    // files; who runs dry takes from the slot with the most left (FileScheduler)
    std::vector<uint64_t> work_cost(work.size(), 1);
    for (size_t w = 0; w < work.size(); w++)
        if (plans[work[w]]) work_cost[w] = plans[work[w]]->cols.n ? plans[work[w]]->cols.n : 1;
    FileScheduler sched(work_cost, devices.size());
    int tpd = opt.threads_per_device;
    if (tpd < 1) {  // not given: two where a second thread pays (pcq_host.hpp), one otherwise
        uint64_t planned_points = 0;
        for (size_t w : work)
            if (plans[w]) planned_points += plans[w]->cols.n;
        const size_t per_device = work.size() / devices.size();
        tpd = opt.collectors_fold_per_file && per_device >= 32 && planned_points / (work.size() ? work.size() : 1) < 8000000ull ? 2 : 1;
    }
    size_t nthreads = devices.size() * (size_t)tpd;
    if (nthreads > work.size()) nthreads = work.size();  // README.md:12
    const bool counting = !opt.collectors_yield_points;
    // Several GPUs, count query: how are the per-GPU counts merged?  main.rs:164-180 is a sum.  This process holds every context,
    // so the sum is N eight-byte reads and a host loop (< 1 ms).  The RCCL all-reduce of the same u64 (pcq_allreduce_sum_u64:
    // what a one-process-per-GPU integration calls, as bench.py does through its process group) costs a process that does
    // not carry RCCL yet 1.0-5.0 s to load librccl and 0.6 s for ncclCommInitAll — measured at ONE rank, tools/r03_rccl_cost.sh
    // — and the load stalls context creation and launches on every other thread meanwhile, on whatever thread it runs (a 6 ms file
    // took 1.0 s next to it).  The whole 16-file query scans for 0.12 s.  So: host sum, unless PCQ_MERGE=rccl asks for the
    // collective (then the communicator is built on a thread of this function while the files are scanned); and whenever the
    // collective fails, the exact per-GPU counts are summed on the host anyway.
    bool merge_rccl = false;
    std::thread comm_thread;
    struct JoinOnExit {
        std::thread &t;
        ~JoinOnExit() {
            if (t.joinable()) t.join();
        }
    } comm_joiner{comm_thread};
    if (counting && !work.empty()) {
        const char *policy = getenv("PCQ_MERGE");
        double planned_bytes = 0;
        for (size_t w : work)
            if (plans[w]) planned_bytes += (double)plans[w]->cols.n * (plans[w]->pred.kind == PCQ_PRED_CLASS ? 1.0 : 12.0);
        const double scan_seconds = planned_bytes / (40e9 * (double)devices.size());
        merge_rccl = opt.test_allreduce_fail != 0 || (policy && !strcmp(policy, "rccl"));
        // RCCL's NCCL_DEBUG output belongs on stderr (collective.hip); the environment is written HERE, before any other thread
        // of this query exists — the workers read the environment (PCQ_TIMING) while the communicator is being built
        if (merge_rccl) setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);
        if (timing)
            fprintf(stderr, "[pcq] count merge: %s (estimated scan time %.2f s on %zu GPU(s))\n", merge_rccl ? "RCCL all-reduce" : devices.size() > 1 ? "host sum of the per-GPU counts" : "one GPU, its counter is the total",
                    scan_seconds, devices.size());
        // PCQ_MERGE=rccl: the communicator is built on a thread of THIS function while the files are scanned, and joined before
        // the merge — and on every other way out of this function (comm_joiner): no thread outlives the query
        if (merge_rccl && devices.size() > 1) comm_thread = std::thread([devices] { (void)pcq_allreduce_prepare(devices.data(), (int)devices.size()); });
    }
    // per-GPU device counters (count queries): owned by the first worker of the device
    std::vector<uint64_t *> dev_counter(devices.size(), nullptr);
    std::vector<pcq_ctx *> dev_ctx(devices.size(), nullptr);
    std::vector<Status> dev_status(devices.size());
    // Collectors hold device memory owned by the worker's context; workers therefore stay alive
    // until the results have been drained.
    std::mutex mu;
    std::condition_variable cv;
    bool release = false;
    size_t finished = 0;
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nthreads; t++) {
        const size_t dslot = t % devices.size();
        const int device = devices[dslot];
        pool.emplace_back([&, device, dslot]() {
            pcq_ctx *ctx = nullptr;
            const auto t_a = std::chrono::steady_clock::now();
            Status cst = thread_context(device, &ctx);
            if (getenv("PCQ_TIMING"))
                fprintf(stderr, "[pcq] context on device %d ready after %.1f ms\n", device,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_a).count());
            uint64_t *counter = nullptr;
            if (cst.ok() && counting) {
                std::lock_guard<std::mutex> g(mu);
                if (!dev_counter[dslot]) {
                    void *p = nullptr;
                    Status st = Status::FromLib(pcq_device_alloc(ctx, 16, &p));
                    if (st.ok()) st = Status::FromLib(pcq_device_memset(ctx, p, 0, 16, nullptr));
                    if (st.ok()) st = Status::FromLib(pcq_ctx_synchronize(ctx));
                    if (st.ok()) dev_counter[dslot] = (uint64_t *)p, dev_ctx[dslot] = ctx;
                    else cst = st;
                }
                counter = dev_counter[dslot];
            }
            for (;;) {
                const size_t w = sched.next(dslot);
                if (w == FileScheduler::npos) break;
                const size_t i = work[w];
                if (!cst.ok()) {
                    results[i] = cst;
                    continue;
                }
                file_device[i] = device;
                const auto t_f = std::chrono::steady_clock::now();
                Status st = factory(ctx, counter, &collectors[i]);  // :156
                if (st.ok()) st = plans[i] ? execute_plan(*plans[i], *collectors[i]) : searcher.search_file(files[i], impl, *collectors[i], &logs[i]);  // :158
                plans[i].reset();
                if (st.ok()) st = collectors[i]->file_done();  // a grid keeps its winners, not a tuple per scanned point
                results[i] = st;
                file_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_f).count();
                if (getenv("PCQ_TIMING")) fprintf(stderr, "[pcq] file %zu searched in %.1f ms\n", i, file_ms[i]);
            }
            if (cst.ok()) {  // everything this worker enqueued has run before the counters are merged
                const Status sst = Status::FromLib(pcq_ctx_synchronize(ctx));
                std::lock_guard<std::mutex> g(mu);
                if (!sst.ok() && dev_status[dslot].ok()) dev_status[dslot] = sst;
            }
            std::unique_lock<std::mutex> lk(mu);
            finished++;
            cv.notify_all();
            cv.wait(lk, [&] { return release; });
            // collectors created on this thread's context must be destroyed before the context
            for (size_t i = 0; i < nfiles; i++)
                if (collectors[i] && collectors[i]->context() == ctx) collectors[i].reset();
            if (counter && dev_ctx[dslot] == ctx) pcq_device_free(ctx, counter);
            lk.unlock();
            release_thread_contexts();  // here, not in the thread-exit phase (core.cpp)
        });
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return finished == nthreads; });
    }
    if (timing) fprintf(stderr, "[pcq] all workers done %.1f ms after the search started\n", since());
    // The communicator's thread redirects descriptor 1 while RCCL prints its banner (collective.hip): it has ended before
    // this function prints its first line behind the scans.
    if (comm_thread.joinable()) comm_thread.join();
    Status final_status = Status::Ok();
    for (size_t i = 0; i < nfiles; i++)
        if (logs[i].las_record_size >= 0) print("Point record size: " + std::to_string(logs[i].las_record_size));
    for (size_t i = 0; i < nfiles && final_status.ok(); i++)
        if (!results[i].ok()) final_status = results[i];  // :161-163 first Err aborts
    for (size_t d = 0; d < devices.size() && final_status.ok(); d++)
        if (!dev_status[d].ok()) final_status = dev_status[d];
    std::optional<size_t> matches;
    if (final_status.ok() && counting) {
        // :164-180 — the sum over the collectors is the sum over the per-GPU counters: one all-reduce, OUT OF PLACE (word 0
        // of a GPU's counter block is its count, word 1 receives the sum): whatever happens to the collective — and
        // whenever: before it ran, or on rank 3's stream after all ranks had reduced — the per-GPU counts are still
        // what they were, so the fallback below cannot read a partially reduced value.
        uint64_t total = 0;
        std::vector<pcq_ctx *> ctxs;
        std::vector<const uint64_t *> sends;
        std::vector<uint64_t *> recvs;
        for (size_t d = 0; d < devices.size(); d++)
            if (dev_counter[d]) ctxs.push_back(dev_ctx[d]), sends.push_back(dev_counter[d]), recvs.push_back(dev_counter[d] + 1);
        if (comm_thread.joinable()) comm_thread.join();
        if (!ctxs.empty() && !merge_rccl && ctxs.size() > 1) {
            total = merge_counts_on_host(ctxs, sends, &final_status);  // (a short query: RCCL would take longer to load than the scans took)
        } else if (!ctxs.empty()) {
            if (opt.test_allreduce_fail) {  // tests: make the collective fail (1 early, 2 late, 3 inside the group), through the real RCCL calls
                (void)pcq_set_option(ctxs[0], "allreduce_single_rank", 1);
                (void)pcq_set_option(ctxs[0], "allreduce_fail", opt.test_allreduce_fail);
            }
            Status ast = Status::FromLib(pcq_allreduce_sum_u64(ctxs.data(), sends.data(), recvs.data(), (int)ctxs.size()));
            if (ast.ok()) {
                final_status = Status::FromLib(pcq_copy_to_host(ctxs[0], &total, recvs[0], 8));
            } else {
                // RCCL missing or unusable must not turn a correct answer into an error: the per-GPU counts are exact, so
                // they are read one by one and summed here
                fprintf(stderr, "warning: all-reduce of the per-GPU counts failed (%s); summing on the host\n", ast.message.c_str());
                total = merge_counts_on_host(ctxs, sends, &final_status);
            }
        }
        if (nfiles) matches = (size_t)total;  // no collector at all: num_matches stays None (main.rs:164)
    } else if (final_status.ok()) {
        for (size_t i = 0; i < nfiles && final_status.ok(); i++) {  // :165-176, input-file order
            if (!collectors[i]) continue;  // resolved on the host: an empty collector, nothing to dump
            std::optional<size_t> none;
            final_status = drain(*collectors[i], dumper, &none);
        }
    }
    if (timing) fprintf(stderr, "[pcq] results merged / drained %.1f ms after the search started\n", since());
    {
        std::unique_lock<std::mutex> lk(mu);
        release = true;
        cv.notify_all();
    }
    for (auto &th : pool) th.join();
    if (timing) fprintf(stderr, "[pcq] contexts released %.1f ms after the search started\n", since());
    if (opt.stats)
        for (size_t i = 0; i < nfiles; i++) opt.stats->push_back({files[i], file_device[i], file_ms[i]});
    if (!final_status.ok()) return final_status;
    if (matches) print("Found " + std::to_string(*matches) + " matching points");  // :178-180
    return Status::Ok();
}

// ---- main.rs:191-319 ---------------------------------------------------------------------------------------------
int query_main(int argc, const char *const *argv, const PrintFn &out, const PrintFn &err, const RunOptions *test_hooks) {
    const auto t_start = std::chrono::steady_clock::now();  // :192
    std::optional<std::string> input, bounds_s, class_s, output, density_s, stats_json;
    std::vector<FileStat> file_stats;
    bool parallel = false, optimized = false;
    RunOptions opt;
    opt.devices = {0};
    opt.collectors_yield_points = true;
    if (test_hooks) opt.test_device_slots = test_hooks->test_device_slots, opt.test_allreduce_fail = test_hooks->test_allreduce_fail;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        std::optional<std::string> *dst = nullptr;
        std::optional<std::string> ext_val;
        if (a == "-i" || a == "--input") dst = &input;
        else if (a == "--bounds") dst = &bounds_s;
        else if (a == "--class") dst = &class_s;
        else if (a == "-o" || a == "--output") dst = &output;
        else if (a == "--density") dst = &density_s;
        else if (a == "--parallel") { parallel = true; continue; }
        else if (a == "--optimized") { optimized = true; continue; }
        else if (a == "--stats-json") dst = &stats_json;  // extra flag: machine-readable timing sidecar
        else if (a == "--gpus" || a == "--device" || a == "--threads-per-gpu") dst = &ext_val;  // extra flags (not in the reference)
        else if (a == "-h" || a == "--help") {
            out("I/O experiments 0.1\nLAS I/O experiments (MI355X-native predicate path)\n\nUSAGE:\n    query [FLAGS] [OPTIONS] --input <FILE>\n\n"
                "FLAGS:\n        --optimized    Run search with optimized implementation\n        --parallel     Run search in parallel\n\n"
                "OPTIONS:\n        --bounds <BOUNDS>    \"minX;minY;minZ;maxX;maxY;maxZ\"\n        --class <CLASS>      object class (u8)\n"
                "        --density <DENSITY>  maximum density (grid cell size)\n    -i, --input <FILE>       file or directory\n"
                "    -o, --output <OUTPUT>    output directory\n        --gpus <N>           (extra) number of GPUs to shard files over\n"
                "        --device <D>         (extra) first GPU to use\n        --threads-per-gpu <T> (extra) host threads feeding each GPU (default: 1; 2 for many small files with --density)\n"
                "        --stats-json <PATH>  (extra) write per-file timings as JSON");
            return 0;
        } else {
            err("error: Found argument '" + a + "' which wasn't expected, or isn't valid in this context");
            return 1;
        }
        if (i + 1 >= argc) {
            err("error: The argument '" + a + "' requires a value but none was supplied");
            return 1;
        }
        *dst = std::string(argv[++i]);
        if (dst == &ext_val) {
            char *end = nullptr;
            const long v = strtol(ext_val->c_str(), &end, 10);
            if (end == ext_val->c_str() || *end || v < 0 || v > 64) {
                err("error: Invalid value for '" + a + "'");
                return 1;
            }
            if (a == "--gpus") {
                const int first = opt.devices.empty() ? 0 : opt.devices[0];
                opt.devices.clear();
                for (int d = 0; d < (v < 1 ? 1 : (int)v); d++) opt.devices.push_back(first + d);
            } else if (a == "--device") {
                const size_t cnt = opt.devices.size();
                opt.devices.clear();
                for (size_t d = 0; d < cnt; d++) opt.devices.push_back((int)v + (int)d);
            } else {
                opt.threads_per_device = v < 1 ? 1 : (int)v;
            }
        }
    }
    if (!input) {
        err("error: The following required arguments were not provided:\n    --input <FILE>");
        return 1;
    }

    std::vector<std::string> all_files, input_files;
    Status st = get_all_input_files(*input, &all_files);  // :222-223
    if (!st.ok()) {
        err("Error: " + st.message);
        return 1;
    }
    for (const auto &f : all_files)
        if (is_valid_file(f)) input_files.push_back(f);

    uint64_t total_file_size = 0;  // :227-231
    for (const auto &f : input_files) {
        struct stat sb;
        if (stat(f.c_str(), &sb) == 0) total_file_size += (uint64_t)sb.st_size;
    }
    const double total_file_size_mib = (double)total_file_size / 1048576.0;

    std::optional<AABB> maybe_bounds;
    if (bounds_s) {  // :235 — expect(): panic
        AABB b;
        st = parse_aabb(*bounds_s, &b);
        if (!st.ok()) {
            err(st.panic ? st.message : "Could not prase argument BOUNDS: " + st.message);
            return 101;
        }
        maybe_bounds = b;
    }
    std::optional<uint8_t> maybe_class;
    if (class_s) {  // :236
        const std::string &s = *class_s;
        bool good = !s.empty() && s.size() <= 4;
        unsigned v = 0;
        size_t k = (good && s[0] == '+') ? 1 : 0;
        good = good && k < s.size();
        for (; good && k < s.size(); k++) {
            if (s[k] < '0' || s[k] > '9') good = false;
            else v = v * 10 + (unsigned)(s[k] - '0');
        }
        if (!good || v > 255) {
            err("Could not prase argument CLASS");
            return 101;
        }
        maybe_class = (uint8_t)v;
    }
    std::optional<double> maybe_density;
    if (density_s) {  // :237
        double d;
        if (!parse_f64(*density_s, &d)) {
            err("Could not prase argument DENSITY");
            return 101;
        }
        maybe_density = d;
    }
    if (maybe_bounds && maybe_class) {  // :238-240
        err("Error: Specifying BOUNDS and CLASS at the same time is invalid! Specify either BOUNDS or CLASS argument!");
        return 1;
    }
    if (!maybe_bounds && !maybe_class) {  // :242-244
        err("Error: Found neither BOUNDS nor CLASS argument but exactly one of these arguments is required!");
        return 1;
    }

    std::unique_ptr<Searcher> searcher;  // :246-251
    if (maybe_bounds) searcher = std::make_unique<BoundsSearcher>(*maybe_bounds);
    else searcher = std::make_unique<ClassSearcher>(*maybe_class);

    CollectorFactoryFn factory;  // :253-273
    if (maybe_density) {
        AABB gb;
        if (maybe_bounds) gb = *maybe_bounds;
        else {
            st = get_total_bounds(input_files, &gb);
            if (!st.ok()) {
                err("Error: " + st.message);
                return 1;
            }
        }
        const double cell = *maybe_density;
        factory = [gb, cell](pcq_ctx *ctx, uint64_t *, std::unique_ptr<ResultCollector> *o) { return GridSampledCollector::create(ctx, gb, cell, o); };
        opt.collectors_fold_per_file = true;
    } else if (output) {
        factory = [](pcq_ctx *ctx, uint64_t *, std::unique_ptr<ResultCollector> *o) { return BufferCollector::create(ctx, o); };
    } else {
        factory = [](pcq_ctx *ctx, uint64_t *shared, std::unique_ptr<ResultCollector> *o) { return CountCollector::create(ctx, o, shared); };
        opt.collectors_yield_points = false;
    }

    std::unique_ptr<PointDumper> dumper;  // :275-281
    if (output) {
        st = FileDumper::create(*output, &dumper, out);
        if (!st.ok()) {
            err("Error: " + st.message);
            return 1;
        }
    } else {
        dumper = std::make_unique<IgnoreDumper>();
    }
    const SearchImplementation impl = optimized ? SearchImplementation::Optimized : SearchImplementation::Regular;  // :283-287

    if (stats_json) opt.stats = &file_stats;
    out("Searching " + std::to_string(input_files.size()) + " files...");  // :289

    st = parallel ? run_search_parallel(input_files, *searcher, impl, factory, *dumper, opt, out)
                  : run_search_sequential(input_files, *searcher, impl, factory, *dumper, opt, out);
    if (!st.ok()) {
        if (st.panic) {
            err(st.message);
            return 101;
        }
        err("Error: " + st.message);
        return 1;
    }

    if (getenv("PCQ_TIMING"))
        fprintf(stderr, "[pcq] total in-process %.1f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();  // :309
    const double throughput_mibs = ((double)total_file_size / elapsed) / 1048576.0;
    char line[256];
    snprintf(line, sizeof line, "Searched %.2f MiB in %.2fs (throughput: %.2fMiB/s)", total_file_size_mib, elapsed, throughput_mibs);  // :313-316
    out(line);
    if (stats_json) {  // sidecar: never changes stdout
        if (FILE *f = fopen(stats_json->c_str(), "w")) {
            fprintf(f, "{\"files\": %zu, \"bytes\": %llu, \"seconds\": %.6f, \"parallel\": %s, \"gpus\": %zu, \"per_file\": [", input_files.size(),
                    (unsigned long long)total_file_size, elapsed, parallel ? "true" : "false", opt.devices.size());
            for (size_t i = 0; i < file_stats.size(); i++) {
                std::string esc;
                for (char ch : file_stats[i].path) {
                    if (ch == '"' || ch == '\\') esc += '\\';
                    esc += ch;
                }
                fprintf(f, "%s{\"path\": \"%s\", \"device\": %d, \"search_ms\": %.3f}", i ? ", " : "", esc.c_str(), file_stats[i].device,
                        file_stats[i].search_ms);
            }
            fprintf(f, "]}\n");
            fclose(f);
        }
    }
    return 0;
}

}  // namespace pcq
