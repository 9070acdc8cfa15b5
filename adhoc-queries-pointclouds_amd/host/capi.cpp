// capi.cpp — extern "C" view of the host layer (include/pcq_query.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "lz4_frame.hpp"
#include "pcq_host.hpp"
#include "pcq_query.h"

using namespace pcq;

struct pcq_host_collector {
    std::unique_ptr<ResultCollector> c;
};

static thread_local std::string g_msg;
static thread_local bool g_panic = false;

static int done(const Status &s) {
    if (!s.ok()) {
        g_msg = s.message;
        g_panic = s.panic;
    }
    return s.code;
}

extern "C" const char *pcq_query_last_error(void) { return g_msg.c_str(); }
extern "C" int pcq_query_last_was_panic(void) { return g_panic ? 1 : 0; }

extern "C" int pcq_query_parse_las_header(const uint8_t *data, size_t len, int mask_format, pcq_las_header_info *out) {
    if (!out) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    LasHeader h;
    Status st = parse_las_header(data, len, mask_format != 0, &h);
    memset(out, 0, sizeof *out);
    out->version_major = h.version_major;
    out->version_minor = h.version_minor;
    out->point_data_record_format = h.point_data_record_format;
    out->header_size = h.header_size;
    out->point_data_record_length = h.point_data_record_length;
    out->offset_to_point_data = h.offset_to_point_data;
    out->number_of_points = h.number_of_points;
    for (int a = 0; a < 3; a++) {
        out->scale[a] = h.scale[a];
        out->offset[a] = h.offset[a];
        out->min[a] = h.bounds.min[a];
        out->max[a] = h.bounds.max[a];
    }
    return done(st);
}

extern "C" int pcq_query_parse_aabb(const char *s, double bmin[3], double bmax[3]) {
    if (!s || !bmin || !bmax) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    AABB b;
    Status st = parse_aabb(s, &b);
    if (st.ok())
        for (int a = 0; a < 3; a++) bmin[a] = b.min[a], bmax[a] = b.max[a];
    return done(st);
}

extern "C" int pcq_query_is_valid_file(const char *path) { return path && is_valid_file(path) ? 1 : 0; }

extern "C" int pcq_query_get_total_bounds(const char *const *files, size_t nfiles, double bmin[3], double bmax[3]) {
    std::vector<std::string> v;
    for (size_t i = 0; i < nfiles; i++) v.emplace_back(files[i]);
    AABB b;
    Status st = get_total_bounds(v, &b);
    if (st.ok())
        for (int a = 0; a < 3; a++) bmin[a] = b.min[a], bmax[a] = b.max[a];
    return done(st);
}

extern "C" int pcq_query_lz4_frame_decode(const uint8_t *src, size_t n, uint64_t need, uint64_t unit, uint8_t *out, uint64_t cap) {
    if ((!src && n) || (!out && need) || cap < need) return done(Status::Err(PCQ_ERR_ARG, "bad argument"));
    // the same entry the LAZER reader uses: straight into the destination, spilling frames redone via a vector
    return done(lz4_frame_decode_into(src, n, (size_t)need, (size_t)unit, out));
}

template <typename F>
static int make_collector(int device, pcq_host_collector **out, F &&create) {
    if (!out) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    *out = nullptr;
    pcq_ctx *ctx = nullptr;
    Status st = thread_context(device, &ctx);
    if (!st.ok()) return done(st);
    auto hc = std::make_unique<pcq_host_collector>();
    st = create(ctx, nullptr, &hc->c);
    if (!st.ok()) return done(st);
    *out = hc.release();
    return PCQ_OK;
}

extern "C" int pcq_query_collector_new_count(int device, pcq_host_collector **out) {
    return make_collector(device, out, [](pcq_ctx *ctx, uint64_t *, std::unique_ptr<ResultCollector> *o) { return CountCollector::create(ctx, o); });
}
extern "C" int pcq_query_collector_new_buffer(int device, pcq_host_collector **out) {
    return make_collector(device, out, [](pcq_ctx *ctx, uint64_t *, std::unique_ptr<ResultCollector> *o) { return BufferCollector::create(ctx, o); });
}
extern "C" int pcq_query_collector_new_grid(int device, const double bmin[3], const double bmax[3], double cell_size,
                                            pcq_host_collector **out) {
    if (!bmin || !bmax) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    const AABB b = AABB::from_min_max_unchecked(bmin, bmax);
    return make_collector(device, out, [&](pcq_ctx *ctx, uint64_t *, std::unique_ptr<ResultCollector> *o) {
        return GridSampledCollector::create(ctx, b, cell_size, o);
    });
}
extern "C" int pcq_query_collector_free(pcq_host_collector *c) {
    delete c;
    return PCQ_OK;
}
extern "C" int pcq_query_collector_point_count(pcq_host_collector *c, uint64_t *out) {
    if (!c || !out) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    size_t n = 0;
    Status st = c->c->point_count(&n);
    *out = n;
    return done(st);
}
extern "C" int pcq_query_collector_has_points(pcq_host_collector *c) { return c && pcq_collector_has_points(c->c->handle()); }
extern "C" int pcq_query_collector_points(pcq_host_collector *c, pcq_point *out, uint64_t cap, uint64_t *out_n) {
    if (!c || !out_n) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    return done(Status::FromLib(pcq_collector_points(c->c->handle(), out, cap, out_n)));
}
extern "C" int pcq_query_collector_grid_cells(pcq_host_collector *c, uint64_t *out, uint64_t cap, uint64_t *out_n) {
    if (!c || !out_n) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    return done(Status::FromLib(pcq_collector_grid_cells(c->c->handle(), out, cap, out_n)));
}

extern "C" int pcq_query_search_file_bounds(const char *path, const double bmin[3], const double bmax[3], int optimized,
                                            pcq_host_collector *c, int *las_record_size) {
    if (!path || !bmin || !bmax || !c) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    if (las_record_size) *las_record_size = -1;
    AABB b;
    Status st = AABB::from_min_max(bmin, bmax, &b);
    if (!st.ok()) return done(st);
    SearchLog log;
    st = BoundsSearcher(b).search_file(path, optimized ? SearchImplementation::Optimized : SearchImplementation::Regular, *c->c, &log);
    if (las_record_size) *las_record_size = log.las_record_size;
    return done(st);
}
extern "C" int pcq_query_search_file_class(const char *path, uint8_t cls, int optimized, pcq_host_collector *c) {
    if (!path || !c) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    return done(ClassSearcher(cls).search_file(path, optimized ? SearchImplementation::Optimized : SearchImplementation::Regular, *c->c));
}

extern "C" int pcq_query_test_plan_replace_execute(const char *path, const char *replacement, const double bmin[3], const double bmax[3],
                                                   pcq_host_collector *c) {
    if (!path || !bmin || !bmax || !c) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    AABB b;
    Status st = AABB::from_min_max(bmin, bmax, &b);
    if (!st.ok()) return done(st);
    FilePlan plan = plan_last_file_by_bounds_optimized(path, b);
    if (replacement && ::rename(replacement, path) != 0) return done(Status::Err(PCQ_ERR_IO, std::string("rename: ") + strerror(errno)));
    return done(execute_plan(plan, *c->c));
}

extern "C" int pcq_query_main(int argc, const char *const *argv) {
    return query_main(
        argc, argv,
        [](const std::string &s) {
            fputs(s.c_str(), stdout);
            fputc('\n', stdout);
            fflush(stdout);
        },
        [](const std::string &s) {
            fputs(s.c_str(), stderr);
            fputc('\n', stderr);
        });
}

// Test entry: the CLI with the driver's test hooks (RunOptions::test_device_slots / test_allreduce_fail).  The `query`
// binary has no way to set them.  device_slots: "0,0"-style list or NULL; allreduce_fail: 0, 1 (early), 2 (late).
extern "C" int pcq_query_main_with_hooks(int argc, const char *const *argv, const char *device_slots, int allreduce_fail) {
    RunOptions hooks;
    for (const char *p = device_slots ? device_slots : ""; *p;) {
        char *end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p) break;
        hooks.test_device_slots.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
    }
    hooks.test_allreduce_fail = allreduce_fail;
    return query_main(
        argc, argv,
        [](const std::string &s) {
            fputs(s.c_str(), stdout);
            fputc('\n', stdout);
            fflush(stdout);
        },
        [](const std::string &s) {
            fputs(s.c_str(), stderr);
            fputc('\n', stderr);
        },
        &hooks);
}

// The file -> device-slot schedule of run_search_parallel (FileScheduler) under staggered context start-up, without a GPU:
// slot_of_file[i] = the slot that takes file i when slot k is ready at ready_ms[k] and a file costs ms_per_unit x cost.
extern "C" int pcq_query_simulate_schedule(const uint64_t *cost, size_t nfiles, const double *ready_ms, int nslots, double ms_per_unit,
                                           int *slot_of_file, int *home_slot, double *makespan_ms) {
    if ((!cost && nfiles) || !ready_ms || nslots < 1 || (!slot_of_file && nfiles)) return done(Status::Err(PCQ_ERR_ARG, "pcq_query_simulate_schedule: bad arguments"));
    const std::vector<uint64_t> c(cost, cost + nfiles);
    const std::vector<double> r(ready_ms, ready_ms + nslots);
    std::vector<int> slots;
    const double end = simulate_schedule(c, r, ms_per_unit, &slots);
    for (size_t i = 0; i < nfiles; i++) slot_of_file[i] = slots[i];
    if (home_slot) {
        FileScheduler sched(c, (size_t)nslots);
        for (size_t i = 0; i < nfiles; i++) home_slot[i] = (int)sched.home_slot(i);
    }
    if (makespan_ms) *makespan_ms = end;
    return 0;
}

// ---- resident dataset -------------------------------------------------------------------------------------
struct pcq_host_resident {
    std::unique_ptr<ResidentDataset> ds;
};
extern "C" int pcq_query_resident_load(int device, const char *const *files, size_t nfiles, pcq_host_resident **out) {
    if (!out || (!files && nfiles)) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    *out = nullptr;
    pcq_ctx *ctx = nullptr;
    Status st = thread_context(device, &ctx);
    if (!st.ok()) return done(st);
    std::vector<std::string> v;
    for (size_t i = 0; i < nfiles; i++) v.emplace_back(files[i]);
    auto h = std::make_unique<pcq_host_resident>();
    st = ResidentDataset::load(ctx, v, &h->ds);
    if (!st.ok()) return done(st);
    *out = h.release();
    return PCQ_OK;
}
extern "C" int pcq_query_resident_free(pcq_host_resident *r) {
    delete r;
    return PCQ_OK;
}
extern "C" int pcq_query_resident_count_bounds(pcq_host_resident *r, const double bmin[3], const double bmax[3], uint64_t *matches,
                                               uint64_t *points_scanned) {
    if (!r || !bmin || !bmax || !matches) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    AABB b;
    Status st = AABB::from_min_max(bmin, bmax, &b);
    if (!st.ok()) return done(st);
    return done(r->ds->count_bounds(b, matches, points_scanned));
}
extern "C" int pcq_query_resident_count_class(pcq_host_resident *r, uint8_t cls, uint64_t *matches, uint64_t *points_scanned) {
    if (!r || !matches) return done(Status::Err(PCQ_ERR_ARG, "null argument"));
    return done(r->ds->count_class(cls, matches, points_scanned));
}
