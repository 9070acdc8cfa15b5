// pcq_internal.h — shared declarations of libpcq.so (not part of the public ABI; see include/pcq.h).
#pragma once

#include <sched.h>
#include <atomic>
#include <thread>
#include "copy_pool.h"
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "pcq.h"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
int pcq_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define PCQ_HIP(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return pcq_fail(PCQ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                                \
    } while (0)

// Every entry point that takes a context or a collector runs on the context's device, whatever device the calling
// thread used before (a driver thread draining the collectors of several GPUs sits on device 0 by default: memory
// allocated and kernels launched from there would land on the wrong GPU).  The caller's device is restored on return.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int device) {
        if (device < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) changed = hipSetDevice(device) == hipSuccess && prev >= 0;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define PCQ_ON_DEVICE_OF_CTX(ctx) DeviceGuard _device_guard((ctx) ? (ctx)->device : -1)
#define PCQ_ON_DEVICE_OF_COLLECTOR(c) DeviceGuard _device_guard((c) && (c)->ctx ? (c)->ctx->device : -1)

// ---------------------------------------------------------------------------------------------
// device-side views
// ---------------------------------------------------------------------------------------------

// Predicate in device form.  The i64 box of the reference (last.rs:98-109) is clamped to the i32
// value range of the stored coordinates; `(uint32)(v - lo) <= width` is then exactly
// `lo <= v && v <= hi` on sign-extended values (last.rs:122-135).
struct DevPred {
    int32_t kind;      // pcq_predicate_kind
    int32_t empty;     // 1: no i32 coordinate can match (box entirely outside the i32 range)
    int32_t lo[3];
    uint32_t width[3];
    uint32_t cls;
    uint32_t _pad;
    double wmin[3], wmax[3];  // PCQ_PRED_BOUNDS_F64
};

struct DevCols {
    const uint8_t *xyz;
    const uint8_t *cls;
    const uint8_t *rgb;
    uint64_t xyz_stride, cls_stride, rgb_stride;
    uint64_t n;
    uint64_t first_index;
    double scale[3];
    double offset[3];
};

// One segment of a batched count launch.
struct DevSegment {
    const int4 *xyz;       // 16-byte aligned positions block
    uint64_t n;            // points
    uint64_t tile_begin;   // first global wave-tile index of this segment
    int32_t lo[3];
    uint32_t width[3];
    int32_t empty;
    int32_t _pad;
};

// One segment of a batched class-count launch: a LAST classification block of any alignment.
struct DevClassSegment {
    const uint8_t *cls;
    uint64_t n;            // bytes (= points)
    uint64_t head;         // bytes in front of the first 16-byte aligned byte
    uint64_t nvec;         // 16-byte vectors in the aligned body
    uint64_t tile_begin;   // first global wave-tile (256 vectors = 4 KiB) of this segment
    uint32_t pat;          // class byte replicated x4
    uint32_t _pad;
};

// SparseGrid parameters (grid_sampling.rs:9-47) in device form.
struct DevGrid {
    double bmin[3], bmax[3];
    double cell_size;
    double dims_f[3];     // dims as f64 (`self.dimensions.x as f64`, grid_sampling.rs:51)
    double inv_extent[3]; // 1 / (bmax - bmin): only to find the cell WITHOUT the division when that is provably safe (grid_common.h cell_of)
    double qk[3];         // RN(dims / (bmax - bmin)), and the range / boundary guard of the short cell computation (grid_common.h cell_fast)
    double qmax[3], guard[3];
    uint64_t mask[3];     // (1 << bits) - 1
    uint32_t shift[3];    // 0, bits_x, bits_x + bits_y  (already & 63)
    uint32_t keys_wide;   // 1: a key can have more than 32 bits (grid_common.h cell_hash)
};

constexpr uint64_t PCQ_EMPTY_KEY = ~0ull;
constexpr uint64_t PCQ_NO_INDEX = ~0ull;

// ---------------------------------------------------------------------------------------------
// host-side objects
// ---------------------------------------------------------------------------------------------
struct GridState;
struct PoolBlock {
    void *p = nullptr;
    size_t bytes = 0;
    bool used = false;
};
struct pcq_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // compute stream
    hipStream_t copy_stream = nullptr;  // H2D stream
    hipStream_t scratch_stream = nullptr;  // the stream whose kernels may still be using the context's scratch (pcq_scratch_stream)
    hipEvent_t copied[2] = {nullptr, nullptr};
    hipEvent_t consumed[2] = {nullptr, nullptr};
    int num_cus = 0;
    hipDeviceProp_t prop;
    // scratch: per-block partial counts / block offsets
    uint64_t *d_partials = nullptr;
    size_t partials_cap = 0;
    uint64_t *d_scalars = nullptr;      // a few device u64 scratch words
    uint64_t *h_scalars = nullptr;      // pinned mirror
    // staging for pcq_scan_host
    uint8_t *h_stage[2] = {nullptr, nullptr};
    uint8_t *d_stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    std::thread stage_warm;             // pcq_prepare_host_scans: joined by whoever touches the staging ring or the copy pool next
    bool stage_busy[2] = {false, false};  // kernels not yet known to be done with staging pair b (event consumed[b])
    // segment table for batched launches
    DevSegment *d_segments = nullptr;
    DevSegment *h_segments = nullptr;
    size_t segments_cap = 0;
    size_t segments_uploaded = 0;       // number of segments of the table currently in d_segments (0 = none)
    int segments_kind = -1;             // predicate kind of the uploaded table
    // device-memory pool (pcq_pool_alloc / pcq_pool_free): the grid collector's tuple runs, partition buffers and
    // winner arrays are gigabytes per file, and a device allocation of that size costs from tens of milliseconds to
    // over a second (profiles/r01_grid_timeline.txt) — per-file grids (main.rs:156) reuse the blocks of the file before
    std::vector<PoolBlock> pool;
    uint64_t pool_limit = 96ull << 30;  // free bytes the pool may keep
    // diagnostics of the grid collector (pcq_get_option): folds run, folds that needed a second partition level,
    // folds repeated because a partition overflowed its LDS table, the last fold's second-level fan-out
    int64_t grid_folds = 0, grid_level2 = 0, grid_refolds = 0, grid_last_f2 = 0;
    int64_t grid_compactions = 0;       // folds whose bins were copied together first (short fragments)
    int64_t grid_level2_exact = 0;      // second levels repeated in the exact (counting) form: a sub-partition had outgrown its region
    int64_t grid_pending_budget = 0;    // option: tuples a grid collector may hold before it folds (0 = default)
    int allreduce_single_rank = 0;      // option: pcq_allreduce_sum_u64 with ONE rank still goes through RCCL (communicator of one
                                        // device, ncclAllReduce) — exercises the run-time binding on a single-GPU box
    int allreduce_fail = 0;             // option (tests): pcq_allreduce_sum_u64 fails — 1: before anything is touched, 2: after the reduction has run, 3: inside the group
    int grid_f2 = 0;                    // option (tests): second-level fan-out a fold starts from (0 = from the measured estimate)
    int grid_agg = 0;                   // option: pass 0 folds a tile's duplicate cells before they travel — 0 = while it pays (per workgroup),
                                        // 1 = every tile, 2 = never; the results are the same, the tuples moved are not
    int grid_stream = 1;                // option (tests): 0 = a coarse grid's bins are folded by k_fold<BIG> (the fallback of the streaming fold) only
    int64_t grid_deferred = 0;          // diagnostics: bins the streaming fold left to k_fold<BIG> (survivor list outgrown)
    int host_in_place = 2;              // option: count and grid scans of host / file data read the pinned staging ring IN PLACE (over PCIe) instead of
                                        // copying it to a device twin first: 0 never, 1 always, 2 while the process's copy path is being set up
                                        // (pcq_api.hip scan_host_impl)
    std::thread copy_warm;              // the thread that sets it up: one pinned megabyte through hipMemcpyAsync, beside the first file's scan
    std::atomic<int> copy_warm_state{0};  // 0 not started, 1 under way, 2 done
    void *copy_warm_h = nullptr, *copy_warm_d = nullptr;
    int emit_park_max = 256;            // option: a tile with at most this many matches leaves them as 16-byte words for the emit (0 = never; <= 256)
    int emit_sparse_max = 64;           // option: a tile of 2048 points with at most this many matches is written by k_emit_sparse (0 = never)
    bool scanned_before = false;        // (PCQ_TIMING: the first host / file scan of a context prints where its time goes)
    int grid_block_pad = 0;             // option: 16-byte units between the end of a tile's block of tuples and the next block
    int grid_tuple16 = 1;               // option (tests): 0 = every scan writes 24-byte tuples (the form a 16-byte tuple falls back to), 2 = 16-byte tuples without the second level's selector
    int64_t grid_last_tuples = 0;       // diagnostics: tuples the last fold found pending (after pass 0's own fold)
    // options
    int grid_blocks_per_cu = 2;   // persistent blocks per CU of the generic (strided) count kernels and the chunk index
#ifdef PCQ_LAB                    // libpcq_lab.so only: the kernel shapes of csrc/lab/scan_count_lab.hip
    int k1_variant = 12;          // per-file K1: 12 = one wave per workgroup, two adjacent 3 KiB tiles per step, software-pipelined (= the product's)
    int k1_waves_per_cu = 3;
    int batch_blocks_per_cu = 3;
    int k1_grid = 0;              // absolute number of workgroups for the one-wave per-file kernels (0 = num_cus x k1_waves_per_cu)
    int batch_variant = 3;        // batched K1: 0 = 256-thread blocks · 1 / 2 = one wave per workgroup, 2 / 3 tiles per step · 3 = pipelined (= the product's)
    int batch_waves_per_cu = 3;
    int class_batch_loads = 4;
    int class_batch_waves_per_cu = 4;
    int class_batch_pipe = 1;
#endif
    int numa_node = -1;               // NUMA node the GPU hangs off (sysfs), -1 if unknown
    cpu_set_t node_cpus;              // its CPUs (empty if unknown)
    int numa_local = 1;               // option "numa_local": staging buffers and copy helpers on that node
    int copy_threads = 16;        // threads filling a staging buffer (caller + helpers): 2-4 reach the PCIe rate from memory next to the
                                  // GPU, page-cache pages on the other socket need 8 (profiles/r01_cli_probe_timing.log); a stream of
                                  // files read with pread: 8 -> 5.5, 12 -> 5.0, 16 -> 4.9 ms per 240 MB (profiles/r04_cli_threads.log).
                                  // pcq_init caps it at the host's hardware threads per GPU.
    CopyPool *copy_pool = nullptr;  // created on first use by pcq_scan_host / pcq_scan_fd
    uint64_t chunk_points = 1ull << 20;    // 12 MB of positions per staging chunk: the steady rate of 24 MB (profiles/r01_host_path_rate.json: 1-8 Mi equal)
                                           // at half the pinning in front of a process's first file (profiles/r04_cli_chunks.log: 25 -> 20 ms)
};

// pcq_api.hip: [offset, offset+bytes) of fd -> device memory through the pinned staging buffers
int pcq_stream_fd_to_device(pcq_ctx *ctx, int fd, uint64_t offset, uint64_t bytes, uint8_t *d_dst);

enum { COLL_COUNT = 0, COLL_BUFFER = 1, COLL_GRID = 2 };

struct pcq_collector {
    int kind = COLL_COUNT;
    pcq_ctx *ctx = nullptr;
    // count
    uint64_t *d_count = nullptr;
    bool owns_count = false;
    // buffer: packed 31-byte points in HBM.  The number of points lives on the device (d_count): every scan reads it as
    // its base and moves it on, so scans need no round trip.  The host only knows an upper bound (one match per
    // scanned point) and asks the device for the truth when that bound outgrows the buffer.
    uint8_t *d_points = nullptr;
    uint64_t n_upper = 0, cap_points = 0;
    int count_slot = 0;                 // which of the two words of d_count holds the current count (a scan reads one, writes the other)
    // grid
    double bmin[3], bmax[3], cell_size = 0;
    uint64_t dims[3], bits[3];
    DevGrid grid;
    GridState *gs = nullptr;            // pending tuple runs + folded winners (grid_host.hip)
    uint64_t next_index = 0;            // file-order index the next scan starts at
    hipStream_t last_stream = nullptr;  // stream of the most recent scan: accessors wait on it
};

// ---------------------------------------------------------------------------------------------
// internal entry points (defined across the .hip files)
// ---------------------------------------------------------------------------------------------
int pcq_make_dev_pred(const pcq_predicate *p, DevPred *out);
int pcq_scratch_stream(pcq_ctx *ctx, hipStream_t s);
int pcq_ensure_partials(pcq_ctx *ctx, size_t n);

// scan_count.hip
int pcq_launch_bounds_count_xyz12(pcq_ctx *ctx, const void *d_xyz, uint64_t n, const DevPred &pred,
                                  uint64_t *d_count, hipStream_t s);
int pcq_launch_class_count_u8(pcq_ctx *ctx, const void *d_cls, uint64_t n, uint8_t cls,
                              uint64_t *d_count, hipStream_t s);
// scan_generic.hip
int pcq_launch_generic_count(pcq_ctx *ctx, const DevCols &cols, const DevPred &pred,
                             uint64_t *d_count, hipStream_t s);
int pcq_launch_emit_points(pcq_ctx *ctx, const DevCols &cols, const DevPred &pred, uint8_t *d_out31, const uint64_t *d_npoints_in,
                           uint64_t *d_npoints_out, hipStream_t s);
// grid_host.hip (kernels: grid_pass0.hip, grid_dir.hip, grid_level2.hip, grid_fold.hip, grid_finish.hip; shared: grid_common.h)
int pcq_grid_scan(pcq_ctx *ctx, pcq_collector *c, const DevCols &cols, const DevPred &pred, hipStream_t s);
void pcq_grid_release(pcq_collector *c);
int pcq_grid_drain(pcq_collector *c, pcq_point *out, uint64_t *keys_out, uint64_t cap, uint64_t *out_n);
int pcq_grid_flush(pcq_collector *c);  // folds what is pending now
// alias_sort.hip: sorted[i] = the i-th of n records (u64 key at +0, u64 order at +8) by (key, order)
int pcq_sort_by_key_then_order(pcq_ctx *ctx, const void *items, size_t stride, uint64_t n, void *sorted, hipStream_t s);
// pcq_api.hip: device-memory pool of the context.  A block may be freed only when the work that used it has completed.
int pcq_pool_alloc(pcq_ctx *ctx, size_t bytes, void **out);
void pcq_pool_free(pcq_ctx *ctx, void *p);
void pcq_pool_clear(pcq_ctx *ctx);
