/*
 * pcq.h — C ABI of the MI355X-native point-cloud predicate path ("libpcq.so").
 *
 * This is the drop-in boundary for the reference's `--optimized` predicate-evaluation hot path.
 * The reference (Rust) has no FFI for this path: the seam is the body of the four free functions
 *
 *     search_last_file_by_bounds_optimized           query/src/search/last.rs:46-166
 *     search_last_file_by_classification_optimized   query/src/search/last.rs:213-293
 *     search_las_file_by_bounds_optimized            query/src/search/las.rs:52-148
 *     search_las_file_by_classification_optimized    query/src/search/las.rs:192-261
 *
 * after they have mmapped the file and parsed the LAS header: a per-point loop that evaluates a
 * predicate over column data and pushes every match into a `&mut dyn ResultCollector`
 * (query/src/collect_points.rs:7-12).  The entry points below replace exactly that loop plus the
 * collector it feeds; file IO and header parsing stay on the host side of the boundary (in the
 * reference: readers/src + las crate; here: adhoc-queries-pointclouds_amd/host/).
 *
 * Plain pointers and sizes only.  `stream` arguments are hipStream_t passed as void*; NULL = the
 * context's own stream.  All functions return 0 (PCQ_OK) or a negative pcq_status; the message of
 * the last failure on the calling thread is available from pcq_last_error().
 * No function falls back to a CPU implementation: without a usable HIP device pcq_init fails.
 *
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add to bind these.
 */
#ifndef PCQ_H
#define PCQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCQ_ABI_VERSION 6

typedef enum pcq_status {
    PCQ_OK = 0,
    PCQ_ERR_IO = -1,          /* reserved for the host layer: File::open / mmap     (last.rs:27-34)  */
    PCQ_ERR_HEADER = -2,      /* reserved for the host layer: LAS header parse      (last.rs:53-54)  */
    PCQ_ERR_FORMAT = -3,      /* reserved for the host layer: "Invalid LAS format"  (last.rs:72-78)  */
    PCQ_ERR_EXTENSION = -4,   /* reserved for the host layer: unsupported extension (searcher.rs:84) */
    PCQ_ERR_EOF = -5,         /* column block extends past the mapped file                           */
    PCQ_ERR_GRID = -6,        /* SparseGrid::new: too many cells                (grid_sampling.rs:32)*/
    PCQ_ERR_PANIC = -7,       /* AABB::from_min_max min > max                       (last.rs:98)     */
    PCQ_ERR_ARG = -8,         /* invalid argument                                                    */
    PCQ_ERR_HIP = -9,         /* a HIP runtime call failed / no device                               */
    PCQ_ERR_CAPACITY = -10,   /* caller buffer too small; required size reported through out param   */
    PCQ_ERR_UNSUPPORTED = -11,/* valid in the reference but not representable here   */
    PCQ_ERR_NOMEM = -12
} pcq_status;

/* readers/src/lib.rs:10-19 — `#[repr(C, packed)] struct Point`, 31 bytes. */
#pragma pack(push, 1)
typedef struct pcq_point {
    double x, y, z;             /* position        @0  */
    uint16_t r, g, b;           /* color           @24 */
    uint8_t classification;     /* classification  @30 */
} pcq_point;
#pragma pack(pop)

typedef struct pcq_ctx pcq_ctx;
typedef struct pcq_collector pcq_collector;

/* ---------------------------------------------------------------------------------------------
 * Context: one per (thread, GPU).  Owns a HIP stream, pinned staging buffers and device scratch.
 * Mirrors the reference's threading model: search_file is called concurrently from rayon workers
 * (main.rs:153-161) — each worker uses its own context; a context is not thread-safe.
 * ------------------------------------------------------------------------------------------- */
int pcq_init(int device, pcq_ctx **out_ctx);
int pcq_shutdown(pcq_ctx *ctx);
const char *pcq_last_error(void);
int pcq_abi_version(void);

typedef struct pcq_device_info {
    char name[128];
    char gcn_arch[64];
    int compute_units;
    int wavefront_size;
    uint64_t hbm_bytes;
    uint64_t lds_bytes_per_block;
    int clock_khz;
} pcq_device_info;
int pcq_get_device_info(pcq_ctx *ctx, pcq_device_info *out);
/* The context's stream (hipStream_t as void*). */
void *pcq_ctx_stream(pcq_ctx *ctx);
int pcq_ctx_synchronize(pcq_ctx *ctx);

/* ---------------------------------------------------------------------------------------------
 * Column view of one file's point data (or of a chunk of it).
 * LAST (readers/src/last_reader.rs:83-144): xyz_stride 12, cls_stride 1, rgb_stride 6.
 * LAS  (las.rs:102-135): all three strides = point_data_record_length, pointers offset into the
 * record (xyz +0, cls +15/+16, rgb +20/+28).
 * scale/offset are the header's (last.rs:156-160); `first_index` is the file-order index of the
 * first point of this view (used to keep "first seen wins" across chunks and files).
 * ------------------------------------------------------------------------------------------- */
typedef struct pcq_columns {
    const void *xyz;            /* n records of {i32 x, i32 y, i32 z}, little endian            */
    const void *cls;            /* n classification bytes; may be NULL for count-only bounds scan */
    const void *rgb;            /* n records of {u16 r, u16 g, u16 b}; NULL = no colour (0,0,0)  */
    uint64_t xyz_stride;
    uint64_t cls_stride;
    uint64_t rgb_stride;
    uint64_t n;
    uint64_t first_index;
    double scale[3];
    double offset[3];
} pcq_columns;

typedef enum pcq_predicate_kind { PCQ_PRED_BOUNDS = 0, PCQ_PRED_CLASS = 1, PCQ_PRED_BOUNDS_F64 = 2 } pcq_predicate_kind;

/* The predicate.
 * BOUNDS:     in the file's local integer space: lmin <= (x,y,z) <= lmax, inclusive, compared as i64
 *             (last.rs:122-135).  lmin/lmax are what pcq_box_to_local produced; values outside the i32
 *             range are legal.
 * CLASS:      classification byte == cls, whole byte (last.rs:259-262).
 * BOUNDS_F64: in world space, the reference's non-integer form used by the LAZER scan
 *             (query/src/search/lazer.rs:65-69 on positions rebuilt as offset + scale * x,
 *             readers/src/lazer_reader.rs:600-607): wmin <= world <= wmax per axis, inclusive
 *             (pasture AABB::contains). */
typedef struct pcq_predicate {
    int32_t kind;               /* pcq_predicate_kind */
    uint8_t cls;
    uint8_t _pad[3];
    int64_t lmin[3];
    int64_t lmax[3];
    double wmin[3];
    double wmax[3];
} pcq_predicate;

/* last.rs:98-109 / las.rs:88-99 — f64 query box -> local integer box, bug-for-bug (all three min
 * components divide by scale[0]; `as i64` truncation/saturation).  PCQ_ERR_PANIC if min > max. */
int pcq_box_to_local(const double bmin[3], const double bmax[3], const double scale[3],
                     const double offset[3], int64_t lmin[3], int64_t lmax[3]);

/* ---------------------------------------------------------------------------------------------
 * Collectors — device-resident counterparts of query/src/collect_points.rs.
 *   count  : CountCollector        (:72-98)   a u64 counter in HBM
 *   buffer : BufferCollector       (:14-44)   matches appended in file order (stable compaction)
 *   grid   : GridSampledCollector  (:100-127) SparseGrid (grid_sampling.rs:9-114): per cell the point closest to
 *            the cell centre, first seen wins ties.  No hash table in HBM: a scan leaves its matches as 16-byte tuples
 *            sorted by a hash bin of their cell, and a fold — when a result is asked for, or pcq_collector_flush —
 *            resolves every bin in an LDS table (csrc/grid_*.hip).
 * A collector may be fed by several scans (sequential mode feeds all files into one collector,
 * main.rs:129-133); scans into one collector must be issued in file order.
 * ------------------------------------------------------------------------------------------- */
int pcq_collector_new_count(pcq_ctx *ctx, pcq_collector **out);
/* Count collector whose counter lives in caller-owned device memory (8 bytes, zeroed by the
 * caller) — so that the caller can all-reduce it with RCCL (main.rs:164-180); pcq_allreduce_sum_u64 does that OUT OF
 * PLACE (the per-GPU counts stay what they were, whatever happens to the collective). */
int pcq_collector_new_count_at(pcq_ctx *ctx, uint64_t *device_counter, pcq_collector **out);
int pcq_collector_new_buffer(pcq_ctx *ctx, pcq_collector **out);
int pcq_collector_new_grid(pcq_ctx *ctx, const double bmin[3], const double bmax[3],
                           double cell_size, pcq_collector **out);
int pcq_collector_free(pcq_collector *c);
/* ResultCollector::point_count (synchronises the collector's stream). */
int pcq_collector_point_count(pcq_collector *c, uint64_t *out);
/* 1 when points()/points_ref() is Some (buffer, grid), 0 for the count collector. */
int pcq_collector_has_points(const pcq_collector *c);
/* ResultCollector::points: copies up to cap points to host memory, *out_n = number available
 * (PCQ_ERR_CAPACITY when cap is too small; call with out=NULL, cap=0 to size).
 * buffer: file order.  grid: hash-table slot order (the reference's HashMap order is unspecified);
 * pcq_collector_grid_cells returns the cell keys in the same order. */
int pcq_collector_points(pcq_collector *c, pcq_point *out, uint64_t cap, uint64_t *out_n);
int pcq_collector_grid_cells(pcq_collector *c, uint64_t *out, uint64_t cap, uint64_t *out_n);
/* SparseGrid::new results (grid_sampling.rs:24-44). */
int pcq_collector_grid_params(const pcq_collector *c, uint64_t dims[3], uint64_t bits[3]);
/* Resets the collector to its freshly constructed state (keeps allocations). */
int pcq_collector_reset(pcq_collector *c);
/* Brings a collector that is kept for later into its compact form NOW and waits for it: a grid collector folds its
 * pending matches into per-cell winners (the reference's HashMap never holds more than the winners,
 * grid_sampling.rs:72-103; the device keeps a tuple per scanned point until it folds) and returns their memory to the
 * context; count and buffer collectors only wait for their scans.  run_search_parallel calls it when a file is done
 * (main.rs:153-161 keeps one collector per file until all files are searched).  Results are unchanged. */
int pcq_collector_flush(pcq_collector *c);

/* ---------------------------------------------------------------------------------------------
 * The scan: evaluates `pred` over `cols` and pushes every match into `c`
 * (the loop at last.rs:117-164 / :253-291 / las.rs:101-146 / :221-259).
 *   pcq_scan_dev : column pointers are DEVICE memory; kernels are enqueued on `stream`
 *                  (asynchronous: results are complete after the stream is synchronised or a
 *                  collector accessor is called).
 *   pcq_scan_host: column pointers are HOST memory (e.g. the mmapped file); the library streams
 *                  them through pinned double buffers with hipMemcpyAsync overlapped with the
 *                  kernels, in chunks; returns when the scan is complete.
 * ------------------------------------------------------------------------------------------- */
int pcq_scan_dev(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c,
                 void *stream);
int pcq_scan_host(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *pred,
                  pcq_collector *c);
/* Same as pcq_scan_host, but the column "pointers" of `cols` are BYTE OFFSETS into the open file
 * `fd` (e.g. offset_to_point_data for xyz): chunks are pread() straight into the pinned staging
 * buffers, which avoids the page-table work of reading through a fresh mmap (last.rs:27-34). */
int pcq_scan_fd(pcq_ctx *ctx, int fd, const pcq_columns *cols, const pcq_predicate *pred,
                pcq_collector *c);
/* pcq_scan_host that returns as soon as the caller's memory has been read (copied into the staging buffers):
 * the caller may overwrite its columns at once and issue the next scan, whose staging copy then overlaps this
 * scan's transfer and kernels.  Results are complete after pcq_ctx_synchronize or any collector accessor. */
int pcq_scan_host_nowait(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c);
/* The same for pcq_scan_fd: returns when the file's last chunk has been read and its kernels are enqueued — the descriptor may
 * be closed at once, and the next file's first chunk is read while this one's last transfer and kernels run (a per-file
 * synchronisation drained the pipeline for about a chunk's read + transfer at every file boundary of main.rs:153-161). */
int pcq_scan_fd_nowait(pcq_ctx *ctx, int fd, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c);
/* Optional, for a caller that is going to scan from host memory or files: starts — on a thread of its own, and returns at
 * once — what the first such scan of a context otherwise does before it can move a byte: pinning the two staging buffers
 * (4 ms each) and starting the copy helpers.  Called right behind pcq_init this runs while the caller opens its first file
 * and creates its collector; the first scan (or pcq_shutdown) waits for whatever is left of it.  A failure here is not
 * reported here: the first scan repeats the allocation and reports it. */
int pcq_prepare_host_scans(pcq_ctx *ctx);

/* Count-only scan of many device-resident LAST files in ONE launch (files = independent units,
 * main.rs:153-161): segment i is scanned with preds[i] (all bounds, over 16-byte aligned positions
 * blocks — or all class, over classification blocks of any alignment); the total is ADDED to
 * *device_total. */
int pcq_scan_dev_count_batch(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *preds,
                             size_t nsegments, uint64_t *device_total, void *stream);

/* ---------------------------------------------------------------------------------------------
 * On-the-fly chunk index for device-resident LAST columns — the reference authors' own next step
 * (improvements.md:3-10), SURVEY.md §8f-3.  The first bounds (class) count scan of a file through
 * pcq_scan_dev_indexed also records the integer AABB of every 4096-point chunk (a 256-bin class
 * histogram per 65536-point chunk); later scans of the SAME columns consult it: chunks disjoint from
 * the box are skipped, chunks inside it are counted without being read, only straddling chunks are
 * scanned (class counts are answered from the histograms alone).  Results are identical to
 * pcq_scan_dev; count collectors only; layouts the index does not cover fall through to pcq_scan_dev.
 * ------------------------------------------------------------------------------------------- */
typedef struct pcq_index pcq_index;
typedef struct pcq_index_stats {
    uint64_t chunks;   /* chunks covered by the last indexed scan                       */
    uint64_t skipped;  /* ... disjoint from the query box: not read                     */
    uint64_t whole;    /* ... inside the box / answered from the histogram: not read    */
    uint64_t scanned;  /* ... read                                                      */
    uint64_t built;    /* 1 when the last scan built the index (it read everything)     */
} pcq_index_stats;
int pcq_index_new(pcq_ctx *ctx, pcq_index **out);
int pcq_index_free(pcq_index *ix);
int pcq_index_get_stats(pcq_index *ix, pcq_index_stats *out);
int pcq_scan_dev_indexed(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *pred, pcq_index *ix,
                         pcq_collector *c, void *stream);

/* The one collective of the path (main.rs:164-180) for callers that drive n GPUs from ONE process:
 * recv[i][0] = the sum over i of send[i][0] (8 bytes each in ctxs[i]'s HBM, e.g. the counter of
 * pcq_collector_new_count_at) — a single RCCL all-reduce(sum, u64, count = 1) over an intra-node communicator (xGMI).
 * Synchronous.  recv[i] may be send[i]; callers that want to fall back to summing the per-GPU values themselves when the
 * collective fails pass a different word, so that send[] is never touched.  One rank per GPU (two entries on one device
 * are refused).  n == 1 copies send to recv without RCCL unless option "allreduce_single_rank" is set on ctxs[0] (then
 * the one-rank communicator and the all-reduce really run).  RCCL is bound at run time; failure to find it is an error.
 * pcq_allreduce_prepare(devices, n) builds the communicator for that device list NOW (synchronous, thread-safe: loading RCCL
 * takes a fresh process 1-5 s, ncclCommInitAll 0.6 s), so that the all-reduce finds it ready; a caller that wants that beside
 * its scans calls it from a thread of its own and joins it before the all-reduce.  The library starts no thread. */
int pcq_allreduce_prepare(const int *devices, int n);
int pcq_allreduce_sum_u64(pcq_ctx *const *ctxs, const uint64_t *const *send, uint64_t *const *recv, int n);

/* Device memory helpers for callers that keep column blocks resident in HBM. */
int pcq_device_alloc(pcq_ctx *ctx, uint64_t bytes, void **out);
int pcq_device_free(pcq_ctx *ctx, void *p);
int pcq_copy_to_device(pcq_ctx *ctx, void *dst_device, const void *src_host, uint64_t bytes);
int pcq_copy_to_host(pcq_ctx *ctx, void *dst_host, const void *src_device, uint64_t bytes);
int pcq_device_memset(pcq_ctx *ctx, void *dst_device, int value, uint64_t bytes, void *stream);

/* Reads [file_offset, file_offset + bytes) of an open file into device memory at the rate of the host block path (pinned
 * double buffering, parallel pread) — for callers that keep column blocks resident in HBM.  Synchronous. */
int pcq_read_fd_to_device(pcq_ctx *ctx, int fd, uint64_t file_offset, uint64_t bytes, void *d_dst);

/* Restricts the CALLING thread to the CPUs of the NUMA node the context's GPU is attached to (no-op when the
 * node is unknown or option "numa_local" is 0).  For caller threads that produce the bytes a scan will read —
 * memory they touch first then sits next to the GPU's staging buffers. */
int pcq_bind_thread_near_device(pcq_ctx *ctx);

/* Options: "blocks_per_cu" (persistent blocks per CU of the strided count kernels), "chunk_points" (points per staging
 * chunk of the host paths), "copy_threads" (threads filling a staging chunk, default 16), "numa_local",
 * "allreduce_single_rank", "allreduce_fail" (tests: 1 = pcq_allreduce_sum_u64 fails before it touches anything, 2 = after the
 * reduction ran, 3 = inside the RCCL group, behind rank 0), "grid_pending_budget" (points a grid collector may hold unfolded; 0 = default), "grid_agg" (pass 0 folds a
 * tile's duplicate cells before they travel: 0 = while it pays, 1 = every tile, 2 = never; same results in every mode),
 * "grid_f2" (tests: the second-level fan-out a fold starts from; 0 = from the measured estimate), "host_in_place" (host / file
 * scans with a count or grid collector read the pinned staging ring in place over PCIe: 0 never, 1 always, 2 = while the
 * process's copy path is being set up — the default), "emit_park_max" / "emit_sparse_max" (buffer collector: a 2048-point tile with at
 * most that many matches leaves them as 16-byte words in the count pass — default 256, bounds queries —
 * or is written by one wave from the count pass's match bits — default 64; 0 = never), "grid_tuple16" / "grid_stream" (tests: force
 * the grid collector's tuple size — 1 = 16 bytes with the selector, 2 = without, 0 = 24 bytes — and the coarse fold's form —
 * 1 = k_fold_stream, 0 = k_fold<BIG>; same results in every combination).  pcq_get_option also
 * reads "numa_node" and the grid diagnostics "grid_folds", "grid_level2" (folds that needed a second partition level),
 * "grid_refolds" (folds repeated with more partitions), "grid_level2_exact" (second levels repeated in the counting form),
 * "grid_compactions" (folds that copied short fragments together first), "grid_last_f2", "grid_last_tuples" (tuples the last
 * fold found pending: what is left of the matches after pass 0's own fold).  (The kernel-shape experiments of round 1 —
 * "k1_variant", "batch_variant", ... — are options of libpcq_lab.so only: include/pcq_lab.h.) */
int pcq_set_option(pcq_ctx *ctx, const char *key, int64_t value);
int pcq_get_option(pcq_ctx *ctx, const char *key, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* PCQ_H */
