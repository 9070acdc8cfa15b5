/*
 * pcq_query.h — C view of the C++ host layer (libpcq_query.so), so that tests and other languages
 * can call the file-level operators of the reference by name:
 *
 *   Searcher::search_file for BoundsSearcher / ClassSearcher     query/src/search/searcher.rs:24-152
 *   CountCollector / BufferCollector / GridSampledCollector      query/src/collect_points.rs:14-127
 *   parse_aabb, get_all_input_files, is_valid_file, get_total_bounds   query/src/main.rs:29-120, 185-189
 *
 * libpcq_query.so contains NO scan code: every function below that touches point data calls the
 * HIP library through include/pcq.h (pcq_scan_host).  Functions return 0 or a negative pcq_status;
 * pcq_query_last_error() returns the message of the last failure on the calling thread and
 * pcq_query_last_was_panic() tells whether the reference would have panicked (exit code 101).
 */
#ifndef PCQ_QUERY_H
#define PCQ_QUERY_H

#include "pcq.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcq_host_collector pcq_host_collector;

const char *pcq_query_last_error(void);
int pcq_query_last_was_panic(void);

/* raw::Header::read_from + Header::from_raw on a memory image (no GPU needed). */
typedef struct pcq_las_header_info {
    uint8_t version_major, version_minor;
    uint8_t point_data_record_format;
    uint8_t _pad;
    uint16_t header_size;
    uint16_t point_data_record_length;
    uint32_t offset_to_point_data;
    uint32_t _pad2;
    uint64_t number_of_points;
    double scale[3], offset[3], min[3], max[3];
} pcq_las_header_info;
int pcq_query_parse_las_header(const uint8_t *data, size_t len, int mask_format, pcq_las_header_info *out);

/* main.rs:59-92 (PCQ_ERR_PANIC when min > max), :185-189, :94-120 — no GPU needed. */
int pcq_query_parse_aabb(const char *s, double bmin[3], double bmax[3]);
int pcq_query_is_valid_file(const char *path);
int pcq_query_get_total_bounds(const char *const *files, size_t nfiles, double bmin[3], double bmax[3]);

/* The LZ4 Frame reader the LAZER searches inflate column blobs with (stand-in for lz4::Decoder,
 * readers/src/lazer_reader.rs:176-265): the first `need` bytes of the frame at src -> out (cap >= need),
 * as read_exact calls of `unit` bytes each would get them (4 = read_i32 ..., 0 = a single call).
 * Host-only, no GPU needed.  PCQ_ERR_EOF = read_exact's UnexpectedEof, PCQ_ERR_HEADER = an LZ4 error. */
int pcq_query_lz4_frame_decode(const uint8_t *src, size_t n, uint64_t need, uint64_t unit, uint8_t *out, uint64_t cap);

/* Collectors on the calling thread's context for `device`. */
int pcq_query_collector_new_count(int device, pcq_host_collector **out);
int pcq_query_collector_new_buffer(int device, pcq_host_collector **out);
int pcq_query_collector_new_grid(int device, const double bmin[3], const double bmax[3], double cell_size,
                                 pcq_host_collector **out);
int pcq_query_collector_free(pcq_host_collector *c);
int pcq_query_collector_point_count(pcq_host_collector *c, uint64_t *out);
/* 0: points() is None; 1: Some.  Copies up to cap points (buffer: file order). */
int pcq_query_collector_has_points(pcq_host_collector *c);
int pcq_query_collector_points(pcq_host_collector *c, pcq_point *out, uint64_t cap, uint64_t *out_n);
int pcq_query_collector_grid_cells(pcq_host_collector *c, uint64_t *out, uint64_t cap, uint64_t *out_n);

/* BoundsSearcher::search_file / ClassSearcher::search_file with SearchImplementation
 * (0 = Regular, 1 = Optimized).  *las_record_size receives the value the reference prints with
 * `Point record size: {}` (las.rs:73) or is left at -1. */
int pcq_query_search_file_bounds(const char *path, const double bmin[3], const double bmax[3], int optimized,
                                 pcq_host_collector *c, int *las_record_size);
int pcq_query_search_file_class(const char *path, uint8_t cls, int optimized, pcq_host_collector *c);

/* A dataset resident in HBM (host/resident.cpp; not in the reference, which re-reads the files for every query): the
 * positions and classification blocks of LAST files are loaded into `device`'s HBM once; every count query over them
 * is the reference's per-file host prologue (early-out last.rs:92-94, box conversion :98-109) + ONE batched launch
 * (pcq_scan_dev_count_batch).  Same counts as `query --bounds|--class ... --optimized --parallel` on those files. */
typedef struct pcq_host_resident pcq_host_resident;
int pcq_query_resident_load(int device, const char *const *files, size_t nfiles, pcq_host_resident **out);
int pcq_query_resident_free(pcq_host_resident *r);
int pcq_query_resident_count_bounds(pcq_host_resident *r, const double bmin[3], const double bmax[3], uint64_t *matches,
                                    uint64_t *points_scanned);
int pcq_query_resident_count_class(pcq_host_resident *r, uint8_t cls, uint64_t *matches, uint64_t *points_scanned);

/* The whole CLI in-process (main.rs:191-319); returns the exit code. */
int pcq_query_main(int argc, const char *const *argv);

/* Test entries (the `query` binary cannot reach them).
 * pcq_query_main_with_hooks: the CLI with the parallel driver's test hooks — device_slots: a "0,0"-style device list, repeats
 * allowed (two device SLOTS on one GPU run the N > 1 paths of main.rs:146-183's merge), or NULL; allreduce_fail: 0, or make the
 * count merge's collective fail through the real RCCL calls, 1 = before anything is touched, 2 = after the reduction ran.
 * pcq_query_simulate_schedule: the file -> device-slot schedule of the parallel driver (every slot starts with its own
 * longest-processing-time share; a slot that runs dry takes from the fullest) when slot k's context is ready at ready_ms[k]
 * and a file costs ms_per_unit x cost[i]; home_slot (may be NULL) = the share a file started in.  No GPU involved. */
int pcq_query_main_with_hooks(int argc, const char *const *argv, const char *device_slots, int allreduce_fail);
int pcq_query_simulate_schedule(const uint64_t *cost, size_t nfiles, const double *ready_ms, int nslots, double ms_per_unit,
                                int *slot_of_file, int *home_slot, double *makespan_ms);
/* Test entry: the two halves of a LAST bounds search with something in between — the file's plan is made (header, offsets,
 * box: the host prologue of run_search_parallel), then, if `replacement` is not NULL, that file is renamed over `path`, then the
 * plan is executed.  A plan must not be executed on another file under the same name. */
int pcq_query_test_plan_replace_execute(const char *path, const char *replacement, const double bmin[3], const double bmax[3],
                                        pcq_host_collector *c);

#ifdef __cplusplus
}
#endif
#endif
