/*
 * pcq_oracle.h — CPU restatement ("oracle") of the reference's ad-hoc predicate path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there only
 * as the checker / reported CPU baseline, never as the thing shipped or measured as the product.
 *
 * PARITY STATUS: the reference is Rust; no Rust toolchain exists in the build container, so the
 * reference cannot be compiled or run.  The only golden vectors the reference's own
 * tests hold for this path are the three SparseGrid tests (query/src/grid_sampling.rs:116-209);
 * the oracle is pinned against those (tests/test_oracle_golden.py).  For the bounds / class scans,
 * the box conversion, collectors and CLI output the reference holds no test: for those rows
 * "parity unpinned" by reference fixtures — they are pinned by hand-derived known-answer vectors
 * (tests/golden/) that follow the cited reference lines.
 *
 * Every function cites the reference file:line (relative to /root/reference) that it restates.
 * Compile with -O2 -ffp-contract=off (Rust never contracts a*b+c into an fma).
 */
#ifndef PCQ_ORACLE_H
#define PCQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes (mirrored by include/pcq.h so tests can compare error classes). */
enum {
    PCQO_OK = 0,
    PCQO_ERR_IO = -1,           /* File::open / mmap failed                     (last.rs:27-34)        */
    PCQO_ERR_HEADER = -2,       /* raw::Header::read_from / Header::from_raw    (last.rs:53-54)        */
    PCQO_ERR_FORMAT = -3,       /* "Invalid LAS format {} in file {}"           (last.rs:72-78)        */
    PCQO_ERR_EXTENSION = -4,    /* "Unsupported file extension in file {}"      (searcher.rs:84-88)    */
    PCQO_ERR_EOF = -5,          /* read past the end of the mmap (UnexpectedEof)                       */
    PCQO_ERR_GRID = -6,         /* "Too many cells ... in SparseGrid"           (grid_sampling.rs:32-34)*/
    PCQO_ERR_PANIC = -7,        /* AABB::from_min_max min > max panic           (last.rs:98)           */
    PCQO_ERR_ARG = -8           /* bad argument to the oracle API itself                               */
};

/* readers/src/lib.rs:10-19 — #[repr(C, packed)] Point, 31 bytes. */
#pragma pack(push, 1)
typedef struct {
    double x, y, z;             /* position        @0  */
    uint16_t r, g, b;           /* color           @24 */
    uint8_t classification;     /* classification  @30 */
} pcqo_point;
#pragma pack(pop)

/* Parsed LAS header: the fields of las 0.7.4 raw::Header that the path consumes.
 * Byte layout restated from query/src/las.rs:7-40 (LAS 1.2, 227 bytes) plus the 1.3/1.4 tails. */
typedef struct {
    uint8_t version_major, version_minor;
    uint16_t header_size;
    uint32_t offset_to_point_data;
    uint32_t number_of_vlrs;
    uint8_t point_data_record_format;   /* raw byte, unmasked */
    uint16_t point_data_record_length;
    uint32_t legacy_number_of_points;
    uint64_t large_number_of_points;    /* LAS 1.4 only, else 0 */
    double scale[3];
    double offset[3];
    double min[3];
    double max[3];
    uint64_t number_of_points;          /* Header::number_of_points() */
} pcqo_las_header;

const char *pcqo_last_error(void);

/* Rust `f64 as i64` / `f64 as u64`: truncate toward zero, saturate, NaN -> 0 (Rust >= 1.45). */
int64_t pcqo_f64_as_i64(double v);
uint64_t pcqo_f64_as_u64(double v);

/* raw::Header::read_from + Header::from_raw.  mask_format != 0 applies `&= 0b1111` first
 * (last.rs:222; the bounds path does not mask, last.rs:53-54). */
int pcqo_parse_las_header(const uint8_t *data, size_t len, int mask_format, pcqo_las_header *out);

/* last.rs:98-109 / las.rs:88-99 — query box -> integer box in the file's local space, bug-for-bug
 * (all three min components divide by x_scale).  Returns PCQO_ERR_PANIC when min > max on an axis. */
int pcqo_box_to_local(const double bmin[3], const double bmax[3], const double scale[3],
                      const double offset[3], int64_t lmin[3], int64_t lmax[3]);

/* pasture AABB::intersects (inclusive) — last.rs:92. */
int pcqo_aabb_intersects(const double amin[3], const double amax[3], const double bmin[3],
                         const double bmax[3]);

/* ---- collectors: query/src/collect_points.rs:7-127 ---- */
typedef struct pcqo_collector pcqo_collector;
enum { PCQO_COLLECT_COUNT = 0, PCQO_COLLECT_BUFFER = 1, PCQO_COLLECT_GRID = 2 };

pcqo_collector *pcqo_collector_new_count(void);                   /* collect_points.rs:72-98   */
pcqo_collector *pcqo_collector_new_buffer(void);                  /* collect_points.rs:14-44   */
/* collect_points.rs:100-127 + grid_sampling.rs:18-47; NULL (+ last_error) when bits > 64. */
pcqo_collector *pcqo_collector_new_grid(const double bmin[3], const double bmax[3], double cell);
void pcqo_collector_free(pcqo_collector *c);
int pcqo_collector_kind(const pcqo_collector *c);
void pcqo_collector_collect_one(pcqo_collector *c, const pcqo_point *p);
uint64_t pcqo_collector_point_count(const pcqo_collector *c);
/* points()/points_ref(): returns 0 when the collector yields None (count collector), else 1. */
int pcqo_collector_has_points(const pcqo_collector *c);
/* Copies up to cap points; returns the number available.  Buffer: file order.  Grid: ascending
 * cell key (the reference iterates a HashMap: order unspecified; we canonicalise for comparison). */
uint64_t pcqo_collector_points(const pcqo_collector *c, pcqo_point *out, uint64_t cap);
/* Grid only: the occupied cell keys in ascending order (SparseGrid::cells, grid_sampling.rs:107). */
uint64_t pcqo_collector_grid_cells(const pcqo_collector *c, uint64_t *out, uint64_t cap);
/* Grid only: dims[3], bits[3] (grid_sampling.rs:24-44). */
int pcqo_collector_grid_params(const pcqo_collector *c, uint64_t dims[3], uint64_t bits[3]);

/* ---- the four optimized scans, on a memory image of the file ---- */
/* last.rs:46-166 */
int pcqo_search_last_mem_by_bounds_optimized(const uint8_t *data, size_t len, const double bmin[3],
                                             const double bmax[3], pcqo_collector *c);
/* last.rs:213-293 */
int pcqo_search_last_mem_by_classification_optimized(const uint8_t *data, size_t len, uint8_t cls,
                                                     pcqo_collector *c);
/* las.rs:52-148 — *record_size_printed receives point_data_record_length when the reference's
 * `println!("Point record size: {}")` (las.rs:73) is reached, else is left untouched. */
int pcqo_search_las_mem_by_bounds_optimized(const uint8_t *data, size_t len, const double bmin[3],
                                            const double bmax[3], pcqo_collector *c,
                                            int *record_size_printed);
/* las.rs:192-261 */
int pcqo_search_las_mem_by_classification_optimized(const uint8_t *data, size_t len, uint8_t cls,
                                                    pcqo_collector *c);

/* ---- LAZER (lazer_oracle.c): query/src/search/lazer.rs:34-116 over readers/src/lazer_reader.rs ---- */
int pcqo_search_lazer_mem_by_bounds(const uint8_t *data, size_t len, const double bmin[3], const double bmax[3],
                                    pcqo_collector *c);
/* bug-for-bug: the point buffer is never cleared, so every chunk re-filters the first chunk's points */
int pcqo_search_lazer_mem_by_classification(const uint8_t *data, size_t len, uint8_t cls, pcqo_collector *c);
/* LAZERSource::from(..).get_metadata().bounds() (main.rs:102-111) */
int pcqo_lazer_mem_bounds(const uint8_t *data, size_t len, double mn[3], double mx[3]);
int pcqo_lazer_file_bounds(const char *path, double mn[3], double mx[3]);
/* lz4::Decoder restatement: the first `need` bytes of the LZ4 frame at src, pulled with read_exact calls of
 * `unit` bytes (0: one call); returns need or a PCQO_ERR_*. */
int64_t pcqo_lz4f_decode(const uint8_t *src, size_t n, uint8_t *out, size_t need, size_t unit);
uint32_t pcqo_xxh32(const uint8_t *d, size_t n);
/* Test-side writers (the reference repository has no LAZER writer).  flags: 1 independent blocks,
 * 2 block checksums, 4 content checksum, 8 content size, 16 stored blocks, 32 leading skippable frame;
 * block_id 4..7 = 64 KiB .. 4 MiB.  Results are malloc'd; release with pcqo_free. */
uint8_t *pcqo_lz4f_compress(const uint8_t *content, size_t n, unsigned flags, int block_id, size_t *out_n);
uint8_t *pcqo_lazer_from_last(const uint8_t *last, size_t len, uint64_t block_size, unsigned flags, int block_id,
                              size_t *out_n);
void pcqo_free(void *p);

/* ---- file-level: open + mmap (last.rs:27-34) then the scans above; dispatch by extension
 *      restates searcher.rs:43-152 for the ("las"|"last", Optimized) and "lazer" arms. ---- */
enum { PCQO_QUERY_BOUNDS = 0, PCQO_QUERY_CLASS = 1 };
int pcqo_search_file(const char *path, int query_kind, const double bmin[3], const double bmax[3],
                     uint8_t cls, pcqo_collector *c, int *record_size_printed);

/* Multi-threaded count scan used as the CPU baseline: one thread per file image, at most
 * `threads` at a time (rayon par_iter, main.rs:153-161), counts summed (main.rs:164-180). */
int pcqo_count_files_parallel(const uint8_t *const *images, const size_t *lens, size_t nfiles,
                              int query_kind, const double bmin[3], const double bmax[3],
                              uint8_t cls, int threads, uint64_t *total);

/* ---- deterministic synthetic data, integer-only so that the device generator
 *      in the product's bench support (include/pcq_synth.h) is bit-identical ---- */
#define PCQO_SYNTH_MAX_CLASSES 8
typedef struct {
    uint64_t seed;
    uint64_t n;
    uint32_t format;            /* LAS point format 0..3 */
    uint32_t n_classes;
    double scale[3];
    double offset[3];
    int32_t lo[3];              /* inclusive integer lower corner                      */
    uint32_t span[3];           /* number of distinct integer values per axis (>= 1)   */
    uint32_t zo_prob16;         /* P(z outlier) * 65536                                */
    int32_t zo_lo;              /* outlier z range                                     */
    uint32_t zo_span;
    uint32_t cls_cum16[PCQO_SYNTH_MAX_CLASSES]; /* cumulative thresholds over 65536    */
    uint8_t cls_val[PCQO_SYNTH_MAX_CLASSES];
} pcqo_synth_spec;

uint64_t pcqo_synth_mix(uint64_t seed, uint64_t k);
/* Fill columns for points [first, first+count): xyz = count*3 int32, cls = count bytes (either may be NULL). */
void pcqo_synth_fill_columns(const pcqo_synth_spec *s, uint64_t first, uint64_t count, int32_t *xyz,
                             uint8_t *cls);
/* Size in bytes of the LAS/LAST image (227-byte LAS 1.2 header + n * record_len). */
size_t pcqo_synth_image_size(const pcqo_synth_spec *s);
/* Build a complete file image.  transposed != 0 -> LAST (column blocks), else LAS (AoS). */
int pcqo_synth_build_image(const pcqo_synth_spec *s, int transposed, uint8_t *out, size_t cap,
                           int threads);
int pcqo_synth_write_file(const pcqo_synth_spec *s, int transposed, const char *path, int threads);
/* Header-only image (227 bytes) for a spec. */
int pcqo_synth_build_header(const pcqo_synth_spec *s, uint8_t out[227]);

#ifdef __cplusplus
}
#endif
#endif
