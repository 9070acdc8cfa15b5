/*
 * pcq_oracle.c — CPU restatement of the reference's `--optimized` predicate path.
 *
 * TEST INFRASTRUCTURE ONLY (see pcq_oracle.h).  Not linked into, loaded by, or called from the
 * product (libpcq.so, libpcq_query.so, the `query` CLI).
 *
 * Each function follows the reference lines it cites, including their quirks.
 * Third-party behaviour that lives in crates absent from /root/reference is restated from the
 * published crate sources and marked [recalled]:
 *   las 0.7.4 (Cargo.lock:401-403)           raw::Header::read_from, Header::from_raw
 *   pasture-core 0.1.0 (Cargo.lock:740-742)  AABB::{from_min_max, intersects}
 *   nalgebra 0.23.2                          distance_squared = dx*dx + dy*dy + dz*dz, (a+b)+c
 */
#define _GNU_SOURCE
#include "pcq_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static __thread char g_err[512];

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const char *pcqo_last_error(void) { return g_err; }
int pcqo_fail_msg(int code, const char *msg) { return fail(code, "%s", msg); } /* for lazer_oracle.c */

/* ------------------------------------------------------------------------------------------ */
/* Rust numeric cast semantics                                                                */
/* ------------------------------------------------------------------------------------------ */

int64_t pcqo_f64_as_i64(double v) {
    if (v != v) return 0;
    if (v >= 9223372036854775808.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v; /* C truncates toward zero, in range here */
}

uint64_t pcqo_f64_as_u64(double v) {
    if (v != v) return 0;
    if (v <= 0.0) return 0;
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)v;
}

/* ------------------------------------------------------------------------------------------ */
/* little-endian readers with Cursor<Mmap> semantics: a read past the end is UnexpectedEof     */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    const uint8_t *p;
    size_t len;
} image;

static int rd_u8(const image *im, uint64_t off, uint8_t *v) {
    if (off >= im->len) return 0;
    *v = im->p[off];
    return 1;
}
static int rd_u16(const image *im, uint64_t off, uint16_t *v) {
    if (off > im->len || im->len - off < 2) return 0;
    *v = (uint16_t)(im->p[off] | (im->p[off + 1] << 8));
    return 1;
}
static int rd_i32(const image *im, uint64_t off, int32_t *v) {
    if (off > im->len || im->len - off < 4) return 0;
    uint32_t u = (uint32_t)im->p[off] | ((uint32_t)im->p[off + 1] << 8) |
                 ((uint32_t)im->p[off + 2] << 16) | ((uint32_t)im->p[off + 3] << 24);
    *v = (int32_t)u;
    return 1;
}
static uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t le32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
static double lef64(const uint8_t *p) {
    uint64_t u = le64(p);
    double d;
    memcpy(&d, &u, 8);
    return d;
}

/* ------------------------------------------------------------------------------------------ */
/* LAS header — field order per query/src/las.rs:7-40; las 0.7.4 raw::Header::read_from and   */
/* Header::from_raw [recalled]                                                                */
/* ------------------------------------------------------------------------------------------ */

static const uint16_t k_format_len[11] = {20, 28, 26, 34, 57, 63, 30, 36, 38, 59, 67};

int pcqo_parse_las_header(const uint8_t *data, size_t len, int mask_format, pcqo_las_header *h) {
    memset(h, 0, sizeof *h);
    if (len < 4) return fail(PCQO_ERR_HEADER, "failed to fill whole buffer");
    if (memcmp(data, "LASF", 4) != 0) return fail(PCQO_ERR_HEADER, "invalid file signature");
    if (len < 227) return fail(PCQO_ERR_HEADER, "failed to fill whole buffer");
    h->version_major = data[24];
    h->version_minor = data[25];
    h->header_size = le16(data + 94);
    h->offset_to_point_data = le32(data + 96);
    h->number_of_vlrs = le32(data + 100);
    h->point_data_record_format = data[104];
    h->point_data_record_length = le16(data + 105);
    h->legacy_number_of_points = le32(data + 107);
    for (int a = 0; a < 3; a++) {
        h->scale[a] = lef64(data + 131 + 8 * a);
        h->offset[a] = lef64(data + 155 + 8 * a);
        h->max[a] = lef64(data + 179 + 16 * a); /* on disk: max_x,min_x,max_y,min_y,max_z,min_z */
        h->min[a] = lef64(data + 187 + 16 * a);
    }
    /* raw::Header::read_from: version-dependent tails, then padding up to header_size. */
    size_t fixed = 227;
    int v13 = (h->version_major == 1 && h->version_minor >= 3) || h->version_major > 1;
    int v14 = (h->version_major == 1 && h->version_minor >= 4) || h->version_major > 1;
    if (v13) fixed += 8;
    if (v14) {
        fixed += 12 + 8 + 15 * 8;
        if (len < fixed) return fail(PCQO_ERR_HEADER, "failed to fill whole buffer");
        h->large_number_of_points = le64(data + 247);
    }
    if (len < fixed) return fail(PCQO_ERR_HEADER, "failed to fill whole buffer");
    if (h->header_size > fixed && len < h->header_size)
        return fail(PCQO_ERR_HEADER, "failed to fill whole buffer");

    if (mask_format) h->point_data_record_format &= 0x0F; /* last.rs:222 */

    /* Header::from_raw -> Builder::new / into_header [recalled]. */
    uint8_t fmt = h->point_data_record_format;
    if (fmt > 10) return fail(PCQO_ERR_HEADER, "invalid point format number: %u", fmt);
    if (h->point_data_record_length < k_format_len[fmt])
        return fail(PCQO_ERR_HEADER, "point data record length %u too small for format %u",
                    h->point_data_record_length, fmt);
    if (fmt >= 6 && !v14)
        return fail(PCQO_ERR_HEADER, "version %u.%u does not support point format %u",
                    h->version_major, h->version_minor, fmt);
    h->number_of_points = h->legacy_number_of_points > 0 ? (uint64_t)h->legacy_number_of_points
                                                         : h->large_number_of_points;
    return PCQO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* AABB helpers — pasture-core 0.1.0 [recalled]                                               */
/* ------------------------------------------------------------------------------------------ */

int pcqo_aabb_intersects(const double amin[3], const double amax[3], const double bmin[3],
                         const double bmax[3]) {
    for (int a = 0; a < 3; a++)
        if (!(amin[a] <= bmax[a] && amax[a] >= bmin[a])) return 0;
    return 1;
}

/* last.rs:98-109, las.rs:88-99 */
int pcqo_box_to_local(const double bmin[3], const double bmax[3], const double scale[3],
                      const double offset[3], int64_t lmin[3], int64_t lmax[3]) {
    for (int a = 0; a < 3; a++) {
        /* NOTE: x_scale_factor on all three axes of the min corner (last.rs:100-102). */
        lmin[a] = pcqo_f64_as_i64((bmin[a] - offset[a]) / scale[0]);
        lmax[a] = pcqo_f64_as_i64((bmax[a] - offset[a]) / scale[a]);
    }
    for (int a = 0; a < 3; a++)
        if (lmin[a] > lmax[a])
            return fail(PCQO_ERR_PANIC,
                        "AABB::from_min_max: Minimum position must be <= maximum position!");
    return PCQO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* SparseGrid — query/src/grid_sampling.rs:9-114                                              */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    uint64_t key;
    pcqo_point pt;
    uint8_t used;
} grid_slot;

typedef struct {
    double bmin[3], bmax[3];
    double cell_size;
    uint64_t dims[3];
    uint64_t bits[3];
    grid_slot *slots; /* stands in for HashMap<u64, Point> */
    uint64_t cap, count;
} sparse_grid;

static uint64_t hash64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static void grid_rehash(sparse_grid *g, uint64_t ncap) {
    grid_slot *ns = (grid_slot *)calloc(ncap, sizeof(grid_slot));
    if (!ns) abort();
    for (uint64_t i = 0; i < g->cap; i++) {
        if (!g->slots[i].used) continue;
        uint64_t h = hash64(g->slots[i].key) & (ncap - 1);
        while (ns[h].used) h = (h + 1) & (ncap - 1);
        ns[h] = g->slots[i];
    }
    free(g->slots);
    g->slots = ns;
    g->cap = ncap;
}

/* grid_sampling.rs:18-47 */
static int grid_new(sparse_grid *g, const double bmin[3], const double bmax[3], double cell) {
    memset(g, 0, sizeof *g);
    uint64_t bitsum = 0;
    for (int a = 0; a < 3; a++) {
        g->bmin[a] = bmin[a];
        g->bmax[a] = bmax[a];
        double extent = bmax[a] - bmin[a];             /* :19-23 */
        double ncells = ceil(extent / cell);           /* :24-28 */
        g->bits[a] = pcqo_f64_as_u64(ceil(log2(ncells))); /* :29-31 */
        g->dims[a] = pcqo_f64_as_u64(ncells);          /* :39-43 */
        bitsum += g->bits[a];
    }
    if (bitsum > 64) /* :32-34 (the reference's message has unfilled {} placeholders) */
        return fail(PCQO_ERR_GRID,
                    "Too many cells ({}*{}*{}) in SparseGrid! The number of cells exceeds the "
                    "capacity of a u64 index!");
    g->cell_size = cell;
    g->cap = 1024;
    g->slots = (grid_slot *)calloc(g->cap, sizeof(grid_slot));
    if (!g->slots) abort();
    return PCQO_OK;
}

/* nalgebra 0.23.2 distance_squared [recalled]: (p2 - p1).norm_squared() = a + b + c */
static double dist_sq(const double c[3], double px, double py, double pz) {
    double dx = px - c[0], dy = py - c[1], dz = pz - c[2];
    double a = dx * dx, b = dy * dy, cc = dz * dz;
    return (a + b) + cc;
}

/* Rust release-mode `1u64 << n` masks the shift amount to 6 bits. */
static uint64_t shl64(uint64_t v, uint64_t n) { return v << (n & 63); }

/* grid_sampling.rs:49-105 */
static int grid_insert(sparse_grid *g, const pcqo_point *pt) {
    double p[3] = {pt->x, pt->y, pt->z};
    uint64_t cell[3];
    for (int a = 0; a < 3; a++) {
        double r = (p[a] - g->bmin[a]) * (double)g->dims[a] / (g->bmax[a] - g->bmin[a]); /* :51-56 */
        cell[a] = pcqo_f64_as_u64(r);                                                     /* :58-60 */
    }
    uint64_t mx = shl64(1, g->bits[0]) - 1, my = shl64(1, g->bits[1]) - 1,
             mz = shl64(1, g->bits[2]) - 1;                                               /* :62-64 */
    uint64_t ys = g->bits[0], zs = g->bits[0] + g->bits[1];                               /* :66-67 */
    uint64_t index = (cell[0] & mx) | shl64(cell[1] & my, ys) | shl64(cell[2] & mz, zs);  /* :68-70 */

    uint64_t h = hash64(index) & (g->cap - 1);
    while (g->slots[h].used && g->slots[h].key != index) h = (h + 1) & (g->cap - 1);
    if (!g->slots[h].used) { /* :73-76 */
        g->slots[h].used = 1;
        g->slots[h].key = index;
        g->slots[h].pt = *pt;
        g->count++;
        if (g->count * 2 > g->cap) grid_rehash(g, g->cap * 2);
        return 1;
    }
    /* :77-103 — centre from the NEW point's unmasked cell */
    double centre[3];
    for (int a = 0; a < 3; a++) centre[a] = ((double)cell[a] + 0.5) * g->cell_size + g->bmin[a];
    pcqo_point *cur = &g->slots[h].pt;
    double cur_d = dist_sq(centre, cur->x, cur->y, cur->z);
    double new_d = dist_sq(centre, p[0], p[1], p[2]);
    if (new_d < cur_d) { /* strict: first seen wins ties */
        *cur = *pt;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* collectors — query/src/collect_points.rs                                                   */
/* ------------------------------------------------------------------------------------------ */

struct pcqo_collector {
    int kind;
    uint64_t count;      /* CountCollector::point_count  */
    pcqo_point *buf;     /* BufferCollector::buffer      */
    uint64_t buf_len, buf_cap;
    sparse_grid grid;    /* GridSampledCollector::grid   */
};

pcqo_collector *pcqo_collector_new_count(void) {
    pcqo_collector *c = (pcqo_collector *)calloc(1, sizeof *c);
    c->kind = PCQO_COLLECT_COUNT;
    return c;
}
pcqo_collector *pcqo_collector_new_buffer(void) {
    pcqo_collector *c = (pcqo_collector *)calloc(1, sizeof *c);
    c->kind = PCQO_COLLECT_BUFFER;
    return c;
}
pcqo_collector *pcqo_collector_new_grid(const double bmin[3], const double bmax[3], double cell) {
    pcqo_collector *c = (pcqo_collector *)calloc(1, sizeof *c);
    c->kind = PCQO_COLLECT_GRID;
    if (grid_new(&c->grid, bmin, bmax, cell) != PCQO_OK) {
        free(c);
        return NULL;
    }
    return c;
}
void pcqo_collector_free(pcqo_collector *c) {
    if (!c) return;
    free(c->buf);
    free(c->grid.slots);
    free(c);
}
int pcqo_collector_kind(const pcqo_collector *c) { return c->kind; }

void pcqo_collector_collect_one(pcqo_collector *c, const pcqo_point *p) {
    switch (c->kind) {
    case PCQO_COLLECT_COUNT: /* collect_points.rs:83-85 */
        c->count += 1;
        break;
    case PCQO_COLLECT_BUFFER: /* collect_points.rs:29-31 */
        if (c->buf_len == c->buf_cap) {
            c->buf_cap = c->buf_cap ? c->buf_cap * 2 : 1024;
            c->buf = (pcqo_point *)realloc(c->buf, c->buf_cap * sizeof(pcqo_point));
            if (!c->buf) abort();
        }
        c->buf[c->buf_len++] = *p;
        break;
    case PCQO_COLLECT_GRID: /* collect_points.rs:112-114 */
        grid_insert(&c->grid, p);
        break;
    }
}

uint64_t pcqo_collector_point_count(const pcqo_collector *c) {
    switch (c->kind) {
    case PCQO_COLLECT_COUNT: return c->count;          /* :95-97  */
    case PCQO_COLLECT_BUFFER: return c->buf_len;       /* :41-43  */
    default: return c->grid.count;                     /* :124-126 */
    }
}

int pcqo_collector_has_points(const pcqo_collector *c) { return c->kind != PCQO_COLLECT_COUNT; }

static int cmp_slot_key(const void *a, const void *b) {
    uint64_t ka = ((const grid_slot *)a)->key, kb = ((const grid_slot *)b)->key;
    return ka < kb ? -1 : ka > kb;
}

static grid_slot *grid_sorted(const sparse_grid *g) {
    grid_slot *v = (grid_slot *)malloc((g->count ? g->count : 1) * sizeof(grid_slot));
    uint64_t n = 0;
    for (uint64_t i = 0; i < g->cap; i++)
        if (g->slots[i].used) v[n++] = g->slots[i];
    qsort(v, n, sizeof(grid_slot), cmp_slot_key);
    return v;
}

uint64_t pcqo_collector_points(const pcqo_collector *c, pcqo_point *out, uint64_t cap) {
    if (c->kind == PCQO_COLLECT_COUNT) return 0;
    if (c->kind == PCQO_COLLECT_BUFFER) {
        uint64_t n = c->buf_len < cap ? c->buf_len : cap;
        if (out && n) memcpy(out, c->buf, n * sizeof(pcqo_point));
        return c->buf_len;
    }
    if (out && cap) {
        grid_slot *v = grid_sorted(&c->grid);
        uint64_t n = c->grid.count < cap ? c->grid.count : cap;
        for (uint64_t i = 0; i < n; i++) out[i] = v[i].pt;
        free(v);
    }
    return c->grid.count;
}

uint64_t pcqo_collector_grid_cells(const pcqo_collector *c, uint64_t *out, uint64_t cap) {
    if (c->kind != PCQO_COLLECT_GRID) return 0;
    if (out && cap) {
        grid_slot *v = grid_sorted(&c->grid);
        uint64_t n = c->grid.count < cap ? c->grid.count : cap;
        for (uint64_t i = 0; i < n; i++) out[i] = v[i].key;
        free(v);
    }
    return c->grid.count;
}

int pcqo_collector_grid_params(const pcqo_collector *c, uint64_t dims[3], uint64_t bits[3]) {
    if (c->kind != PCQO_COLLECT_GRID) return PCQO_ERR_ARG;
    for (int a = 0; a < 3; a++) {
        dims[a] = c->grid.dims[a];
        bits[a] = c->grid.bits[a];
    }
    return PCQO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* LAST scans — query/src/search/last.rs                                                      */
/* ------------------------------------------------------------------------------------------ */

static int color_offset_for(uint8_t fmt, uint64_t *off) { /* last.rs:83-88, las.rs:38-45 */
    switch (fmt) {
    case 2: *off = 20; return 1;
    case 3: *off = 28; return 1;
    case 5: *off = 28; return 1;
    default: return 0;
    }
}

/* last.rs:46-166 */
int pcqo_search_last_mem_by_bounds_optimized(const uint8_t *data, size_t len, const double bmin[3],
                                             const double bmax[3], pcqo_collector *c) {
    image im = {data, len};
    pcqo_las_header h;
    int rc = pcqo_parse_las_header(data, len, 0, &h); /* :53-54 (format byte not masked) */
    if (rc) return rc;

    uint8_t fmt = h.point_data_record_format; /* :68 */
    uint64_t cls_in_point;
    if (fmt <= 5) cls_in_point = 15;          /* :69-71 */
    else if (fmt <= 10) cls_in_point = 16;
    else return fail(PCQO_ERR_FORMAT, "Invalid LAS format %u", fmt);
    uint64_t n = h.number_of_points;
    uint64_t cls_block = (uint64_t)h.offset_to_point_data + n * cls_in_point; /* :80-81 */
    uint64_t col_in_point = 0;
    int has_color = color_offset_for(fmt, &col_in_point);                     /* :83-88 */
    uint64_t col_block = (uint64_t)h.offset_to_point_data + n * col_in_point; /* :89-90 */

    if (!pcqo_aabb_intersects(h.min, h.max, bmin, bmax)) return PCQO_OK; /* :92-94 */

    int64_t lmin[3], lmax[3];
    rc = pcqo_box_to_local(bmin, bmax, h.scale, h.offset, lmin, lmax); /* :98-109 */
    if (rc) return rc;

    uint64_t pos_block = h.offset_to_point_data; /* :114 */
    for (uint64_t idx = 0; idx < n; idx++) {     /* :117 */
        uint64_t at = pos_block + idx * 12;      /* :118-121 */
        int32_t xi, yi, zi;
        if (!rd_i32(&im, at, &xi)) return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        int64_t px = xi;
        if (px < lmin[0] || px > lmax[0]) continue; /* :122-125 */
        if (!rd_i32(&im, at + 4, &yi)) return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        int64_t py = yi;
        if (py < lmin[1] || py > lmax[1]) continue; /* :127-130 */
        if (!rd_i32(&im, at + 8, &zi)) return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        int64_t pz = zi;
        if (pz < lmin[2] || pz > lmax[2]) continue; /* :132-135 */

        pcqo_point pt;
        if (!rd_u8(&im, cls_block + idx, &pt.classification)) /* :138-142 */
            return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        pt.r = pt.g = pt.b = 0;
        if (has_color) { /* :145-153 */
            uint64_t co = idx * 6 + col_block;
            if (!rd_u16(&im, co, &pt.r) || !rd_u16(&im, co + 2, &pt.g) || !rd_u16(&im, co + 4, &pt.b))
                return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        }
        pt.x = ((double)px * h.scale[0]) + h.offset[0]; /* :156-160, unfused */
        pt.y = ((double)py * h.scale[1]) + h.offset[1];
        pt.z = ((double)pz * h.scale[2]) + h.offset[2];
        pcqo_collector_collect_one(c, &pt); /* :155 */
    }
    return PCQO_OK;
}

/* last.rs:213-293 */
int pcqo_search_last_mem_by_classification_optimized(const uint8_t *data, size_t len, uint8_t cls,
                                                     pcqo_collector *c) {
    image im = {data, len};
    pcqo_las_header h;
    int rc = pcqo_parse_las_header(data, len, 1, &h); /* :220-223 (format masked &0b1111) */
    if (rc) return rc;
    uint8_t fmt = h.point_data_record_format; /* :225 */
    uint64_t cls_in_point;
    if (fmt <= 5) cls_in_point = 15; /* :226-236 */
    else if (fmt <= 10) cls_in_point = 16;
    else return fail(PCQO_ERR_FORMAT, "Invalid LAS format %u", fmt);
    uint64_t col_in_point = 0;
    int has_color = color_offset_for(fmt, &col_in_point); /* :238-243 */
    uint64_t n = h.number_of_points;
    uint64_t cls_block = cls_in_point * n;                                     /* :245-246 */
    uint64_t col_block = (uint64_t)h.offset_to_point_data + n * col_in_point;  /* :249-250 */

    for (uint64_t idx = 0; idx < n; idx++) { /* :253 */
        uint64_t co = idx + cls_block + (uint64_t)h.offset_to_point_data; /* :254-256 */
        uint8_t k;
        if (!rd_u8(&im, co, &k)) return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        if (k != cls) continue; /* :259-262: whole byte compared */

        uint64_t at = idx * 12 + (uint64_t)h.offset_to_point_data; /* :265 */
        int32_t xi, yi, zi;
        if (!rd_i32(&im, at, &xi) || !rd_i32(&im, at + 4, &yi) || !rd_i32(&im, at + 8, &zi))
            return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        pcqo_point pt;
        pt.r = pt.g = pt.b = 0;
        if (has_color) { /* :272-280 */
            uint64_t cc = idx * 6 + col_block;
            if (!rd_u16(&im, cc, &pt.r) || !rd_u16(&im, cc + 2, &pt.g) || !rd_u16(&im, cc + 4, &pt.b))
                return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        }
        pt.x = ((double)xi * h.scale[0]) + h.offset[0]; /* :283-287 */
        pt.y = ((double)yi * h.scale[1]) + h.offset[1];
        pt.z = ((double)zi * h.scale[2]) + h.offset[2];
        pt.classification = k;
        pcqo_collector_collect_one(c, &pt);
    }
    return PCQO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* LAS (AoS) scans — query/src/search/las.rs                                                  */
/* ------------------------------------------------------------------------------------------ */

/* las.rs:52-148 */
int pcqo_search_las_mem_by_bounds_optimized(const uint8_t *data, size_t len, const double bmin[3],
                                            const double bmax[3], pcqo_collector *c,
                                            int *record_size_printed) {
    image im = {data, len};
    pcqo_las_header h;
    int rc = pcqo_parse_las_header(data, len, 0, &h); /* :59-60 */
    if (rc) return rc;
    if (record_size_printed) *record_size_printed = h.point_data_record_length; /* :73 */
    uint64_t col_off = 0;
    int has_color = color_offset_for(h.point_data_record_format, &col_off); /* :74-80 */

    if (!pcqo_aabb_intersects(h.min, h.max, bmin, bmax)) return PCQO_OK; /* :82-84 */

    int64_t lmin[3], lmax[3];
    rc = pcqo_box_to_local(bmin, bmax, h.scale, h.offset, lmin, lmax); /* :88-99 */
    if (rc) return rc;

    uint64_t n = h.number_of_points;
    for (uint64_t idx = 0; idx < n; idx++) { /* :101 */
        uint64_t at = idx * (uint64_t)h.point_data_record_length + (uint64_t)h.offset_to_point_data;
        int32_t xi, yi, zi;
        if (!rd_i32(&im, at, &xi)) return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        int64_t px = xi;
        if (px < lmin[0] || px > lmax[0]) continue; /* :106-109 */
        if (!rd_i32(&im, at + 4, &yi)) return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        int64_t py = yi;
        if (py < lmin[1] || py > lmax[1]) continue; /* :111-114 */
        if (!rd_i32(&im, at + 8, &zi)) return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        int64_t pz = zi;
        if (pz < lmin[2] || pz > lmax[2]) continue; /* :116-119 */

        pcqo_point pt;
        /* :121-124 — seek(Current(3)) after 12 bytes: the class byte is ALWAYS read at +15 here */
        if (!rd_u8(&im, at + 15, &pt.classification))
            return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        pt.r = pt.g = pt.b = 0;
        if (has_color) { /* :127-135: (+16) + (color_offset - 16) */
            uint64_t cc = at + col_off;
            if (!rd_u16(&im, cc, &pt.r) || !rd_u16(&im, cc + 2, &pt.g) || !rd_u16(&im, cc + 4, &pt.b))
                return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        }
        pt.x = ((double)px * h.scale[0]) + h.offset[0]; /* :138-142 */
        pt.y = ((double)py * h.scale[1]) + h.offset[1];
        pt.z = ((double)pz * h.scale[2]) + h.offset[2];
        pcqo_collector_collect_one(c, &pt);
    }
    return PCQO_OK;
}

/* las.rs:192-261 */
int pcqo_search_las_mem_by_classification_optimized(const uint8_t *data, size_t len, uint8_t cls,
                                                    pcqo_collector *c) {
    image im = {data, len};
    pcqo_las_header h;
    int rc = pcqo_parse_las_header(data, len, 0, &h); /* :199-200 */
    if (rc) return rc;
    uint8_t fmt = h.point_data_record_format; /* raw, unmasked: :202 */
    uint64_t cls_in_point;
    if (fmt <= 5) cls_in_point = 15;
    else if (fmt <= 10) cls_in_point = 16;
    else return fail(PCQO_ERR_FORMAT, "Invalid LAS format %u", fmt);
    uint64_t col_off = 0;
    int has_color = color_offset_for(fmt, &col_off); /* :214-219 */

    uint64_t n = h.number_of_points;
    for (uint64_t idx = 0; idx < n; idx++) { /* :221 */
        uint64_t at = idx * (uint64_t)h.point_data_record_length + (uint64_t)h.offset_to_point_data;
        uint8_t k;
        if (!rd_u8(&im, at + cls_in_point, &k)) /* :224-228 */
            return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        if (k != cls) continue;
        int32_t xi, yi, zi; /* :234-237 */
        if (!rd_i32(&im, at, &xi) || !rd_i32(&im, at + 4, &yi) || !rd_i32(&im, at + 8, &zi))
            return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        pcqo_point pt;
        pt.r = pt.g = pt.b = 0;
        if (has_color) { /* :240-248 */
            uint64_t cc = at + col_off;
            if (!rd_u16(&im, cc, &pt.r) || !rd_u16(&im, cc + 2, &pt.g) || !rd_u16(&im, cc + 4, &pt.b))
                return fail(PCQO_ERR_EOF, "failed to fill whole buffer");
        }
        pt.x = ((double)xi * h.scale[0]) + h.offset[0]; /* :251-255 */
        pt.y = ((double)yi * h.scale[1]) + h.offset[1];
        pt.z = ((double)zi * h.scale[2]) + h.offset[2];
        pt.classification = k;
        pcqo_collector_collect_one(c, &pt);
    }
    return PCQO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* file level: open + mmap (last.rs:27-34); dispatch (searcher.rs:43-152)                     */
/* ------------------------------------------------------------------------------------------ */

static const char *path_extension(const char *path) { /* Path::extension() */
    const char *base = strrchr(path, '/');
    base = base ? base + 1 : path;
    const char *dot = strrchr(base, '.');
    if (!dot || dot == base) return NULL;
    return dot + 1;
}

int pcqo_search_file(const char *path, int query_kind, const double bmin[3], const double bmax[3],
                     uint8_t cls, pcqo_collector *c, int *record_size_printed) {
    const char *ext = path_extension(path);
    if (!ext) return fail(PCQO_ERR_EXTENSION, "Invalid extension on file %s", path);
    int is_las = strcmp(ext, "las") == 0, is_last = strcmp(ext, "last") == 0, is_lazer = strcmp(ext, "lazer") == 0;
    if (!is_las && !is_last && !is_lazer)
        return fail(PCQO_ERR_EXTENSION, "Unsupported file extension in file %s", path);

    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(PCQO_ERR_IO, "%s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return fail(PCQO_ERR_IO, "%s: %s", path, strerror(errno));
    }
    const uint8_t *p = NULL;
    size_t len = (size_t)st.st_size;
    if (len > 0) {
        p = (const uint8_t *)mmap(NULL, len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) {
            close(fd);
            return fail(PCQO_ERR_IO, "%s: mmap: %s", path, strerror(errno));
        }
    }
    close(fd);
    int rc;
    if (is_lazer) /* searcher.rs:83, :144 — one implementation for Regular and Optimized */
        rc = query_kind == PCQO_QUERY_BOUNDS ? pcqo_search_lazer_mem_by_bounds(p, len, bmin, bmax, c)
                                             : pcqo_search_lazer_mem_by_classification(p, len, cls, c);
    else if (is_last)
        rc = query_kind == PCQO_QUERY_BOUNDS
                 ? pcqo_search_last_mem_by_bounds_optimized(p, len, bmin, bmax, c)
                 : pcqo_search_last_mem_by_classification_optimized(p, len, cls, c);
    else
        rc = query_kind == PCQO_QUERY_BOUNDS
                 ? pcqo_search_las_mem_by_bounds_optimized(p, len, bmin, bmax, c, record_size_printed)
                 : pcqo_search_las_mem_by_classification_optimized(p, len, cls, c);
    if (p) munmap((void *)p, len);
    return rc;
}

int pcqo_lazer_file_bounds(const char *path, double mn[3], double mx[3]) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(PCQO_ERR_IO, "%s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size == 0) {
        close(fd);
        return fail(PCQO_ERR_HEADER, "%s: failed to fill whole buffer", path);
    }
    const uint8_t *p = (const uint8_t *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(PCQO_ERR_IO, "%s: mmap failed", path);
    int rc = pcqo_lazer_mem_bounds(p, (size_t)st.st_size, mn, mx);
    munmap((void *)p, (size_t)st.st_size);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* run_search_parallel with CountCollectors — main.rs:146-183                                 */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    const uint8_t *const *images;
    const size_t *lens;
    size_t nfiles;
    int query_kind;
    const double *bmin, *bmax;
    uint8_t cls;
    uint64_t *counts;
    int *rcs;
    size_t next;
    pthread_mutex_t mu;
} par_job;

static void *par_worker(void *arg) {
    par_job *j = (par_job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t i = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (i >= j->nfiles) break;
        pcqo_collector *c = pcqo_collector_new_count(); /* main.rs:156 */
        /* the baseline scans LAST images only (the north-star format) */
        j->rcs[i] = j->query_kind == PCQO_QUERY_BOUNDS
                        ? pcqo_search_last_mem_by_bounds_optimized(j->images[i], j->lens[i], j->bmin,
                                                                   j->bmax, c)
                        : pcqo_search_last_mem_by_classification_optimized(j->images[i], j->lens[i],
                                                                           j->cls, c);
        j->counts[i] = pcqo_collector_point_count(c);
        pcqo_collector_free(c);
    }
    return NULL;
}

int pcqo_count_files_parallel(const uint8_t *const *images, const size_t *lens, size_t nfiles,
                              int query_kind, const double bmin[3], const double bmax[3],
                              uint8_t cls, int threads, uint64_t *total) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > nfiles) threads = (int)nfiles; /* README.md:12 */
    par_job j = {images, lens, nfiles, query_kind, bmin, bmax, cls, NULL, NULL, 0,
                 PTHREAD_MUTEX_INITIALIZER};
    j.counts = (uint64_t *)calloc(nfiles ? nfiles : 1, sizeof(uint64_t));
    j.rcs = (int *)calloc(nfiles ? nfiles : 1, sizeof(int));
    pthread_t *th = (pthread_t *)calloc((size_t)(threads ? threads : 1), sizeof(pthread_t));
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, par_worker, &j);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    int rc = PCQO_OK;
    uint64_t sum = 0;
    for (size_t i = 0; i < nfiles; i++) { /* main.rs:161-176: first Err aborts, else sum */
        if (j.rcs[i] && !rc) rc = j.rcs[i];
        sum += j.counts[i];
    }
    *total = sum;
    free(j.counts);
    free(j.rcs);
    free(th);
    return rc;
}
