/*
 * lazer_oracle.c — CPU restatement of the reference's LAZER searches.
 *
 * TEST INFRASTRUCTURE ONLY (see pcq_oracle.h).
 *
 * Restates, line by line and deliberately in the reference's own shape (a streaming LZ4 reader per
 * attribute, a per-point read loop, a growing point buffer):
 *   readers/src/lazer_reader.rs:58-127    LAZERSource::from
 *   readers/src/lazer_reader.rs:136-265   move_decoders_to_point_in_block
 *   readers/src/lazer_reader.rs:514-764   PointReader::read_into
 *   query/src/search/lazer.rs:34-116      search_lazer_file_by_{bounds,classification}
 *
 * Third-party pieces that are not in /root/reference:
 *   lz4 1.23.2 (Cargo.lock) — `Decoder<R>: Read`, a thin loop over liblz4's LZ4F_decompress.  Restated
 *   from the published LZ4 Frame format (v1.6.x) and LZ4 Block format documents.  PINNED: the image
 *   ships liblz4.so.1.9.3, the same library family lz4-sys wraps; tests/test_lz4_lazer.py checks this
 *   decoder (and the product's) against frames produced by the real LZ4F_compressFrame with every flag
 *   combination, and tests/golden/ holds such frames for hosts without the library.
 *
 * Also here, for the tests only: an LZ4 frame *writer* and a LAST -> LAZER converter (the reference
 * has no LAZER writer in this repository; the block layout is the one lazer_reader.rs reads).
 */
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcq_oracle.h"

int pcqo_fail_msg(int code, const char *msg); /* pcq_oracle.c */

/* ------------------------------------------------------------------------------------------ */
/* xxHash32, seed 0                                                                            */
/* ------------------------------------------------------------------------------------------ */
#define XP1 0x9E3779B1u
#define XP2 0x85EBCA77u
#define XP3 0xC2B2AE3Du
#define XP4 0x27D4EB2Fu
#define XP5 0x165667B1u
static uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
static uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }

uint32_t pcqo_xxh32(const uint8_t *d, size_t n) {
    size_t i = 0;
    uint32_t acc;
    if (n >= 16) {
        uint32_t a[4] = {XP1 + XP2, XP2, 0, 0u - XP1};
        for (; i + 16 <= n; i += 16)
            for (int k = 0; k < 4; k++) a[k] = rol(a[k] + le32(d + i + 4 * k) * XP2, 13) * XP1;
        acc = rol(a[0], 1) + rol(a[1], 7) + rol(a[2], 12) + rol(a[3], 18);
    } else {
        acc = XP5;
    }
    acc += (uint32_t)n;
    for (; i + 4 <= n; i += 4) acc = rol(acc + le32(d + i) * XP3, 17) * XP4;
    for (; i < n; i++) acc = rol(acc + d[i] * XP5, 11) * XP1;
    acc ^= acc >> 15;
    acc *= XP2;
    acc ^= acc >> 13;
    acc *= XP3;
    acc ^= acc >> 16;
    return acc;
}

/* ------------------------------------------------------------------------------------------ */
/* lz4::Decoder — a streaming reader over one LZ4 frame                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *src;
    size_t n, ip;          /* compressed input and the read position in it                    */
    uint8_t *content;      /* everything inflated so far (history for linked blocks)          */
    size_t have, cap, rd;  /* inflated bytes, capacity, bytes already handed out              */
    int stage;             /* 0 header not parsed, 1 in blocks, 2 frame finished, <0 error     */
    int dry;               /* an inflated block is buffered in liblz4 but no input byte is left  */
    int last_stored;       /* the newest block was a stored one                                  */
    int stored_open;       /* the newest block is a stored one whose checksum has not been looked at */
    const uint8_t *stored_at;
    size_t stored_len;
    int independent, block_sum, content_sum, has_size;
    uint64_t content_size;
    size_t max_block;
} lz4_reader;

static void lz4r_init(lz4_reader *r, const uint8_t *src, size_t n) {
    memset(r, 0, sizeof *r);
    r->src = src;
    r->n = n;
}
static void lz4r_free(lz4_reader *r) { free(r->content); }
static int lz4r_room(lz4_reader *r, size_t extra) {
    if (r->have + extra <= r->cap) return 0;
    size_t nc = r->cap ? r->cap : 1 << 16;
    while (nc < r->have + extra) nc *= 2;
    uint8_t *p = (uint8_t *)realloc(r->content, nc);
    if (!p) return -1;
    r->content = p;
    r->cap = nc;
    return 0;
}

/* frame header: magic | FLG | BD | [size] | [dict] | HC */
static int lz4r_header(lz4_reader *r) {
    if (r->n - r->ip < 4) return PCQO_ERR_EOF;
    uint32_t magic = le32(r->src + r->ip);
    /* a skippable frame ends the stream for lz4::Decoder: LZ4F_decompress returns 0 after it, the
     * crate sets `next = 0`, every later read() returns 0 -> read_exact: UnexpectedEof */
    if ((magic >> 4) == (0x184D2A50u >> 4)) return PCQO_ERR_EOF;
    if (magic != 0x184D2204u) return PCQO_ERR_HEADER;
    size_t d0 = r->ip + 4, p = d0;
    if (r->n - p < 3) return PCQO_ERR_EOF;
    uint8_t flg = r->src[p], bd = r->src[p + 1];
    p += 2;
    if (((flg >> 6) & 3) != 1) return PCQO_ERR_HEADER;
    if ((flg >> 1) & 1) return PCQO_ERR_HEADER;
    if ((bd >> 7) || (bd & 15)) return PCQO_ERR_HEADER;
    int id = (bd >> 4) & 7;
    if (id < 4) return PCQO_ERR_HEADER;
    static const size_t sizes[4] = {64u << 10, 256u << 10, 1u << 20, 4u << 20};
    r->max_block = sizes[id - 4];
    r->independent = (flg >> 5) & 1;
    r->block_sum = (flg >> 4) & 1;
    r->has_size = (flg >> 3) & 1;
    r->content_sum = (flg >> 2) & 1;
    if (r->has_size) {
        if (r->n - p < 8) return PCQO_ERR_EOF;
        r->content_size = le64(r->src + p);
        p += 8;
    }
    if (flg & 1) {
        if (r->n - p < 4) return PCQO_ERR_EOF;
        p += 4;
    }
    if (r->n - p < 1) return PCQO_ERR_EOF;
    if (r->src[p] != (uint8_t)(pcqo_xxh32(r->src + d0, p - d0) >> 8)) return PCQO_ERR_HEADER;
    r->ip = p + 1;
    r->stage = 1;
    return 0;
}

/* One compressed block -> appended to content.  liblz4's LZ4_decompress_safe (lz4.c,
 * LZ4_decompress_generic: endOnInput, full decode, output capacity = the frame's maximum block size, as
 * LZ4F uses it for its tmpOut buffer) restated with its end-of-block parsing restrictions, so that a
 * damaged block is refused exactly when the real library refuses it:
 *   MFLIMIT 12, LASTLITERALS 5, RUN_MASK 15; `ip` / `op` count from the start of the block. */
static int lz4r_sequences(lz4_reader *r, const uint8_t *b, size_t bn, size_t floor) {
    long long iend = (long long)bn, oend = (long long)r->max_block, ip = 0, op = 0;
    if (bn == 0) return PCQO_ERR_HEADER;
    for (;;) {
        unsigned tok = b[ip++];
        long long len = tok >> 4;
        if (len == 15) { /* read_variable_length(&ip, iend - RUN_MASK, 1, 1): only the initial check is fatal */
            if (ip >= iend - 15) return PCQO_ERR_HEADER;
            for (;;) {
                unsigned x = b[ip++];
                len += x;
                if (ip >= iend - 15) break; /* loop_error: ignored by the caller */
                if (x != 255) break;
            }
        }
        long long cpy = op + len;
        if (cpy > oend - 12 || ip + len > iend - (2 + 1 + 5)) { /* parsing restriction: must be the last sequence */
            if (ip + len != iend || cpy > oend) return PCQO_ERR_HEADER;
            if (lz4r_room(r, (size_t)len)) return PCQO_ERR_ARG;
            memcpy(r->content + r->have, b + ip, (size_t)len);
            r->have += (size_t)len;
            return 0;
        }
        if (lz4r_room(r, (size_t)len)) return PCQO_ERR_ARG;
        memcpy(r->content + r->have, b + ip, (size_t)len);
        r->have += (size_t)len;
        ip += len;
        op = cpy;
        size_t dist = b[ip] | (b[ip + 1] << 8);
        ip += 2;
        long long ml = tok & 15;
        if (ml == 15) /* read_variable_length(&ip, iend - LASTLITERALS + 1, 1, 0): any error is fatal */
            for (;;) {
                unsigned x = b[ip++];
                ml += x;
                if (ip >= iend - 5 + 1) return PCQO_ERR_HEADER;
                if (x != 255) break;
            }
        ml += 4;
        if (dist > r->have - floor) return PCQO_ERR_HEADER; /* match + dictSize < lowPrefix */
        if (dist == 0) return PCQO_ERR_HEADER;              /* liblz4 1.9.3 would copy stale buffer bytes: undefined result */
        if (op + ml > oend - 5) return PCQO_ERR_HEADER;     /* last LASTLITERALS bytes must be literals */
        if (lz4r_room(r, (size_t)ml)) return PCQO_ERR_ARG;
        for (long long k = 0; k < ml; k++, r->have++) r->content[r->have] = r->content[r->have - dist];
        op += ml;
    }
}

/* dstage_getBlockChecksum behind a stored block: reached once all of its bytes have been handed out */
static int lz4r_close_stored(lz4_reader *r) {
    if (!r->stored_open) return 0;
    if (r->stored_open == 2) return PCQO_ERR_EOF; /* the block was cut short */
    if (r->block_sum) {
        if (r->n - r->ip < 4) return PCQO_ERR_EOF;
        if (le32(r->src + r->ip) != pcqo_xxh32(r->stored_at, r->stored_len)) return PCQO_ERR_HEADER;
        r->ip += 4;
    }
    r->stored_open = 0;
    return 0;
}

static int lz4r_next_block(lz4_reader *r) {
    int pending = lz4r_close_stored(r);
    if (pending) return pending;
    if (r->n - r->ip < 4) return PCQO_ERR_EOF;
    uint32_t word = le32(r->src + r->ip);
    r->ip += 4;
    if (word == 0) { /* EndMark: LZ4F_decompress verifies size and checksum here */
        if (r->has_size && r->content_size != r->have) return PCQO_ERR_HEADER;
        if (r->content_sum) {
            if (r->n - r->ip < 4) return PCQO_ERR_EOF;
            if (le32(r->src + r->ip) != pcqo_xxh32(r->content, r->have)) return PCQO_ERR_HEADER;
            r->ip += 4;
        }
        r->stage = 2;
        return 0;
    }
    size_t len = word & 0x7FFFFFFFu;
    if (len > r->max_block) return PCQO_ERR_HEADER;
    const uint8_t *b = r->src + r->ip;
    r->last_stored = (int)(word >> 31);
    if (word >> 31) { /* stored: LZ4F's copyDirect passes the bytes through as they arrive; the block checksum
                       * comes after the data and is checked once the data is out */
        size_t avail = r->n - r->ip < len ? r->n - r->ip : len;
        if (lz4r_room(r, avail)) return PCQO_ERR_ARG;
        memcpy(r->content + r->have, b, avail);
        r->have += avail;
        r->ip += avail;
        r->stored_open = avail < len ? 2 : 1;
        r->stored_at = b;
        r->stored_len = len;
        return 0;
    }
    if (r->n - r->ip < len + (r->block_sum ? 4u : 0u)) return PCQO_ERR_EOF;
    if (r->block_sum && le32(b + len) != pcqo_xxh32(b, len)) return PCQO_ERR_HEADER;
    r->ip += len + (r->block_sum ? 4 : 0);
    int e = lz4r_sequences(r, b, len, r->independent ? r->have : 0);
    /* lz4::Decoder::read only calls LZ4F_decompress while it has unread input or can fetch more
     * (decoder.rs: `if self.pos >= self.len { self.len = self.r.read(..)?; if self.len == 0 { break; } }`).
     * A compressed block sits in liblz4's tmpOut and is flushed a few bytes per call, without using up
     * input; when the input ends exactly behind the block, the call that inflated it is the last one
     * that delivers anything. */
    if (!e && r->ip == r->n) r->dry = 1;
    return e;
}

/* Read::read_exact on the Decoder: 0, or the error (a short read is UnexpectedEof). */
static int lz4r_read_exact(lz4_reader *r, void *dst, size_t want) {
    if (want == 0) return 0;
    if (r->stage < 0) return r->stage;
    if (r->dry) return PCQO_ERR_EOF;
    if (r->stage == 0) {
        int e = lz4r_header(r);
        if (e) return r->stage = e;
    }
    while (r->have - r->rd < want) {
        if (r->stage == 2) return PCQO_ERR_EOF; /* frame over: read() returns 0 from now on */
        int e = lz4r_next_block(r);
        if (e) return r->stage = e;
    }
    memcpy(dst, r->content + r->rd, want);
    r->rd += want;
    /* (the checksum of a stored block is looked at by the NEXT read: with the usual full-size stored
     * block the crate's 32 KiB input buffer ends exactly at the end of the block data) */
    /* The LZ4F_decompress call that flushes the last byte of a block moves on to the next block header
     * if the input holds it (the crate feeds `block + 4` bytes at a time): an EndMark is followed by the
     * content-size check (dstage_getSuffix) — the content checksum needs 4 more input bytes, which only
     * a further read would fetch — and a header announcing more than the maximum block size is an error. */
    /* (behind a stored block the crate's input buffering decides what else liblz4 sees: not modelled) */
    if (r->rd == r->have && r->stage == 1 && !r->dry && !r->last_stored && r->n - r->ip >= 4) {
        uint32_t word = le32(r->src + r->ip);
        if (word == 0) {
            if (r->has_size && r->content_size != r->have) return r->stage = PCQO_ERR_HEADER;
        } else if ((word & 0x7FFFFFFFu) > r->max_block) {
            return r->stage = PCQO_ERR_HEADER;
        }
    }
    return 0;
}

/* test hook: first `need` bytes of a frame; returns the number of bytes written or a negative error */
int64_t pcqo_lz4f_decode(const uint8_t *src, size_t n, uint8_t *out, size_t need, size_t unit) {
    lz4_reader r;
    lz4r_init(&r, src, n);
    int e = 0;
    if (unit == 0) unit = need;
    for (size_t at = 0; at < need && !e; at += unit) e = lz4r_read_exact(&r, out + at, need - at < unit ? need - at : unit);
    lz4r_free(&r);
    return e ? e : (int64_t)need;
}

/* ------------------------------------------------------------------------------------------ */
/* LAZERSource                                                                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *data;
    size_t len;
    pcqo_las_header h;
    uint64_t block_size, num_blocks;
    uint64_t *block_offsets, *block_byte_sizes;
    size_t nattr;
    int has_colors;
    size_t current_point_index;
    lz4_reader pos, cls, col; /* decoders_per_attribute (intensity is created but never read) */
    int decoders_live;
} lazer_source;

static int fmt_color(unsigned f) { return f == 2 || f == 3 || f == 5 || f == 7 || f == 8 || f == 10; }
static int fmt_gps(unsigned f) { return f == 1 || (f >= 3 && f <= 10); }
static int fmt_wave(unsigned f) { return f == 4 || f == 5 || f == 9 || f == 10; }
static int fmt_nir(unsigned f) { return f == 8 || f == 10; }

static void lazer_drop_decoders(lazer_source *s) {
    if (s->decoders_live) {
        lz4r_free(&s->pos);
        lz4r_free(&s->cls);
        lz4r_free(&s->col);
        s->decoders_live = 0;
    }
}
static void lazer_close(lazer_source *s) {
    lazer_drop_decoders(s);
    free(s->block_offsets);
    free(s->block_byte_sizes);
}

/* lazer_reader.rs:136-265 with point_in_block == 0 */
static int lazer_move_to_block(lazer_source *s, uint64_t b) {
    uint64_t at = s->block_offsets[b], table = 8 * (uint64_t)s->nattr;
    if (at > s->len || s->len - at < table) return pcqo_fail_msg(PCQO_ERR_EOF, "failed to fill whole buffer"); /* :148-150 */
    uint64_t off[12];
    for (size_t k = 0; k < s->nattr; k++) off[k] = le64(s->data + at + 8 * k);
    uint64_t bytes = s->block_byte_sizes[b];
    if (bytes < table) return pcqo_fail_msg(PCQO_ERR_PANIC, "attempt to subtract with overflow"); /* :161 */
    uint64_t blob = bytes - table;
    if (s->len - at - table < blob) return pcqo_fail_msg(PCQO_ERR_EOF, "failed to fill whole buffer"); /* :169-170 */
    const uint8_t *cache = s->data + at + table;
    /* :176-262 — slices by unchecked pointer arithmetic in the reference; out-of-block offsets are UB
     * there, an error here */
    uint64_t a[3] = {off[0], off[3], s->has_colors ? off[8] : off[0]};
    uint64_t e[3] = {off[1], off[4], s->has_colors ? (s->nattr > 9 ? off[9] : at + bytes) : off[0]};
    lz4_reader *rd[3] = {&s->pos, &s->cls, &s->col};
    lazer_drop_decoders(s);
    for (int k = 0; k < 3; k++) {
        if (a[k] < off[0] || e[k] < a[k] || a[k] - off[0] > blob)
            return pcqo_fail_msg(PCQO_ERR_HEADER, "LAZER attribute offsets outside the block");
        uint64_t lo = a[k] - off[0], hi = e[k] - off[0];
        if (hi > blob) hi = blob;
        lz4r_init(rd[k], cache + lo, (size_t)(hi - lo));
    }
    s->decoders_live = 1;
    return 0;
}

/* lazer_reader.rs:58-127 */
static int lazer_from(lazer_source *s, const uint8_t *data, size_t len) {
    memset(s, 0, sizeof *s);
    s->data = data;
    s->len = len;
    int rc = pcqo_parse_las_header(data, len, 0, &s->h); /* :59-60 */
    if (rc) return rc;
    uint64_t otp = s->h.offset_to_point_data, n = s->h.number_of_points;
    if (otp > len || len - otp < 8) return pcqo_fail_msg(PCQO_ERR_EOF, "failed to fill whole buffer"); /* :66 */
    s->block_size = le64(data + otp);
    if (s->block_size == 0) return pcqo_fail_msg(PCQO_ERR_PANIC, "attempt to divide by zero"); /* :67 */
    s->num_blocks = n / s->block_size + (n % s->block_size != 0);
    if ((len - otp - 8) / 8 < s->num_blocks) return pcqo_fail_msg(PCQO_ERR_EOF, "failed to fill whole buffer"); /* :70-72 */
    s->block_offsets = (uint64_t *)calloc(s->num_blocks + 1, 8);
    s->block_byte_sizes = (uint64_t *)calloc(s->num_blocks + 1, 8);
    for (uint64_t b = 0; b < s->num_blocks; b++) s->block_offsets[b] = le64(data + otp + 8 + 8 * b);
    for (uint64_t b = 0; b < s->num_blocks; b++) { /* :79-87 */
        uint64_t end = b == s->num_blocks - 1 ? (uint64_t)len : s->block_offsets[b + 1];
        if (end < s->block_offsets[b]) return pcqo_fail_msg(PCQO_ERR_PANIC, "attempt to subtract with overflow");
        s->block_byte_sizes[b] = end - s->block_offsets[b];
    }
    unsigned f = s->h.point_data_record_format;
    s->has_colors = fmt_color(f); /* :93-105 */
    s->nattr = 8 + fmt_color(f) + fmt_gps(f) + fmt_wave(f) + fmt_nir(f);
    if (s->num_blocks == 0) return pcqo_fail_msg(PCQO_ERR_PANIC, "index out of bounds: the len is 0 but the index is 0"); /* :123 -> :143 */
    return lazer_move_to_block(s, 0); /* :123 */
}

/* PerAttributeVecPointStorage with Point::layout() */
typedef struct {
    double *pos;     /* 3 per point */
    uint16_t *col;   /* 3 per point */
    uint8_t *cls;
    size_t len, cap;
} point_buffer;
static int pb_reserve(point_buffer *pb, size_t extra) {
    if (pb->len + extra <= pb->cap) return 0;
    size_t nc = pb->cap ? pb->cap : 1024;
    while (nc < pb->len + extra) nc *= 2;
    double *p = (double *)realloc(pb->pos, nc * 24);
    if (!p) return -1;
    pb->pos = p;
    uint16_t *c = (uint16_t *)realloc(pb->col, nc * 6);
    if (!c) return -1;
    pb->col = c;
    uint8_t *k = (uint8_t *)realloc(pb->cls, nc);
    if (!k) return -1;
    pb->cls = k;
    pb->cap = nc;
    return 0;
}
static void pb_free(point_buffer *pb) {
    free(pb->pos);
    free(pb->col);
    free(pb->cls);
}

/* lazer_reader.rs:514-764: appends `count` points (all from the current block when called with
 * chunk == block, as the searches do) to the buffer */
static int lazer_read_into(lazer_source *s, point_buffer *pb, size_t count) {
    size_t left = (size_t)s->h.number_of_points - s->current_point_index;
    uint64_t todo = count < left ? count : left; /* :525-531 */
    if (todo == 0) return 0;
    uint64_t B = s->block_size;
    uint64_t first_block = s->current_point_index / B, last_block = (s->current_point_index + todo) / B; /* :533-535 */
    uint64_t first_point = s->current_point_index, last_point = first_point + todo;
    for (uint64_t b = first_block; b <= last_block; b++) { /* :573 */
        uint64_t bstart = b * B;
        uint64_t in0 = first_point < bstart ? 0 : first_point - bstart;
        uint64_t in1 = last_point - bstart < B ? last_point - bstart : B;
        uint64_t cnt = in1 - in0; /* :583 */
        if (pb_reserve(pb, cnt)) return pcqo_fail_msg(PCQO_ERR_ARG, "out of memory");
        size_t base = pb->len;
        for (uint64_t i = 0; i < cnt; i++) { /* :597-624 */
            uint8_t raw[12];
            int e = lz4r_read_exact(&s->pos, raw, 12);
            if (e) return pcqo_fail_msg(e, "positions: LZ4 read failed");
            int32_t x = (int32_t)le32(raw), y = (int32_t)le32(raw + 4), z = (int32_t)le32(raw + 8);
            pb->pos[(base + i) * 3 + 0] = s->h.offset[0] + s->h.scale[0] * (double)x;
            pb->pos[(base + i) * 3 + 1] = s->h.offset[1] + s->h.scale[1] * (double)y;
            pb->pos[(base + i) * 3 + 2] = s->h.offset[2] + s->h.scale[2] * (double)z;
        }
        /* :627-655 intensity: Point::layout() has no INTENSITY, the decoder is never read */
        for (uint64_t i = 0; i < cnt; i++) { /* :664-680 */
            int e = lz4r_read_exact(&s->cls, &pb->cls[base + i], 1);
            if (e) return pcqo_fail_msg(e, "classifications: LZ4 read failed");
        }
        for (uint64_t i = 0; i < cnt; i++) { /* :683-716; no colour decoder -> zeros */
            uint8_t raw[6] = {0, 0, 0, 0, 0, 0};
            if (s->has_colors) {
                int e = lz4r_read_exact(&s->col, raw, 6);
                if (e) return pcqo_fail_msg(e, "colors: LZ4 read failed");
            }
            for (int k = 0; k < 3; k++) pb->col[(base + i) * 3 + k] = (uint16_t)(raw[2 * k] | (raw[2 * k + 1] << 8));
        }
        pb->len += cnt; /* :718-729 push */
        if (b != s->num_blocks - 1 && in1 == B) { /* :735-737 */
            int e = lazer_move_to_block(s, b + 1);
            if (e) return e;
        }
        s->current_point_index += cnt;
    }
    return 0;
}

static void pb_get_point(const point_buffer *pb, size_t i, pcqo_point *p) {
    p->x = pb->pos[3 * i];
    p->y = pb->pos[3 * i + 1];
    p->z = pb->pos[3 * i + 2];
    p->r = pb->col[3 * i];
    p->g = pb->col[3 * i + 1];
    p->b = pb->col[3 * i + 2];
    p->classification = pb->cls[i];
}

/* pasture AABB::contains: rejects on `p < min || p > max` per axis [recalled, pasture-core 0.1.0] */
static int aabb_contains(const double mn[3], const double mx[3], const double *p) {
    if (p[0] < mn[0] || p[1] < mn[1] || p[2] < mn[2]) return 0;
    if (p[0] > mx[0] || p[1] > mx[1] || p[2] > mx[2]) return 0;
    return 1;
}

/* lazer.rs:34-78 */
int pcqo_search_lazer_mem_by_bounds(const uint8_t *data, size_t len, const double bmin[3], const double bmax[3],
                                    pcqo_collector *c) {
    lazer_source s;
    int rc = lazer_from(&s, data, len); /* :41 */
    if (rc) {
        lazer_close(&s);
        return rc;
    }
    size_t n = (size_t)s.h.number_of_points;
    if (!pcqo_aabb_intersects(s.h.min, s.h.max, bmin, bmax)) { /* :47-53 */
        lazer_close(&s);
        return PCQO_OK;
    }
    size_t chunk = (size_t)s.block_size; /* :56 */
    point_buffer pb = {0};
    size_t chunks = (n + chunk - 1) / chunk; /* :59 */
    for (size_t idx = 0; idx < chunks && !rc; idx++) {
        size_t in_chunk = n - idx * chunk < chunk ? n - idx * chunk : chunk; /* :61 */
        rc = lazer_read_into(&s, &pb, in_chunk);                               /* :62 */
        if (rc) break;
        for (size_t i = 0; i < in_chunk; i++) /* :64-73 */
            if (aabb_contains(bmin, bmax, &pb.pos[3 * i])) {
                pcqo_point p;
                pb_get_point(&pb, i, &p);
                pcqo_collector_collect_one(c, &p);
            }
        pb.len = 0; /* :75 clear */
    }
    pb_free(&pb);
    lazer_close(&s);
    return rc;
}

/* lazer.rs:80-116 — no clear(): the buffer keeps growing and index i keeps meaning "point i of the
 * first chunk" */
int pcqo_search_lazer_mem_by_classification(const uint8_t *data, size_t len, uint8_t cls, pcqo_collector *c) {
    lazer_source s;
    int rc = lazer_from(&s, data, len); /* :87 */
    if (rc) {
        lazer_close(&s);
        return rc;
    }
    size_t n = (size_t)s.h.number_of_points;
    size_t chunk = (size_t)s.block_size; /* :95 */
    point_buffer pb = {0};
    size_t chunks = (n + chunk - 1) / chunk; /* :98 */
    for (size_t idx = 0; idx < chunks && !rc; idx++) {
        size_t in_chunk = n - idx * chunk < chunk ? n - idx * chunk : chunk; /* :100 */
        rc = lazer_read_into(&s, &pb, in_chunk);                               /* :101 */
        if (rc) break;
        for (size_t i = 0; i < in_chunk; i++) /* :103-112 */
            if (pb.cls[i] == cls) {
                pcqo_point p;
                pb_get_point(&pb, i, &p);
                pcqo_collector_collect_one(c, &p);
            }
    }
    pb_free(&pb);
    lazer_close(&s);
    return rc;
}

/* main.rs:102-107,111 — LAZERSource::from(file) then metadata bounds */
int pcqo_lazer_mem_bounds(const uint8_t *data, size_t len, double mn[3], double mx[3]) {
    lazer_source s;
    int rc = lazer_from(&s, data, len);
    if (!rc)
        for (int a = 0; a < 3; a++) mn[a] = s.h.min[a], mx[a] = s.h.max[a];
    lazer_close(&s);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Test-side writers: LZ4 frames and LAZER images                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint8_t *p;
    size_t n, cap;
    int oom;
} sink;
static void put(sink *s, const void *d, size_t n) {
    if (s->n + n > s->cap) {
        size_t nc = s->cap ? s->cap : 4096;
        while (nc < s->n + n) nc *= 2;
        uint8_t *q = (uint8_t *)realloc(s->p, nc);
        if (!q) {
            s->oom = 1;
            return;
        }
        s->p = q;
        s->cap = nc;
    }
    memcpy(s->p + s->n, d, n);
    s->n += n;
}
static void put32(sink *s, uint32_t v) {
    uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
    put(s, b, 4);
}
static void put64(sink *s, uint64_t v) {
    put32(s, (uint32_t)v);
    put32(s, (uint32_t)(v >> 32));
}
static void put_len(sink *s, size_t v) { /* the 255-run length continuation */
    while (v >= 255) {
        uint8_t b = 255;
        put(s, &b, 1);
        v -= 255;
    }
    uint8_t b = (uint8_t)v;
    put(s, &b, 1);
}

#define HASH_BITS 14
static uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> (32 - HASH_BITS); }

/* Greedy single-probe LZ4 block compressor over content[start, end); matches may start at >= floor. */
static void lz4_compress_block(const uint8_t *content, size_t start, size_t end, size_t floor, int64_t *table, sink *out) {
    size_t anchor = start, ip = start;
    if (end - start >= 13) {
        const size_t mflimit = end - 12, matchlimit = end - 5;
        while (ip <= mflimit) {
            uint32_t h = hash4(le32(content + ip));
            int64_t cand = table[h];
            table[h] = (int64_t)ip;
            if (cand >= (int64_t)floor && ip - (size_t)cand <= 65535 && le32(content + cand) == le32(content + ip)) {
                size_t ml = 4;
                while (ip + ml < matchlimit && content[cand + ml] == content[ip + ml]) ml++;
                size_t lit = ip - anchor;
                uint8_t tok = (uint8_t)((lit >= 15 ? 15 : lit) << 4 | (ml - 4 >= 15 ? 15 : ml - 4));
                put(out, &tok, 1);
                if (lit >= 15) put_len(out, lit - 15);
                put(out, content + anchor, lit);
                uint8_t off[2] = {(uint8_t)(ip - cand), (uint8_t)((ip - cand) >> 8)};
                put(out, off, 2);
                if (ml - 4 >= 15) put_len(out, ml - 4 - 15);
                ip += ml;
                anchor = ip;
            } else {
                ip++;
            }
        }
    }
    size_t lit = end - anchor;
    uint8_t tok = (uint8_t)((lit >= 15 ? 15 : lit) << 4);
    put(out, &tok, 1);
    if (lit >= 15) put_len(out, lit - 15);
    put(out, content + anchor, lit);
}

/* flags: bit0 independent blocks, bit1 block checksums, bit2 content checksum, bit3 content size,
 * bit4 store every block uncompressed, bit5 put a skippable frame first; block_id 4..7. */
static void lz4_write_frame(const uint8_t *content, size_t n, unsigned flags, int block_id, sink *out) {
    if (flags & 32) {
        put32(out, 0x184D2A53u);
        put32(out, 5);
        put(out, "hello", 5);
    }
    put32(out, 0x184D2204u);
    uint8_t desc[14];
    size_t dn = 0;
    desc[dn++] = (uint8_t)(0x40 | ((flags & 1) << 5) | (((flags >> 1) & 1) << 4) | (((flags >> 3) & 1) << 3) | (((flags >> 2) & 1) << 2));
    desc[dn++] = (uint8_t)(block_id << 4);
    if (flags & 8)
        for (int k = 0; k < 8; k++) desc[dn++] = (uint8_t)((uint64_t)n >> (8 * k));
    put(out, desc, dn);
    uint8_t hc = (uint8_t)(pcqo_xxh32(desc, dn) >> 8);
    put(out, &hc, 1);
    const size_t bmax = (size_t)1 << (8 + 2 * block_id);
    int64_t *table = (int64_t *)malloc(sizeof(int64_t) << HASH_BITS);
    for (size_t i = 0; i < ((size_t)1 << HASH_BITS); i++) table[i] = -1;
    for (size_t at = 0; at < n; at += bmax) {
        size_t end = at + bmax < n ? at + bmax : n;
        sink blk = {0};
        if (!(flags & 16)) {
            if (flags & 1)
                for (size_t i = 0; i < ((size_t)1 << HASH_BITS); i++) table[i] = -1;
            lz4_compress_block(content, at, end, (flags & 1) ? at : 0, table, &blk);
        }
        const uint8_t *payload;
        size_t plen;
        if ((flags & 16) || blk.n >= end - at) {
            put32(out, (uint32_t)(end - at) | 0x80000000u);
            payload = content + at;
            plen = end - at;
        } else {
            put32(out, (uint32_t)blk.n);
            payload = blk.p;
            plen = blk.n;
        }
        put(out, payload, plen);
        if (flags & 2) put32(out, pcqo_xxh32(payload, plen));
        free(blk.p);
    }
    free(table);
    put32(out, 0);
    if (flags & 4) put32(out, pcqo_xxh32(content, n));
}

/* test hook: returns a malloc'd frame (caller frees with pcqo_free) */
uint8_t *pcqo_lz4f_compress(const uint8_t *content, size_t n, unsigned flags, int block_id, size_t *out_n) {
    if (block_id < 4 || block_id > 7) return NULL;
    sink s = {0};
    lz4_write_frame(content, n, flags, block_id, &s);
    if (s.oom) {
        free(s.p);
        return NULL;
    }
    *out_n = s.n;
    return s.p;
}
void pcqo_free(void *p) { free(p); }

/* LAST image -> LAZER image: same header; points regrouped into blocks of `block_size`, each attribute
 * column an LZ4 frame written with (flags, block_id).  Attribute columns the searches do not read
 * (intensity, bit fields, scan angle, user data, source id, gps time, ...) are zero-filled frames.
 * Returns a malloc'd image. */
uint8_t *pcqo_lazer_from_last(const uint8_t *last, size_t len, uint64_t block_size, unsigned flags, int block_id, size_t *out_n) {
    pcqo_las_header h;
    if (pcqo_parse_las_header(last, len, 0, &h) || block_size == 0 || block_id < 4 || block_id > 7) return NULL;
    unsigned f = h.point_data_record_format;
    uint64_t n = h.number_of_points, otp = h.offset_to_point_data;
    size_t nattr = 8 + fmt_color(f) + fmt_gps(f) + fmt_wave(f) + fmt_nir(f);
    uint64_t cls_in_point = f <= 5 ? 15 : 16;
    uint64_t col_in_point = f == 2 ? 20 : (f == 3 || f == 5) ? 28 : 30; /* 7, 8, 10: colour after gps time */
    if (otp + n * cls_in_point + n > len) return NULL;
    if (fmt_color(f) && otp + n * col_in_point + n * 6 > len) return NULL;
    uint64_t nb = n / block_size + (n % block_size != 0);
    sink s = {0};
    put(&s, last, otp);
    put64(&s, block_size);
    size_t table_at = s.n;
    for (uint64_t b = 0; b < nb; b++) put64(&s, 0);
    /* sizes of the other attribute columns per point: 1 intensity u16, 2 bit byte, 4 scan angle, 5 user
     * data, 6 point source id u16, 7.. extras; only their existence matters */
    static const size_t other_size[12] = {0, 2, 1, 0, 1, 1, 2, 1, 0, 8, 8, 2};
    for (uint64_t b = 0; b < nb; b++) {
        uint64_t first = b * block_size, cnt = n - first < block_size ? n - first : block_size;
        uint64_t block_at = s.n;
        memcpy(s.p + table_at + 8 * b, &block_at, 8);
        size_t attr_table = s.n;
        for (size_t k = 0; k < nattr; k++) put64(&s, 0);
        for (size_t k = 0; k < nattr; k++) {
            uint64_t here = s.n;
            memcpy(s.p + attr_table + 8 * k, &here, 8);
            if (k == 0) lz4_write_frame(last + otp + first * 12, cnt * 12, flags, block_id, &s);
            else if (k == 3) lz4_write_frame(last + otp + n * cls_in_point + first, cnt, flags, block_id, &s);
            else if (k == 8 && fmt_color(f)) lz4_write_frame(last + otp + n * col_in_point + first * 6, cnt * 6, flags, block_id, &s);
            else {
                size_t zn = cnt * other_size[k];
                uint8_t *z = (uint8_t *)calloc(zn ? zn : 1, 1);
                lz4_write_frame(z, zn, flags, block_id, &s);
                free(z);
            }
        }
    }
    if (s.oom) {
        free(s.p);
        return NULL;
    }
    *out_n = s.n;
    return s.p;
}
