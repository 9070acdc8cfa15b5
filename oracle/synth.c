/*
 * synth.c — deterministic synthetic LAS / LAST data.
 *
 * TEST INFRASTRUCTURE ONLY (see pcq_oracle.h).  The reference ships no sample data and its LAST
 * writer lives in another repository (README.md:29), so test inputs are generated here from the
 * format as the reference's readers define it:
 *   LAS 1.2 header, 227 bytes, field order per query/src/las.rs:7-40;
 *   LAS record layout (formats 0-3) per the offsets the scans use (las.rs:38-45, 102-135, 202-212);
 *   LAST = the LAS record transposed attribute-by-attribute: attribute at record offset `o` with
 *   size `s` occupies [otp + N*o, otp + N*(o+s))  (readers/src/last_reader.rs:83-144, last.rs:68-90).
 *
 * All arithmetic is integer (counter-based splitmix64 + 64x64->128 multiply-high) so that the
 * device-side generator the bench uses (csrc/synth_kernels.hip) produces the same bits.
 */
#define _GNU_SOURCE
#include "pcq_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

static const uint16_t k_rec_len[4] = {20, 28, 26, 34};

uint64_t pcqo_synth_mix(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint32_t mulhi(uint64_t h, uint32_t span) {
    return (uint32_t)(((unsigned __int128)h * (unsigned __int128)span) >> 64);
}

static inline void gen_xyz(const pcqo_synth_spec *s, uint64_t i, int32_t out[3], uint64_t *aux) {
    uint64_t h0 = pcqo_synth_mix(s->seed, 8 * i + 0);
    uint64_t h1 = pcqo_synth_mix(s->seed, 8 * i + 1);
    uint64_t h2 = pcqo_synth_mix(s->seed, 8 * i + 2);
    uint64_t h3 = pcqo_synth_mix(s->seed, 8 * i + 3);
    out[0] = (int32_t)((int64_t)s->lo[0] + (int64_t)mulhi(h0, s->span[0]));
    out[1] = (int32_t)((int64_t)s->lo[1] + (int64_t)mulhi(h1, s->span[1]));
    if (((h3 >> 16) & 0xFFFF) < s->zo_prob16)
        out[2] = (int32_t)((int64_t)s->zo_lo + (int64_t)mulhi(h2, s->zo_span));
    else
        out[2] = (int32_t)((int64_t)s->lo[2] + (int64_t)mulhi(h2, s->span[2]));
    *aux = h3;
}

static inline uint8_t gen_class(const pcqo_synth_spec *s, uint64_t aux) {
    uint32_t u = (uint32_t)(aux & 0xFFFF);
    if (s->n_classes == 0) return 0;
    for (uint32_t j = 0; j < s->n_classes; j++)
        if (u < s->cls_cum16[j]) return s->cls_val[j];
    return s->cls_val[s->n_classes - 1];
}

void pcqo_synth_fill_columns(const pcqo_synth_spec *s, uint64_t first, uint64_t count, int32_t *xyz,
                             uint8_t *cls) {
    for (uint64_t k = 0; k < count; k++) {
        int32_t p[3];
        uint64_t aux;
        gen_xyz(s, first + k, p, &aux);
        if (xyz) {
            xyz[3 * k + 0] = p[0];
            xyz[3 * k + 1] = p[1];
            xyz[3 * k + 2] = p[2];
        }
        if (cls) cls[k] = gen_class(s, aux);
    }
}

static void put16(uint8_t *p, uint16_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
}
static void put32(uint8_t *p, uint32_t v) {
    put16(p, (uint16_t)v);
    put16(p + 2, (uint16_t)(v >> 16));
}
static void putf64(uint8_t *p, double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    put32(p, (uint32_t)u);
    put32(p + 4, (uint32_t)(u >> 32));
}

/* One AoS LAS record for point i (formats 0-3). */
static void gen_record(const pcqo_synth_spec *s, uint64_t i, uint8_t *rec) {
    int32_t p[3];
    uint64_t aux;
    gen_xyz(s, i, p, &aux);
    put32(rec + 0, (uint32_t)p[0]);
    put32(rec + 4, (uint32_t)p[1]);
    put32(rec + 8, (uint32_t)p[2]);
    put16(rec + 12, (uint16_t)(aux >> 32));      /* intensity              */
    rec[14] = (uint8_t)(aux >> 48);              /* return / flag bits      */
    rec[15] = gen_class(s, aux);                 /* classification          */
    rec[16] = (uint8_t)(aux >> 56);              /* scan angle rank         */
    rec[17] = 0;                                 /* user data               */
    uint64_t h4 = pcqo_synth_mix(s->seed, 8 * i + 4);
    put16(rec + 18, (uint16_t)(h4 >> 48));       /* point source id         */
    uint32_t rgb_at = 20;
    if (s->format == 1 || s->format == 3) {
        putf64(rec + 20, (double)i * 0.001);     /* gps time                */
        rgb_at = 28;
    }
    if (s->format == 2 || s->format == 3) {
        put16(rec + rgb_at + 0, (uint16_t)h4);
        put16(rec + rgb_at + 2, (uint16_t)(h4 >> 16));
        put16(rec + rgb_at + 4, (uint16_t)(h4 >> 32));
    }
}

size_t pcqo_synth_image_size(const pcqo_synth_spec *s) {
    if (s->format > 3) return 0;
    return 227 + (size_t)s->n * k_rec_len[s->format];
}

int pcqo_synth_build_header(const pcqo_synth_spec *s, uint8_t out[227]) {
    if (s->format > 3 || s->n > 0xFFFFFFFFull) return PCQO_ERR_ARG;
    memset(out, 0, 227);
    memcpy(out, "LASF", 4);
    out[24] = 1;
    out[25] = 2;
    memcpy(out + 26, "pcq-synth", 9);
    memcpy(out + 58, "pcq-synth", 9);
    put16(out + 90, 1);
    put16(out + 92, 2026);
    put16(out + 94, 227);
    put32(out + 96, 227);
    put32(out + 100, 0);
    out[104] = (uint8_t)s->format;
    put16(out + 105, k_rec_len[s->format]);
    put32(out + 107, (uint32_t)s->n);
    put32(out + 111, (uint32_t)s->n);
    for (int a = 0; a < 3; a++) {
        putf64(out + 131 + 8 * a, s->scale[a]);
        putf64(out + 155 + 8 * a, s->offset[a]);
        int64_t lo = s->lo[a], hi = (int64_t)s->lo[a] + (int64_t)s->span[a] - 1;
        if (a == 2 && s->zo_prob16 > 0) {
            int64_t zlo = s->zo_lo, zhi = (int64_t)s->zo_lo + (int64_t)s->zo_span - 1;
            if (zlo < lo) lo = zlo;
            if (zhi > hi) hi = zhi;
        }
        /* header bounds = the extreme reconstructable coordinates, built like last.rs:156-160 */
        double a0 = ((double)lo * s->scale[a]) + s->offset[a];
        double a1 = ((double)hi * s->scale[a]) + s->offset[a];
        double mn = a0 < a1 ? a0 : a1, mx = a0 < a1 ? a1 : a0;
        putf64(out + 179 + 16 * a, mx);
        putf64(out + 187 + 16 * a, mn);
    }
    return PCQO_OK;
}

typedef struct {
    const pcqo_synth_spec *s;
    int transposed;
    uint8_t *out;
    uint64_t first, count;
} build_job;

static void *build_worker(void *arg) {
    build_job *j = (build_job *)arg;
    const pcqo_synth_spec *s = j->s;
    const uint64_t n = s->n;
    const uint32_t rl = k_rec_len[s->format];
    uint8_t *pd = j->out + 227;
    uint8_t rec[34];
    /* attribute table of the LAS record: (offset, size) */
    uint32_t att_off[9], att_sz[9], na = 0;
    att_off[na] = 0, att_sz[na++] = 12;  /* XYZ        */
    att_off[na] = 12, att_sz[na++] = 2;  /* intensity  */
    att_off[na] = 14, att_sz[na++] = 1;  /* bit fields */
    att_off[na] = 15, att_sz[na++] = 1;  /* class      */
    att_off[na] = 16, att_sz[na++] = 1;  /* scan angle */
    att_off[na] = 17, att_sz[na++] = 1;  /* user data  */
    att_off[na] = 18, att_sz[na++] = 2;  /* source id  */
    if (s->format == 1 || s->format == 3) att_off[na] = 20, att_sz[na++] = 8;
    if (s->format == 2) att_off[na] = 20, att_sz[na++] = 6;
    if (s->format == 3) att_off[na] = 28, att_sz[na++] = 6;
    for (uint64_t i = j->first; i < j->first + j->count; i++) {
        gen_record(s, i, rec);
        if (!j->transposed) {
            memcpy(pd + i * rl, rec, rl);
        } else {
            for (uint32_t a = 0; a < na; a++)
                memcpy(pd + n * att_off[a] + i * att_sz[a], rec + att_off[a], att_sz[a]);
        }
    }
    return NULL;
}

int pcqo_synth_build_image(const pcqo_synth_spec *s, int transposed, uint8_t *out, size_t cap,
                           int threads) {
    size_t need = pcqo_synth_image_size(s);
    if (need == 0 || cap < need) return PCQO_ERR_ARG;
    int rc = pcqo_synth_build_header(s, out);
    if (rc) return rc;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t th[64];
    build_job jobs[64];
    uint64_t per = (s->n + (uint64_t)threads - 1) / (uint64_t)threads;
    int started = 0;
    for (int t = 0; t < threads; t++) {
        uint64_t first = per * (uint64_t)t;
        if (first >= s->n) break;
        uint64_t count = s->n - first < per ? s->n - first : per;
        jobs[t] = (build_job){s, transposed, out, first, count};
        pthread_create(&th[t], NULL, build_worker, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    return PCQO_OK;
}

int pcqo_synth_write_file(const pcqo_synth_spec *s, int transposed, const char *path, int threads) {
    size_t need = pcqo_synth_image_size(s);
    if (need == 0) return PCQO_ERR_ARG;
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return PCQO_ERR_IO;
    if (ftruncate(fd, (off_t)need) != 0) {
        close(fd);
        return PCQO_ERR_IO;
    }
    uint8_t *p = (uint8_t *)mmap(NULL, need, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return PCQO_ERR_IO;
    int rc = pcqo_synth_build_image(s, transposed, p, need, threads);
    munmap(p, need);
    return rc;
}
