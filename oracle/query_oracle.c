/*
 * query_oracle.c — CPU restatement of the reference's `query` CLI (query/src/main.rs) on top of the
 * oracle scans.  TEST INFRASTRUCTURE ONLY: tests run it beside the product's `query` binary on the
 * same inputs and compare stdout (modulo the timing line) and exit status.
 *
 * Restates: get_all_input_files main.rs:29-57 · parse_aabb :59-92 · get_total_bounds :94-120 ·
 * run_search_sequential :122-144 · run_search_parallel :146-183 · is_valid_file :185-189 ·
 * main :191-319.  Only the `--optimized` arms for .las/.last exist here (SURVEY.md §2 rows 12-15 are
 * out of scope); anything else is reported as an error.
 */
#define _GNU_SOURCE
#include "pcq_oracle.h"

#include <dirent.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

typedef struct {
    char **v;
    size_t n, cap;
} strvec;

static void sv_push(strvec *s, const char *p) {
    if (s->n == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 16;
        s->v = (char **)realloc(s->v, s->cap * sizeof(char *));
    }
    s->v[s->n++] = strdup(p);
}

static const char *ext_of(const char *path) {
    const char *base = strrchr(path, '/');
    base = base ? base + 1 : path;
    const char *dot = strrchr(base, '.');
    if (!dot || dot == base) return NULL;
    return dot + 1;
}

/* main.rs:185-189 */
static int is_valid_file(const char *p) {
    const char *e = ext_of(p);
    return e && (!strcmp(e, "las") || !strcmp(e, "laz") || !strcmp(e, "last") || !strcmp(e, "lazer"));
}

/* main.rs:29-57 */
static int get_all_input_files(const char *input, strvec *out) {
    struct stat st;
    if (stat(input, &st) != 0) {
        fprintf(stderr, "Error: Input path %s does not exist!\n", input);
        return 1;
    }
    if (S_ISREG(st.st_mode)) {
        sv_push(out, input);
        return 0;
    }
    if (S_ISDIR(st.st_mode)) {
        DIR *d = opendir(input);
        if (!d) {
            fprintf(stderr, "Error: %s\n", strerror(errno));
            return 1;
        }
        struct dirent *e;
        while ((e = readdir(d))) {
            if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
            char buf[4096];
            size_t l = strlen(input);
            snprintf(buf, sizeof buf, "%s%s%s", input, (l && input[l - 1] == '/') ? "" : "/", e->d_name);
            sv_push(out, buf);
        }
        closedir(d);
        return 0;
    }
    fprintf(stderr, "Error: Input path %s is neither file nor directory!\n", input);
    return 1;
}

/* Rust str::parse::<f64> accepts a subset of strtod's grammar: no leading whitespace, no hex. */
static int parse_f64(const char *s, double *v) {
    if (!*s || *s == ' ' || *s == '\t' || *s == '\n') return 0;
    const char *q = s;
    if (*q == '+' || *q == '-') q++;
    if (q[0] == '0' && (q[1] == 'x' || q[1] == 'X')) return 0;
    char *end;
    errno = 0;
    *v = strtod(s, &end);
    return end != s && *end == 0;
}

/* main.rs:59-92; returns 0 ok, 1 parse error */
static int parse_aabb(const char *str, double mn[3], double mx[3]) {
    double c[6];
    int n = 0;
    char *dup = strdup(str), *save = dup;
    for (;;) {
        char *semi = strchr(dup, ';');
        if (semi) *semi = 0;
        if (n >= 6 || !parse_f64(dup, &c[n])) {
            free(save);
            return 1;
        }
        n++;
        if (!semi) break;
        dup = semi + 1;
    }
    free(save);
    if (n != 6) return 1;
    for (int a = 0; a < 3; a++) mn[a] = c[a], mx[a] = c[3 + a];
    return 0;
}

static int read_header(const char *path, int mask, pcqo_las_header *h) {
    FILE *f = fopen(path, "rb");
    if (!f) return PCQO_ERR_IO;
    uint8_t buf[4096];
    size_t n = fread(buf, 1, sizeof buf, f);
    fclose(f);
    return pcqo_parse_las_header(buf, n, mask, h);
}

int main(int argc, char **argv) {
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const char *input = NULL, *bounds_s = NULL, *class_s = NULL, *output = NULL, *density_s = NULL;
    int parallel = 0, optimized = 0;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        const char **dst = NULL;
        if (!strcmp(a, "-i") || !strcmp(a, "--input")) dst = &input;
        else if (!strcmp(a, "--bounds")) dst = &bounds_s;
        else if (!strcmp(a, "--class")) dst = &class_s;
        else if (!strcmp(a, "-o") || !strcmp(a, "--output")) dst = &output;
        else if (!strcmp(a, "--density")) dst = &density_s;
        else if (!strcmp(a, "--parallel")) { parallel = 1; continue; }
        else if (!strcmp(a, "--optimized")) { optimized = 1; continue; }
        else {
            fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", a);
            return 1;
        }
        if (i + 1 >= argc) {
            fprintf(stderr, "error: The argument '%s' requires a value but none was supplied\n", a);
            return 1;
        }
        *dst = argv[++i];
    }
    if (!input) {
        fprintf(stderr, "error: The following required arguments were not provided:\n    --input <FILE>\n");
        return 1;
    }

    strvec all = {0}, files = {0};
    if (get_all_input_files(input, &all)) return 1; /* main.rs:222-223 */
    for (size_t i = 0; i < all.n; i++)
        if (is_valid_file(all.v[i])) sv_push(&files, all.v[i]);

    uint64_t total_size = 0; /* main.rs:227-231 */
    for (size_t i = 0; i < files.n; i++) {
        struct stat st;
        if (stat(files.v[i], &st) == 0) total_size += (uint64_t)st.st_size;
    }
    double total_mib = (double)total_size / 1048576.0;

    double bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0}, density = 0;
    unsigned cls = 0;
    if (bounds_s && parse_aabb(bounds_s, bmin, bmax)) { /* main.rs:235 expect() -> panic */
        fprintf(stderr, "Could not prase argument BOUNDS\n");
        return 101;
    }
    if (bounds_s)
        for (int a = 0; a < 3; a++)
            if (bmin[a] > bmax[a]) { /* AABB::from_min_max panic [recalled] */
                fprintf(stderr, "AABB::from_min_max: Minimum position must be <= maximum position!\n");
                return 101;
            }
    if (class_s) { /* main.rs:236 */
        char *end;
        long v = strtol(class_s, &end, 10);
        if (end == class_s || *end || v < 0 || v > 255 || class_s[0] == '-' || class_s[0] == ' ') {
            fprintf(stderr, "Could not prase argument CLASS\n");
            return 101;
        }
        cls = (unsigned)v;
    }
    if (density_s && !parse_f64(density_s, &density)) { /* main.rs:237 */
        fprintf(stderr, "Could not prase argument DENSITY\n");
        return 101;
    }
    if (bounds_s && class_s) { /* main.rs:238-240 */
        fprintf(stderr, "Error: Specifying BOUNDS and CLASS at the same time is invalid! Specify "
                        "either BOUNDS or CLASS argument!\n");
        return 1;
    }
    if (!bounds_s && !class_s) { /* main.rs:242-244 */
        fprintf(stderr, "Error: Found neither BOUNDS nor CLASS argument but exactly one of these "
                        "arguments is required!\n");
        return 1;
    }
    int kind = bounds_s ? PCQO_QUERY_BOUNDS : PCQO_QUERY_CLASS;

    double gmin[3], gmax[3];
    if (density_s) { /* main.rs:253-264 */
        if (bounds_s) {
            memcpy(gmin, bmin, sizeof gmin);
            memcpy(gmax, bmax, sizeof gmax);
        } else { /* get_total_bounds, main.rs:94-120 */
            for (int a = 0; a < 3; a++) gmin[a] = 1.7976931348623157e308, gmax[a] = -1.7976931348623157e308;
            for (size_t i = 0; i < files.n; i++) {
                const char *e = ext_of(files.v[i]);
                pcqo_las_header h;
                int rc;
                if (e && !strcmp(e, "lazer")) /* main.rs:102-107: the full LAZERSource::from */
                    rc = pcqo_lazer_file_bounds(files.v[i], h.min, h.max);
                else
                    rc = read_header(files.v[i], e && !strcmp(e, "last"), &h);
                if (rc == PCQO_ERR_PANIC) {
                    fprintf(stderr, "thread 'main' panicked: %s\n", pcqo_last_error());
                    return 101;
                }
                if (rc) {
                    fprintf(stderr, "Error: %s\n", pcqo_last_error());
                    return 1;
                }
                for (int a = 0; a < 3; a++) { /* AABB::union */
                    if (h.min[a] < gmin[a]) gmin[a] = h.min[a];
                    if (h.max[a] > gmax[a]) gmax[a] = h.max[a];
                }
            }
        }
    }
    if (output) { /* FileDumper::new, dump_points.rs:46-53 */
        struct stat st;
        if (stat(output, &st) != 0) {
            fprintf(stderr, "Error: Path %s does not exist!\n", output);
            return 1;
        }
        if (!S_ISDIR(st.st_mode)) {
            fprintf(stderr, "Error: Path %s is no directory!\n", output);
            return 1;
        }
    }
    if (!optimized) { /* .lazer has one implementation for both settings (searcher.rs:83, :144) */
        for (size_t i = 0; i < files.n; i++) {
            const char *e = ext_of(files.v[i]);
            if (!e || strcmp(e, "lazer")) {
                fprintf(stderr, "Error: the oracle restates only the --optimized search implementation\n");
                return 1;
            }
        }
    }

    printf("Searching %zu files...\n", files.n); /* main.rs:289 */

    size_t ncoll = parallel ? files.n : 1;
    pcqo_collector **coll = (pcqo_collector **)calloc(ncoll ? ncoll : 1, sizeof *coll);
    for (size_t i = 0; i < ncoll; i++) { /* collector_factory, main.rs:253-273 */
        if (density_s) coll[i] = pcqo_collector_new_grid(gmin, gmax, density);
        else if (output) coll[i] = pcqo_collector_new_buffer();
        else coll[i] = pcqo_collector_new_count();
        if (!coll[i]) {
            fprintf(stderr, "Error: %s\n", pcqo_last_error());
            return 1;
        }
    }
    /* The per-file work is order-independent in parallel mode; the oracle runs it in file order. */
    for (size_t i = 0; i < files.n; i++) {
        int recsz = -1;
        int rc = pcqo_search_file(files.v[i], kind, bmin, bmax, (uint8_t)cls,
                                  coll[parallel ? i : 0], &recsz);
        if (recsz >= 0) printf("Point record size: %d\n", recsz); /* las.rs:73 */
        if (rc == PCQO_ERR_PANIC) {
            fprintf(stderr, "%s\n", pcqo_last_error());
            return 101;
        }
        if (rc) {
            fprintf(stderr, "Error: %s\n", pcqo_last_error());
            return 1;
        }
    }
    /* drain: main.rs:135-141 (sequential) / :164-180 (parallel) */
    int have_matches = 0;
    uint64_t matches = 0;
    for (size_t i = 0; i < ncoll; i++) {
        if (pcqo_collector_has_points(coll[i])) {
            uint64_t n = pcqo_collector_point_count(coll[i]);
            if (output && n > 0) printf("Writing %llu points\n", (unsigned long long)n); /* dump_points.rs:108 */
        } else {
            have_matches = 1;
            matches += pcqo_collector_point_count(coll[i]);
        }
    }
    if (have_matches) printf("Found %llu matching points\n", (unsigned long long)matches);

    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    double mibs = ((double)total_size / secs) / 1048576.0;
    printf("Searched %.2f MiB in %.2fs (throughput: %.2fMiB/s)\n", total_mib, secs, mibs); /* :313-316 */
    return 0;
}
