// grid.hip — GridSampledCollector / SparseGrid on the device (kernel K4): partition by cell key, fold in LDS.
//
// Restates query/src/grid_sampling.rs:49-105 (SparseGrid::insert_point).  The reference folds points sequentially
// into a HashMap<u64, Point>: a cell keeps the point closest to the cell centre, replaced only when a later point is
// STRICTLY closer, so the earliest point in file order wins ties.  For every cell key whose points all fall into the
// same unmasked cell (always, except the mask-aliasing case below) that fold is the lexicographic arg-min of
// (squared distance, file-order index).
//
// A hash table in HBM costs one random 128-byte line per matched point (round 1: 29 ms per 163 M-point file at 10 m).
// Here the random access happens in LDS, and what travels through HBM in between is written and read in order:
//   pass 0  (per scan, asynchronous, ONE reading of the points)  k_p0_part: a tile of 5120 points becomes one
//           BLOCK of tuples {x, y, z, index, class | entry [, colour]} (20 or 24 bytes), sorted in LDS by the level-1
//           bin of the tuple's cell key (top 9 bits of hash(key)) and written to its own place — tile t's block is at
//           t x 5120 tuples — as one sequential stream, next to a 513-entry directory row (where each bin starts in
//           the block).  No histogram pass, no cursors, no atomics in global memory, nothing read back (round 2
//           counted first — a second reading of the points — and scattered runs of 10 tuples to 131 072 cursors).
//           Before the sort a tile FOLDS ITS OWN DUPLICATES: the tuples of one cell key inside a tile are
//           consecutive in file order, so for a key without aliased tuples the fold's result cannot change when
//           only the tile's (distance, file order) minimum travels on (proof at k_p0_part).  A scan-ordered file
//           (flight lines: hundreds of consecutive points per coarse cell) sheds most of its tuples there; a file in
//           random order sheds none, and a workgroup that sees that stops trying for a while.
//   fold    (lazy: when a result is asked for, or when too much is pending)  the directory rows are transposed
//           into per-bin fragment lists (k_dir_transpose, k_bin_prefix): bin b = the pieces [start, start + count)
//           of every tile's block.  A reader keeps a window of that list in LDS and turns "tuples j .. j + chunk of
//           bin b" into addresses by binary search, so the consumers still see dense chunks.  One workgroup per
//           partition folds its tuples into an open-addressing table in LDS — atomicMin on the f64 distance bits, then
//           on the file order among the tuples at the minimum, then the winner parks its payload — and writes one
//           32-byte record + key per cell, coalesced.  A coarse grid folds its level-1 bins directly (k_fold<BIG>:
//           a CU's whole LDS as one 6400-slot table); a denser grid first gets a second partition level (k_level2:
//           one pass into fixed regions with slack, fan-out chosen from a measured estimate of the distinct cells per
//           bin) and folds the small partitions two workgroups to a CU (k_fold_dense; k_fold<SMALL> for what that
//           leaves: earlier winners, partitions longer than a chunk).
// What the kernels had to learn about gfx950: every pass is bound by vector instructions before
// it is bound by memory unless the cell arithmetic is cut down (cell_fast); loads and stores share one in-order
// counter, so a prefetch must be waited for before the stores behind it are issued; pointers loaded from memory make
// flat loads, which also hold every LDS wait; a device-scope fence writes the L2 back; registers spilled to scratch are
// HBM traffic (the dense fold: 4.8 GB each way per file until it ran with more registers and fewer waves).
// The folded winners are kept grouped by partition, so a later fold (more scans into the same collector: sequential
// mode shares one grid, main.rs:129-133; a file streamed in chunks) merges them with the new tuples partition by
// partition: an old winner is earlier in file order than every new tuple and its distance is recomputed from its
// record, bit for bit.
//
// Mask aliasing (grid_sampling.rs:62-82): the key masks each axis to `bits`, but the cell centre uses the UNMASKED
// cell, so a cell >= 2^bits folds onto another key while comparing against a different centre.  For such keys the
// result depends on the visiting order; their slots are flagged during the fold and the key is re-folded exactly,
// in file order, from its tuples (k_alias_*: gather, rank sort, sequential replay of insert_point).
//
// Integer / f64 work over streamed tuples (written and read once per partition level); not reshaped into GEMMs.
#include <algorithm>
#include <cmath>

#include "dev_common.h"

using namespace pcqdev;

namespace {

constexpr int F1_BITS = 9;
constexpr int F1 = 1 << F1_BITS;       // level-1 bins: the top F1_BITS bits of hash(key)
constexpr int F2_MAX = 4096;           // largest second-level fan-out
constexpr int P0_NT = 1024, P0_ITEMS = 5;      // pass 0: one workgroup per CU sorts tiles of 5120 points in LDS
constexpr int P0_TILE = P0_NT * P0_ITEMS;      // points per tile = tuples a tile's block has room for
constexpr int DIR_STRIDE = 520;        // u16 per directory row: [b] = first place of bin b in the block, [512] = tuples in the block
constexpr int DIR_WORDS = F1 / 2 + 1;  // the 513 entries as 32-bit words
constexpr int AGG_SLOTS = 8192;        // pass 0: slots of the tile's duplicate table
constexpr int AGG_POS_BITS = 13;       // a tuple's place in its tile (< 5120) in the low bits of a table word
static_assert(P0_TILE <= (1 << AGG_POS_BITS) && P0_TILE <= 65535, "tile places fit the table word and the 16-bit directory");
static_assert((P0_TILE * 20) % 16 == 0 && (P0_TILE * 24) % 16 == 0, "a tile's block starts 16-byte aligned and holds whole 16-byte words");
// The fold's shapes.  BIG: one 1024-thread workgroup owns a CU's whole LDS — 6400 slots of {key, distance, file order},
// the winner's payload parked in HBM scratch — and folds a level-1 bin directly (coarse grids: few cells, many tuples
// per cell).  SMALL / DENSE: 2048 slots, three / two workgroups per CU, a whole partition of the second level (about 1000
// cells, 1330 tuples) in registers, so that the winner of a cell writes its record straight from there (dense grids:
// about one tuple per cell, many small partitions) — 256 threads x 6 tuples in the general kernel, 512 x 3 in
// k_fold_dense.
constexpr int BIG_SLOTS = 6400, BIG_NT = 1024, BIG_LIMIT = 5440, BIG_DIRECT = 4700;
constexpr int SMALL_SLOTS = 2048, SMALL_NT = 256, SMALL_K = 6, SMALL_LIMIT = 1740, SMALL_TARGET = 1000;
constexpr int BIG_K = 4;               // fold: tuples per thread and chunk
constexpr int BIG_FB = 512;            // big fold: fragments in the reader's window (what is left of the LDS)
constexpr int DENSE_NT = 512, DENSE_K = 3;  // k_fold_dense: the same chunk (1536 tuples) on twice the waves
constexpr int L2_NT = 512;             // exact second level: threads per workgroup (one workgroup per level-1 bin)
constexpr int L2_UNROLL = 4;
constexpr int L2_FB = 1024;            // exact second level, alias gather: fragments in the reader's window
constexpr int L2S_NT = 1024, L2S_ITEMS = 4, L2S_TILE = L2S_NT * L2S_ITEMS;  // k_level2: one workgroup per CU, tiles of 4096 tuples
constexpr int L2S_FB = 2048;           // k_level2: fragments in the reader's window (about five tiles)
constexpr int L2_STAGED_F2 = 1024;     // largest fan-out of the staged form (its per-tile tables live in LDS)
constexpr int PROBE_BINS = 2;          // bins whose distinct cells are counted to estimate the grid's density
constexpr uint64_t ALIAS_QUADRATIC = 8192;  // aliased tuples up to which the replay order comes from the quadratic rank kernel
constexpr int MAX_RUNS = 1024;         // pending pass-0 runs per collector before a fold is forced
constexpr uint64_t RUN_POINTS = 1ull << 30;  // points per pass-0 run
constexpr uint64_t PENDING_MAX = (1ull << 32) - RUN_POINTS - 1;  // tuple counts and offsets of a fold are 32-bit

constexpr uint8_t R_HAS = 1;    // byte 31 of a winner record: the record holds a point
constexpr uint8_t R_ALIAS = 2;  // the key has seen a point whose unmasked cell differs from the masked one (sticky)

// One matched point on its way to the fold, as the kernels hold it.  `idx` is the file-order index relative to its
// entry's base index.  In memory: 20 bytes {x, y, z, idx, w0 (sel in the red half)}, or 24 with w1 when the scan had a colour column
// (synthetic ca13 is format 1: no colour — a sixth less to move through every partition level).
struct GridTuple {
    int32_t x, y, z;
    uint32_t idx;
    uint32_t w0;  // classification | entry << 8 | red << 16
    uint32_t w1;  // green | blue << 16
    uint32_t sel; // 20-byte tuples only: the 16 hash bits that pick the second-level partition (sel16_of), which pass 0 —
                  // it has the hash at hand — leaves in the red half of w0 (a tuple without colour has no red); ~0 otherwise
};
__host__ __device__ __forceinline__ uint32_t tuple_bytes(bool wide) { return wide ? 24u : 20u; }

// What turns a tuple's integers back into a position: the header scale / offset of the file it came from
// (last.rs:156-160).  Consecutive scans with the same scale and offset share one entry.
struct GridEntryDev {
    double scale[3], offset[3];
};

// Tuples cut into partitions, back to back or in regions (the second level's output): partition p is the tuples
// off[p] .. off[p] + cnt[p] — or, without cnt, .. off[p + 1] — of `tuples`, `wide` saying how long a tuple is.
struct GridSeg {
    const uint8_t *tuples;
    const uint32_t *off;
    const uint32_t *cnt;
    uint32_t wide;
};

// Pass 0's output as the fold reads it: bin b = fragment t of every tile t (all pending runs, in scan order), fragment
// (b, t) = startT[b][t] .. of tile t's block, preT[b][t] tuples of the bin in front of it.
struct BinSrc {
    const uint32_t *preT;       // [F1][Tp1]; preT[b][T] = the bin's tuples
    const uint16_t *startT;     // [F1][Tp]
    const uint64_t *tile_addr;  // [T] the block's address | 1 when its tuples are 24 bytes
    uint32_t T, Tp1, Tp;
};

struct AliasItem {  // a tuple of an aliased key, for the exact replay (key at +0, order at +8: alias_sort.hip)
    uint64_t key, ord;
    int32_t x, y, z;
    uint32_t w0, w1, _pad;
};

struct CellInfo {
    uint64_t key;
    uint64_t cell[3];  // unmasked
    bool alias;
};

// grid_sampling.rs:51-70.  The cell is trunc(RN(num / extent)) — the correctly rounded quotient, truncated (Rust `as u64`).
// A correctly rounded f64 division is ~15 instructions, three per point, in every pass over the matches; but the
// quotient itself is not needed, only its integer part.  q = num * (1 / extent) lies within 2 ulp of the true
// quotient x and RN(x) within half an ulp, so when q is further than q * 2^-50 (>= 4 ulp) from an integer — and
// 0 <= q < 2^51 — no integer lies between them and trunc(q) IS trunc(RN(x)).  Everything else (a point within a few
// ulp of a cell boundary, negative, huge, NaN) takes the division, so the result is the reference's in every case.
__device__ __forceinline__ uint64_t cell_index(double num, double extent, double inv_extent) {
    const double q = num * inv_extent;
    const double fl = floor(q);
    const double frac = q - fl, guard = q * 0x1p-50;
    if (q >= 0.0 && q < 0x1p51 && frac > guard && 1.0 - frac > guard) return (uint64_t)fl;
    return f64_as_u64(num / extent);
}

__device__ __forceinline__ CellInfo cell_of(const DevGrid &g, double px, double py, double pz) {
    const double p[3] = {px, py, pz};
    CellInfo ci;
    ci.alias = false;
    ci.key = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double num = (p[a] - g.bmin[a]) * g.dims_f[a];
        const uint64_t cell = cell_index(num, g.bmax[a] - g.bmin[a], g.inv_extent[a]);
        ci.cell[a] = cell;
        const uint64_t masked = cell & g.mask[a];
        ci.alias |= masked != cell;
        ci.key |= masked << g.shift[a];
    }
    return ci;
}

// grid_sampling.rs:78-95 — squared distance of (px,py,pz) to the centre of the unmasked cell.
__device__ __forceinline__ double centre_dist(const DevGrid &g, const uint64_t (&cell)[3], double px, double py, double pz) {
    const double cx = ((double)cell[0] + 0.5) * g.cell_size + g.bmin[0];
    const double cy = ((double)cell[1] + 0.5) * g.cell_size + g.bmin[1];
    const double cz = ((double)cell[2] + 0.5) * g.cell_size + g.bmin[2];
    const double dx = px - cx, dy = py - cy, dz = pz - cz;
    const double a = dx * dx, b = dy * dy, c = dz * dz;
    return (a + b) + c;
}

// The same cell, the short way.  With cell_of + the murmur hash a point cost ~125 vector instructions per pass and the passes
// were bound by them; the common case is cut down to what it needs (today the exact f64 arithmetic is 19-47 of a pass's
// 115-280 vector instructions per 64 points and vector issue is 17-31 % busy: profiles/r03_grid_valu_mix.txt):
//   q = (p - bmin) * k, k = RN(dims / extent) computed once on the host — one multiply instead of two.  q is within
//   (1 + 2^-53)^3 of the exact quotient num / extent the reference rounds (num = RN((p - bmin) * dims)), and RN of that is
//   another half ulp away: |q - RN(num / extent)| < 4.01 * 2^-53 * q.  With q < qmax <= 2^31 and both q - floor(q) and
//   1 - (q - floor(q)) above guard = qmax * 2^-50 (= 8 * 2^-53 * qmax) no integer lies between the two, so
//   floor(q) IS trunc(RN(num / extent)).  A negative q (the reference's `as u64` saturates to 0) gives 0 as well: the
//   conversion saturates, and the quotient has the sign of p - bmin either way.
//   The cell fits 32 bits then: one conversion instruction each way instead of the emulated 64-bit ones.
// Anything else — within the guard of a cell boundary, beyond qmax, NaN, a grid with a zero extent — reports !ok and the
// caller takes cell_of().
struct CellFast {
    uint32_t c[3];
    double f[3];  // c as f64
    bool ok;
};
// what the short computation reads of the grid (the kernel arguments of k_fold_dense: 32 scalar registers instead of 53)
struct DevGridFast {
    double bmin[3], qk[3], qmax[3], guard[3];
    double cell_size;
    uint32_t mask[3], shift[3];
};
template <typename G>
__device__ __forceinline__ CellFast cell_fast(const G &g, double px, double py, double pz) {
    const double p[3] = {px, py, pz};
    CellFast r;
    r.ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double q = (p[a] - g.bmin[a]) * g.qk[a];
        const double fl = floor(q);
        const double frac = q - fl;
        r.ok &= (frac > g.guard[a]) & (1.0 - frac > g.guard[a]) & (q < g.qmax[a]);
        r.f[a] = fl > 0.0 ? fl : 0.0;
        r.c[a] = (uint32_t)r.f[a];
    }
    return r;
}
template <typename G>
__device__ __forceinline__ uint64_t key_fast(const G &g, const CellFast &cf, bool *alias) {
    uint64_t key = 0;
    uint32_t beyond = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const uint32_t m = (uint32_t)g.mask[a];  // 32 or more bits: all ones, like the cell's zero upper half
        beyond |= cf.c[a] & ~m;
        key |= (uint64_t)(cf.c[a] & m) << g.shift[a];
    }
    *alias = beyond != 0;
    return key;
}
template <typename G>
__device__ __forceinline__ double centre_dist_fast(const G &g, const CellFast &cf, double px, double py, double pz) {
    const double cx = (cf.f[0] + 0.5) * g.cell_size + g.bmin[0];
    const double cy = (cf.f[1] + 0.5) * g.cell_size + g.bmin[1];
    const double cz = (cf.f[2] + 0.5) * g.cell_size + g.bmin[2];
    const double dx = px - cx, dy = py - cy, dz = pz - cz;
    const double a = dx * dx, b = dy * dy, c = dz * dz;
    return (a + b) + c;
}

// The partition hash of a cell key: one 64-bit multiply (Fibonacci hashing; the murmur finaliser costs two and three
// shifts, a third of what is left of a point's instructions).  Bits 63..55 pick the level-1 bin, bits 52..37 the
// second-level partition, bits 36..16 the first LDS slot; the fold of the upper half in front makes every key bit count
// in the slot bits too.
__device__ __forceinline__ uint64_t cell_hash(uint64_t k) {
    k ^= k >> 32;
    return k * 0x9e3779b97f4a7c15ull;
}
__device__ __forceinline__ uint32_t bin_of(uint64_t h) { return (uint32_t)(h >> (64 - F1_BITS)); }
// the second-level partition comes from the 16 bits under the bin bits
__device__ __forceinline__ uint32_t sel16_of(uint64_t h) { return (uint32_t)(h >> 37) & 0xffffu; }
__device__ __forceinline__ uint32_t sub_from_sel16(uint32_t sel16, uint32_t f2) { return (sel16 * f2) >> 16; }
__device__ __forceinline__ uint32_t sub_of(uint64_t h, uint32_t f2) { return sub_from_sel16(sel16_of(h), f2); }
template <int NSLOT>
__device__ __forceinline__ uint32_t slot_of(uint64_t h) { return (uint32_t)((((h >> 16) & 0x1fffffull) * NSLOT) >> 21); }


// Pointers that a kernel reads out of a table in memory are "generic" to the compiler: it emits flat loads, and
// a flat load counts on the LDS counter as well — every wait for an LDS operation (each barrier of the tile loops) would
// then also wait for the tuples in flight.  Everything here lives in global memory; these say so.
#define PCQ_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ T ldg(const T *p) {
    return *(const PCQ_GLOBAL T *)p;
}
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte access at a 4-byte aligned address
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
// (every tuple buffer ends in 64 spare bytes: the word behind a 20-byte tuple is always readable, so the second colour
// word costs a select instead of a load in a branch of its own — see k_p0_part on loads in branches)
__device__ __forceinline__ GridTuple ld_tuple(const uint8_t *p, bool wide) {
    const u32x4_a4 a = *(const PCQ_GLOBAL u32x4_a4 *)p;
    const u32x2_a4 b = *(const PCQ_GLOBAL u32x2_a4 *)(p + 16);
    GridTuple t;
    t.x = (int32_t)a.x, t.y = (int32_t)a.y, t.z = (int32_t)a.z, t.idx = a.w;
    t.w0 = wide ? b.x : b.x & 0xffffu, t.w1 = wide ? b.y : 0u, t.sel = wide ? ~0u : b.x >> 16;
    return t;
}
__device__ __forceinline__ void st_tuple(uint8_t *p, const GridTuple &t, bool wide) {
    u32x4_a4 a = {(uint32_t)t.x, (uint32_t)t.y, (uint32_t)t.z, t.idx};
    *(PCQ_GLOBAL u32x4_a4 *)p = a;
    *(PCQ_GLOBAL uint32_t *)(p + 16) = wide ? t.w0 : t.w0 | (t.sel << 16);  // (20-byte output: every input was 20 bytes, sel is there)
    if (wide) *(PCQ_GLOBAL uint32_t *)(p + 20) = t.w1;
}
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// Workgroups are dealt to the 8 XCDs round-robin (workgroup id mod 8), each XCD with an L2 of its own.  Consumers of pass 0's
// bins take them in this order, so that an XCD walks a contiguous eighth of the bins: the fragments of neighbouring bins
// are neighbours in every tile's block, and the 128-byte line two of them share is then fetched by one L2 instead of two
// (counted: the big fold fetched 1.7 x the tuples it read).  A bijection of 0 .. n for any n that is a multiple of 8.
__device__ __forceinline__ uint32_t xcd_order(uint32_t it, uint32_t n) { return n % 8 == 0 ? (it % 8) * (n / 8) + it / 8 : it; }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return (uint64_t)uni32((uint32_t)v) | ((uint64_t)uni32((uint32_t)(v >> 32)) << 32); }

// The entry table as the kernels see it: entry 0 (often the only one) travels in the kernel arguments, so that the
// common case costs no dependent global load.
struct EntryRef {
    const GridEntryDev *table;
    GridEntryDev e0;
    // (written field by field with explicit global loads: as `id == 0 ? e0 : table[id]` the compiler selects between the
    // two ADDRESSES — kernel argument segment or table — and loads six doubles through flat instructions for every tuple)
    __device__ __forceinline__ GridEntryDev get(uint32_t id) const {
        GridEntryDev e = e0;
        if (id != 0) {
            const double *src = reinterpret_cast<const double *>(table + id);
#pragma unroll
            for (int a = 0; a < 3; a++) e.scale[a] = ldg(src + a), e.offset[a] = ldg(src + 3 + a);
        }
        return e;
    }
};

struct TupleEval {
    uint64_t key, dbits;
    bool alias;
};
__device__ __forceinline__ TupleEval eval_exact(const DevGrid &g, double px, double py, double pz) {
    const CellInfo ci = cell_of(g, px, py, pz);
    TupleEval r;
    r.key = ci.key;
    r.alias = ci.alias;
    r.dbits = (uint64_t)__double_as_longlong(centre_dist(g, ci.cell, px, py, pz));
    return r;
}
// The grid as the fold kernels carry it: what the short computation reads, by value (32 scalar registers), and the whole
// grid behind a pointer for the exact computation — with DevGrid by value (53 registers) next to the other arguments the
// kernels moved scalars in and out of vector lanes a thousand times (k_fold<BIG>: 1034 v_readlane).
struct GridRef {
    DevGridFast f;
    const DevGrid *full;
};
// key, alias flag and distance bits of a world position: THE definition every pass uses (pass 0's duplicate fold must see
// the bits the fold will see).
__device__ __forceinline__ TupleEval eval_world(const GridRef &g, double px, double py, double pz) {
    const CellFast cf = cell_fast(g.f, px, py, pz);
    if (!cf.ok) return eval_exact(*g.full, px, py, pz);
    TupleEval r;
    r.key = key_fast(g.f, cf, &r.alias);
    r.dbits = (uint64_t)__double_as_longlong(centre_dist_fast(g.f, cf, px, py, pz));
    return r;
}
__device__ __forceinline__ TupleEval eval_world(const DevGrid &g, double px, double py, double pz) {
    const CellFast cf = cell_fast(g, px, py, pz);
    if (!cf.ok) return eval_exact(g, px, py, pz);
    TupleEval r;
    r.key = key_fast(g, cf, &r.alias);
    r.dbits = (uint64_t)__double_as_longlong(centre_dist_fast(g, cf, px, py, pz));
    return r;
}
template <typename G>
__device__ __forceinline__ TupleEval eval_tuple(const G &g, const EntryRef &entries, const GridTuple &t) {
    const GridEntryDev e = entries.get((t.w0 >> 8) & 0xff);
    return eval_world(g, world(t.x, e.scale[0], e.offset[0]), world(t.y, e.scale[1], e.offset[1]), world(t.z, e.scale[2], e.offset[2]));
}
// the second-level partition of a tuple: from the bits pass 0 left in it, or — a tuple with colour — from its cell again
template <typename G>
__device__ __forceinline__ uint32_t tuple_sub(const G &g, const EntryRef &entries, const GridTuple &t, uint32_t f2) {
    if (t.sel != ~0u) return sub_from_sel16(t.sel, f2);
    return sub_of(cell_hash(eval_tuple(g, entries, t).key), f2);
}
// file order among tuples: entries are numbered in scan order; 0 is reserved for an earlier fold's winner
__device__ __forceinline__ uint64_t ord_of(const GridTuple &t) { return ((uint64_t)((t.w0 >> 8) & 0xff) << 32 | t.idx) + 1; }

// The winners' records: pcq_point (31 bytes) + flag byte = 32 bytes per cell, kept as TWO arrays of 16-byte halves — {x, y}
// and {z, colour, class, flags} — so that the lanes of one store instruction (consecutive winners) write consecutive 16-byte
// words: whole lines.  As 32-byte records every store instruction wrote the even or the odd halves of its lines, and the
// memory side fetched what it was not given (counted: the dense fold fetched 6.2 GB for 3.3 GB of tuples).
struct RecArr {
    uint8_t *base;
    uint64_t cap;  // records the arrays have room for: half a of record o at base + 16 o, half b at base + 16 (cap + o)
    __host__ __device__ __forceinline__ uint4 *a(uint64_t o) const { return reinterpret_cast<uint4 *>(base + o * 16); }
    __host__ __device__ __forceinline__ uint4 *b(uint64_t o) const { return reinterpret_cast<uint4 *>(base + (cap + o) * 16); }
};
__device__ __forceinline__ uint8_t rec_flags(const uint4 &b) { return (uint8_t)(b.w >> 24); }
__device__ __forceinline__ void st_record(const RecArr &recs, uint64_t o, const GridEntryDev &e, int32_t x, int32_t y, int32_t z, uint32_t w0, uint32_t w1,
                                          uint8_t flags) {
    const uint64_t bx = (uint64_t)__double_as_longlong(world(x, e.scale[0], e.offset[0])),
                   by = (uint64_t)__double_as_longlong(world(y, e.scale[1], e.offset[1])),
                   bz = (uint64_t)__double_as_longlong(world(z, e.scale[2], e.offset[2]));
    uint4 a, b;
    a.x = (uint32_t)bx, a.y = (uint32_t)(bx >> 32), a.z = (uint32_t)by, a.w = (uint32_t)(by >> 32);
    b.x = (uint32_t)bz, b.y = (uint32_t)(bz >> 32);
    b.z = (w0 >> 16) | (w1 << 16);                                          // red, green
    b.w = (w1 >> 16) | ((w0 & 0xffu) << 16) | ((uint32_t)flags << 24);      // blue, classification, flags
    *recs.a(o) = a;
    *recs.b(o) = b;
}

// ---------------------------------------------------------------------------------------------------------------
// reading a bin of pass 0's output: a window of the bin's fragment list in LDS
// ---------------------------------------------------------------------------------------------------------------
// Window = fragments f_lo .. f_lo + nfr of bin `bin`: s_pre[0 .. nfr] (tuples of the bin in front of each, and behind the
// last), s_addr[0 .. nfr) (address of the fragment's first tuple | 1 when its tuples are 24 bytes).
__device__ __forceinline__ uint64_t frag_addr(const BinSrc &S, uint32_t bin, uint32_t f) {
    const uint64_t ta = ldg(S.tile_addr + f);
    const uint32_t st = ldg(S.startT + (size_t)bin * S.Tp + f);
    return ((ta & ~1ull) + (uint64_t)st * tuple_bytes(ta & 1)) | (ta & 1);
}
template <int NT>
__device__ __forceinline__ void frag_window_fill(const BinSrc &S, uint32_t bin, uint32_t f_lo, uint32_t nfr, uint32_t *s_pre, uint64_t *s_addr) {
    for (uint32_t t = threadIdx.x; t <= nfr; t += NT) {
        s_pre[t] = ldg(S.preT + (size_t)bin * S.Tp1 + f_lo + t);
        if (t < nfr) s_addr[t] = frag_addr(S, bin, f_lo + t);
    }
}
// The fragment of the window that holds tuple j of the bin (s_pre[0] <= j < s_pre[nfr]): the last f with s_pre[f] <= j.
// Empty fragments repeat their neighbour's value and are never the answer.
__device__ __forceinline__ uint32_t frag_find(const uint32_t *s_pre, uint32_t nfr, uint32_t j) {
    uint32_t lo = 0, hi = nfr;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_pre[mid] <= j) lo = mid;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ GridTuple frag_ld_tuple(const uint32_t *s_pre, const uint64_t *s_addr, uint32_t nfr, uint32_t j) {
    const uint32_t f = frag_find(s_pre, nfr, j);
    const uint64_t a = s_addr[f];
    const bool wide = a & 1;
    return ld_tuple(reinterpret_cast<const uint8_t *>(a & ~1ull) + (uint64_t)(j - s_pre[f]) * tuple_bytes(wide), wide);
}

// body(tuple) for every tuple of bin `bin`, some thread each, no particular order; whole workgroup, ends on a barrier.
template <int NT, int FB, int UNROLL, typename F>
__device__ __forceinline__ void bin_for_each(const BinSrc &S, uint32_t bin, uint32_t *s_pre, uint64_t *s_addr, F &&body) {
    const uint32_t total = uni32(ldg(S.preT + (size_t)bin * S.Tp1 + S.T));
    uint32_t f_lo = 0, j0 = 0;
    while (j0 < total) {  // (the same for every thread)
        const uint32_t nfr = S.T - f_lo < (uint32_t)FB ? S.T - f_lo : (uint32_t)FB;
        frag_window_fill<NT>(S, bin, f_lo, nfr, s_pre, s_addr);
        __syncthreads();
        const uint32_t wend = s_pre[nfr];
        for (uint32_t i0 = j0 + threadIdx.x; i0 < wend; i0 += NT * UNROLL) {  // the loads of UNROLL steps are issued together
            GridTuple t[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const uint32_t j = i0 + u * NT;
                t[u] = frag_ld_tuple(s_pre, s_addr, nfr, j < wend ? j : wend - 1);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
                if (i0 + u * NT < wend) body(t[u]);
        }
        j0 = wend, f_lo += nfr;
        __syncthreads();  // the window is rewritten
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pass 0: one reading of a scan's points -> per tile one block of tuples sorted by level-1 bin + a directory row
// ---------------------------------------------------------------------------------------------------------------
// What the predicate reads of a point: its position (bounds kinds) or its class byte.
template <int KIND>
struct P0In {
    RawPoint rp;
    uint32_t cls;
};
template <int KIND>
__device__ __forceinline__ P0In<KIND> p0_load(const DevCols &c, uint64_t i) {
    P0In<KIND> in;
    if (KIND == PCQ_PRED_CLASS) in.cls = c.cls[i * c.cls_stride];
    else in.rp = ld_xyz_stream(c, i);
    return in;
}
// The same for point `li` of the tile that starts at point `base` (li clamped to the tile's last point).  PACKED: the
// columns are LAST blocks — 12-byte positions at a 4-byte aligned address, one class byte per point, 6-byte colours — so
// that a point's address is the tile's (the same for the whole workgroup, scalar registers) plus a 32-bit offset, instead
// of a 64-bit multiply-add and an alignment test per point and column.
template <int KIND, bool PACKED>
__device__ __forceinline__ P0In<KIND> p0_load_tile(const DevCols &c, uint64_t base, uint32_t li, uint32_t nvalid) {
    const uint32_t lc = li < nvalid ? li : nvalid - 1;
    if (!PACKED) return p0_load<KIND>(c, base + lc);
    P0In<KIND> in;
    if (KIND == PCQ_PRED_CLASS) {
        in.cls = *(const PCQ_GLOBAL uint8_t *)(c.cls + base + lc);
    } else {
        const i32x3_a4 v = __builtin_nontemporal_load(reinterpret_cast<const i32x3_a4 *>(c.xyz + base * 12 + lc * 12u));  // (sizeof(i32x3) is 16: bytes, not elements)
        in.rp.x = v.x, in.rp.y = v.y, in.rp.z = v.z;
    }
    return in;
}
template <int KIND>
__device__ __forceinline__ bool p0_pass(const DevCols &c, const DevPred &pr, const P0In<KIND> &in) {
    if (KIND == PCQ_PRED_CLASS) return in.cls == pr.cls;
    const RawPoint &rp = in.rp;
    if (KIND == PCQ_PRED_BOUNDS)
        return (pr.empty == 0) & ((uint32_t)(rp.x - pr.lo[0]) <= pr.width[0]) & ((uint32_t)(rp.y - pr.lo[1]) <= pr.width[1]) &
               ((uint32_t)(rp.z - pr.lo[2]) <= pr.width[2]);
    const double wx = c.offset[0] + c.scale[0] * (double)rp.x, wy = c.offset[1] + c.scale[1] * (double)rp.y,
                 wz = c.offset[2] + c.scale[2] * (double)rp.z;
    return !((wx < pr.wmin[0]) | (wy < pr.wmin[1]) | (wz < pr.wmin[2]) | (wx > pr.wmax[0]) | (wy > pr.wmax[1]) | (wz > pr.wmax[2]));
}
__device__ __forceinline__ uint64_t key_only(const DevGrid &g, double px, double py, double pz) {
    const CellFast cf = cell_fast(g, px, py, pz);
    if (!cf.ok) return cell_of(g, px, py, pz).key;
    bool alias;
    return key_fast(g, cf, &alias);
}

// Workgroup w takes the tiles w, w + gridDim.x, ...: tile t = points t * 5120 .. of the scan; its matches leave as tile
// t's block (out + t * 5120 * tuple bytes: the tuples sorted by level-1 bin, written front to back as one stream) and
// directory row (dir + t * DIR_STRIDE: where each bin starts in the block; [512] = the block's tuples).  The inputs of the
// next tile are on their way while this one is sorted.
//
// The tile's own fold (`agg`).  insert_point (grid_sampling.rs:72-103) on a key's state P with a new point q, both
// measured against the centre of q's cell: P' = q if P is empty or d(q) < d(P), else P.  Take the tuples q1 .. qn of one
// key inside one tile, none of them aliased: they share one unmasked cell, hence one centre, and they are CONSECUTIVE
// among the key's tuples in file order (a tile is a range of the file).  Applying q1 .. qn to any state P gives the
// earliest qi of least distance if that distance is below d(P), else P — exactly what applying that one qi gives.  So
// the tile may drop every tuple of the key except its (distance, file order) minimum m, and any superset of {m} is as
// good; the fold downstream (and the exact replay, should the key turn out aliased elsewhere) sees an equivalent
// sequence.  A key with an aliased tuple in the tile keeps all its tuples.
// Mechanics: one 64-bit LDS word per table slot, atomicMin of (distance bits >> 13 + 1) << 13 | place in the tile — the
// truncation keeps the minimum a minimum and only lets near-ties survive together; an aliased tuple (or a distance that
// is not finite) enters as 0 << 13 | place and so wins its slot.  Afterwards a tuple reads its slot: the winner is of
// another key (compared through the tile's key array) -> kept, nothing is known; of its own key with the 0 mark -> kept;
// otherwise kept iff its truncated distance equals the winner's.  No key is stored in the table and nothing probes.
// A file in random order has no duplicates inside a tile: when a tile sheds less than a quarter of its matches the
// workgroup leaves the next 2, 4, .. 16 tiles alone before it tries again (agg_mode 0; 1 = every tile, 2 = never —
// the result is the same in every mode, only the number of tuples that travel differs).
//
// gfx950 counts loads and stores in ONE in-order counter (vmcnt): a wait for a load also waits for every store issued
// before it, and the compiler cannot count the stores of the copy-out loop — so a load must never be waited for right
// behind the copy-out.  Per tile: the attributes of this tile's matches are asked for at its head (nothing is computed on
// them until the staging); the next tile's positions were asked for before that and are waited for BEFORE this tile's
// stores are issued.  A load in a branch of its own (`c.cls ? c.cls[i] : 0` in an unrolled loop) is a serial round trip
// per point.
template <int KIND, bool RGB, bool PACKED>
__global__ __launch_bounds__(P0_NT, 4) void k_p0_part(DevCols c, DevPred pr, DevGrid g, uint32_t ntiles, uint8_t *__restrict__ out,
                                                      uint16_t *__restrict__ dir, uint32_t entry, uint64_t idx_base, int agg_mode) {
    constexpr int NT = P0_NT, ITEMS = P0_ITEMS;
    constexpr uint32_t TS = RGB ? 24 : 20;
    constexpr int STAGE_BYTES = P0_TILE * 16 + P0_TILE * 4 * (RGB ? 2 : 1);
    constexpr int AGG_BYTES = P0_TILE * 8 + AGG_SLOTS * 8;
    constexpr int RAW_BYTES = STAGE_BYTES > AGG_BYTES ? STAGE_BYTES : AGG_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[RAW_BYTES];  // the tile's sorted image; before that, the duplicate table
    uint32_t *s_img = reinterpret_cast<uint32_t *>(s_raw);                     // the block as it will lie in memory: TS bytes per tuple
    uint64_t *s_akey = reinterpret_cast<uint64_t *>(s_raw);                    // the cell key of every place in the tile
    uint64_t *s_atab = s_akey + P0_TILE;                                       // the table
    __shared__ uint32_t s_cnt[F1], s_base[F1 + 1], s_npass[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t t = tid; t < F1; t += NT) s_cnt[t] = 0;
    if (tid < 2) s_npass[tid] = 0;
    __syncthreads();
    uint32_t agg_skip = 0, agg_backoff = 1, parity = 0;
    uint32_t tile = blockIdx.x;
    P0In<KIND> cur[ITEMS], nxt[ITEMS];
    auto tile_points = [&](uint32_t t) {  // points of tile t (the last one may be short)
        const uint64_t left = c.n - (uint64_t)t * P0_TILE;
        return left < (uint64_t)P0_TILE ? (uint32_t)left : (uint32_t)P0_TILE;
    };
    if (tile < ntiles) {
        const uint32_t nv = tile_points(tile);
#pragma unroll
        for (int j = 0; j < ITEMS; j++) cur[j] = p0_load_tile<KIND, PACKED>(c, (uint64_t)tile * P0_TILE, (uint32_t)j * NT + tid, nv);
    }
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {  // (arrived: inside the loop nothing is pending at its head)
        if (KIND == PCQ_PRED_CLASS) asm volatile("" ::"v"(cur[j].cls));
        else asm volatile("" ::"v"(cur[j].rp.x), "v"(cur[j].rp.y), "v"(cur[j].rp.z));
    }
    for (; tile < ntiles; tile += gridDim.x, parity ^= 1) {
        const uint64_t base = (uint64_t)tile * P0_TILE;
        const uint32_t ntile = tile + gridDim.x;
        const uint32_t nvalid = tile_points(tile);
        if (ntile < ntiles) {
            const uint32_t nv = tile_points(ntile);
#pragma unroll
            for (int j = 0; j < ITEMS; j++) nxt[j] = p0_load_tile<KIND, PACKED>(c, (uint64_t)ntile * P0_TILE, (uint32_t)j * NT + tid, nv);
        }
        const bool agg = agg_mode == 1 || (agg_mode == 0 && agg_skip == 0);  // (the same for the whole workgroup)
        bool passes[ITEMS];
        uint32_t metas[ITEMS], ranks[ITEMS], rg[ITEMS], bb[ITEMS], cl[ITEMS];
        uint64_t pk[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const uint32_t li = (uint32_t)j * NT + tid;
            const uint64_t i = base + li;
            passes[j] = li < nvalid && p0_pass<KIND>(c, pr, cur[j]);
            rg[j] = 0, bb[j] = 0, cl[j] = KIND == PCQ_PRED_CLASS ? cur[j].cls : 0;
            pk[j] = 0, metas[j] = 0, ranks[j] = 0;
            if (!passes[j]) continue;
            if (KIND == PCQ_PRED_CLASS) {
                if (PACKED) {
                    const i32x3_a4 v = *(const PCQ_GLOBAL i32x3_a4 *)(c.xyz + base * 12 + li * 12u);
                    cur[j].rp.x = v.x, cur[j].rp.y = v.y, cur[j].rp.z = v.z;
                } else {
                    cur[j].rp = ld_xyz(c, i);
                }
            }
            if (RGB) {  // last.rs:145-153
                const uint8_t *q = PACKED ? c.rgb + base * 6 + li * 6u : c.rgb + i * c.rgb_stride;
                rg[j] = ld_u16(q) | (ld_u16(q + 2) << 16);
                bb[j] = ld_u16(q + 4);
            }
            if (KIND != PCQ_PRED_CLASS && c.cls) cl[j] = PACKED ? *(const PCQ_GLOBAL uint8_t *)(c.cls + base + li) : c.cls[i * c.cls_stride];  // last.rs:138-142
        }
        if (agg)
            for (uint32_t k = tid; k < (uint32_t)AGG_SLOTS; k += NT) s_atab[k] = ~0ull;
        uint32_t npass = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            if (!passes[j]) continue;
            npass++;
            const double px = world(cur[j].rp.x, c.scale[0], c.offset[0]), py = world(cur[j].rp.y, c.scale[1], c.offset[1]),
                         pz = world(cur[j].rp.z, c.scale[2], c.offset[2]);
            if (agg) {
                const TupleEval ev = eval_world(g, px, py, pz);
                const uint64_t h = cell_hash(ev.key);
                const uint32_t place = (uint32_t)j * NT + tid;
                s_akey[place] = ev.key;
                const bool through = ev.alias || ev.dbits >= 0x7ff0000000000000ull;
                pk[j] = (through ? 0ull : ((ev.dbits >> AGG_POS_BITS) + 1) << AGG_POS_BITS) | place;
                metas[j] = bin_of(h) | (sel16_of(h) << 16);  // (the bin, and the second level's selector for the red half of a colourless w0)
                ranks[j] = (uint32_t)(h >> 24) & (AGG_SLOTS - 1);  // the table slot, until the tile's fold is over
            } else {
                const uint64_t h = cell_hash(key_only(g, px, py, pz));
                metas[j] = bin_of(h) | (sel16_of(h) << 16);
            }
        }
        if (agg) {
            __syncthreads();  // the table is clear, the keys are in place
#pragma unroll
            for (int j = 0; j < ITEMS; j++)
                if (passes[j]) atomicMin((unsigned long long *)&s_atab[ranks[j]], (unsigned long long)pk[j]);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                if (!passes[j]) continue;
                const uint64_t w = s_atab[ranks[j]];
                const uint64_t wkey = s_akey[(uint32_t)w & ((1u << AGG_POS_BITS) - 1)], mykey = s_akey[(uint32_t)j * NT + tid];
                passes[j] = wkey != mykey || (w >> AGG_POS_BITS) == 0 || (w >> AGG_POS_BITS) == (pk[j] >> AGG_POS_BITS);
            }
        }
#pragma unroll
        for (int j = 0; j < ITEMS; j++)
            if (passes[j]) ranks[j] = atomicAdd(&s_cnt[metas[j] & (F1 - 1)], 1u);
        {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) npass += __shfl_xor(npass, o, 64);
            if (lane == 0 && npass) atomicAdd(&s_npass[parity], npass);
        }
        __syncthreads();
        if (wave == 0) {  // exclusive scan of the tile's counts over the bins, by ONE wave (lane l = bins 8 l .. 8 l + 7): no barrier inside;
                          // the counters are cleared for the next tile
            constexpr int BPL = F1 / 64;
            uint32_t v[BPL], mine = 0;
#pragma unroll
            for (int q = 0; q < BPL; q++) v[q] = s_cnt[lane * BPL + q], mine += v[q];
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if (lane >= (uint32_t)off) incl += up;
            }
            uint32_t before = incl - mine;
#pragma unroll
            for (int q = 0; q < BPL; q++) {
                s_base[lane * BPL + q] = before;
                s_cnt[lane * BPL + q] = 0;
                before += v[q];
            }
            if (lane == 63) s_base[F1] = incl;
        }
        __syncthreads();
        const uint32_t total = s_base[F1], matched = s_npass[parity];
        if (tid < (uint32_t)DIR_WORDS) {  // the directory row, two entries per word
            const uint32_t lo = s_base[2 * tid], hi = 2 * tid + 1 <= (uint32_t)F1 ? s_base[2 * tid + 1] : 0;
            *(PCQ_GLOBAL uint32_t *)(reinterpret_cast<uint32_t *>(dir + (size_t)tile * DIR_STRIDE) + tid) = lo | (hi << 16);
        }
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            if (!passes[j]) continue;
            const uint32_t at = s_base[metas[j] & (F1 - 1)] + ranks[j];
            uint32_t *q = s_img + at * (TS / 4);  // (5 or 6 words per tuple: an odd stride, or two-way conflicts — the LDS is not what this kernel waits for)
            q[0] = (uint32_t)cur[j].rp.x, q[1] = (uint32_t)cur[j].rp.y, q[2] = (uint32_t)cur[j].rp.z;
            q[3] = (uint32_t)(idx_base + base) + (uint32_t)j * NT + tid;
            q[4] = cl[j] | (entry << 8) | (RGB ? rg[j] << 16 : metas[j] & 0xffff0000u);
            if (RGB) q[5] = (rg[j] >> 16) | (bb[j] << 16);
        }
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {  // the next tile's inputs have arrived (asked for a whole tile ago) — before the stores below
            cur[j] = nxt[j];
            if (KIND == PCQ_PRED_CLASS) asm volatile("" ::"v"(cur[j].cls));
            else asm volatile("" ::"v"(cur[j].rp.x), "v"(cur[j].rp.y), "v"(cur[j].rp.z));
        }
        if (tid == 0) s_npass[parity ^ 1] = 0;
        __syncthreads();
        // The block, front to back, as whole 16-byte words of the image (a block starts 16-byte aligned and has room for whole
        // words): every store instruction of a wave is 1 KiB without a gap.  (Stored tuple by tuple — 16 + 4 bytes at a stride
        // of 20 — the same bytes left as twice the instructions with holes for the other one to fill.)
        uint4 *blk = reinterpret_cast<uint4 *>(out + (uint64_t)tile * P0_TILE * TS);
        const uint32_t nq = (total * TS + 15) / 16;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        for (uint32_t t = tid; t < nq; t += NT) *(PCQ_GLOBAL u32x4 *)(blk + t) = reinterpret_cast<const u32x4 *>(s_img)[t];
        if (agg && agg_mode == 0) {
            if (total * 4 > matched * 3) {  // less than a quarter shed: not worth the table for a while
                agg_backoff = agg_backoff < 16 ? agg_backoff * 2 : 16;
                agg_skip = agg_backoff;
            } else {
                agg_backoff = 1;
            }
        } else if (agg_skip) {
            agg_skip--;
        }
        // The next tile rewrites the image.  Without the tile's table that happens behind its first two barriers — no thread
        // gets there before every thread has left this copy-out —, so only a tile that starts with the table (clear + key
        // array, in the image's LDS) needs a barrier here: three barriers per tile instead of four (five at the start of the
        // round: 1.18 -> 1.10 -> see profiles/r03_grid_progress.txt).
        if (agg_mode == 1 || (agg_mode == 0 && agg_skip == 0)) __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fold preparation: the directory rows, transposed into per-bin fragment lists
// ---------------------------------------------------------------------------------------------------------------
struct DevRun {        // one pending pass-0 run
    const uint8_t *tuples;
    const uint16_t *dir;
    uint32_t tile0;    // its first tile among all pending tiles
    uint32_t ntiles;
    uint32_t wide, _pad;
};

// 64 tiles per workgroup: their directory rows through LDS; lane = tile, so that what leaves are whole lines of
// startT[b] / preT[b] (preT gets the fragment's COUNT here; k_bin_prefix turns the counts into the prefix).
__global__ __launch_bounds__(BLOCK) void k_dir_transpose(const DevRun *__restrict__ runs, int nruns, uint32_t T, uint32_t Tp, uint32_t Tp1,
                                                         uint16_t *__restrict__ startT, uint32_t *__restrict__ preT, uint64_t *__restrict__ tile_addr) {
    __shared__ uint32_t s_rows[64 * DIR_WORDS];  // 64 rows of 257 words: lane r reads word r * 257 + k — no two lanes in one bank
    __shared__ uint64_t s_rowptr[64];
    const uint32_t t0 = blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const uint32_t t = t0 + threadIdx.x;
        uint64_t rowptr = 0;
        if (t < T) {
            int lo = 0, hi = nruns;  // the last run with tile0 <= t
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (runs[mid].tile0 <= t) lo = mid;
                else hi = mid;
            }
            const DevRun r = runs[lo];
            const uint32_t local = t - r.tile0;
            rowptr = reinterpret_cast<uint64_t>(r.dir + (size_t)local * DIR_STRIDE);
            tile_addr[t] = reinterpret_cast<uint64_t>(r.tuples + (uint64_t)local * P0_TILE * tuple_bytes(r.wide)) | (r.wide ? 1u : 0u);
        }
        s_rowptr[threadIdx.x] = rowptr;
    }
    __syncthreads();
    for (int r = 0; r < 64; r++) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(s_rowptr[r]);
        for (int w = threadIdx.x; w < DIR_WORDS; w += BLOCK) s_rows[r * DIR_WORDS + w] = row ? ldg(row + w) : 0u;
    }
    __syncthreads();
    const uint32_t r = threadIdx.x & 63;
    if (t0 + r >= T) return;
    for (uint32_t b = threadIdx.x >> 6; b < (uint32_t)F1; b += WAVES) {
        const uint32_t w0 = s_rows[r * DIR_WORDS + (b >> 1)], w1 = s_rows[r * DIR_WORDS + ((b + 1) >> 1)];
        const uint32_t v0 = (b & 1) ? w0 >> 16 : w0 & 0xffffu, v1 = ((b + 1) & 1) ? w1 >> 16 : w1 & 0xffffu;
        startT[(size_t)b * Tp + t0 + r] = (uint16_t)v0;
        preT[(size_t)b * Tp1 + t0 + r] = v1 - v0;
    }
}

// preT[b][0 .. T): counts -> exclusive prefix, preT[b][T] = bintot[b] = the bin's tuples.  One workgroup per bin.
__global__ __launch_bounds__(1024) void k_bin_prefix(uint32_t *__restrict__ preT, uint32_t T, uint32_t Tp1, uint32_t *__restrict__ bintot) {
    __shared__ uint32_t s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *row = preT + (size_t)blockIdx.x * Tp1;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < T; c0 += 4096) {
        const uint32_t i0 = c0 + threadIdx.x * 4;
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = i0 + k < T ? row[i0 + k] : 0;
            sum += v[k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t run = carry + incl - sum, total = 0;
        for (int w = 0; w < 16; w++) {
            run += w < wave ? s_wave[w] : 0;
            total += s_wave[w];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < T) row[i0 + k] = run;
            run += v[k];
        }
        carry += total;
        __syncthreads();  // s_wave is rewritten
    }
    if (threadIdx.x == 0) {
        row[T] = carry;
        bintot[blockIdx.x] = carry;
    }
}

// Short fragments (a scan whose tiles shed most of their tuples, a box that few points match): the reader's window
// would hold a handful of tuples per round.  Then the bins are first copied together — one thread per fragment, comp =
// bin 0's tuples, bin 1's, ... — and described to the fold as pass-0 output of F1 "tiles", tile t = bin t's piece:
// fragment (b, t) is empty unless t == b.  Every consumer reads that through the same window code.
__global__ __launch_bounds__(BLOCK) void k_bin_compact(BinSrc S, const uint32_t *__restrict__ binbase, uint8_t *__restrict__ comp, uint32_t wide_out) {
    const uint64_t q = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= (uint64_t)S.T * F1) return;
    const uint32_t bin = (uint32_t)(q / S.T), f = (uint32_t)(q % S.T);
    const uint32_t lo = ldg(S.preT + (size_t)bin * S.Tp1 + f), hi = ldg(S.preT + (size_t)bin * S.Tp1 + f + 1);
    if (lo == hi) return;
    const uint64_t a = frag_addr(S, bin, f);
    const bool wide = a & 1;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(a & ~1ull);
    uint8_t *dst = comp + (uint64_t)(ldg(binbase + bin) + lo) * tuple_bytes(wide_out);
    for (uint32_t i = 0; i < hi - lo; i++) st_tuple(dst + (uint64_t)i * tuple_bytes(wide_out), ld_tuple(src + (uint64_t)i * tuple_bytes(wide), wide), wide_out);
}
// the directory of the compacted bins: preT[b][t] = (t <= b ? 0 : the bin's tuples), startT = 0, tile_addr[t] = bin t's piece
__global__ __launch_bounds__(BLOCK) void k_compact_dir(const uint32_t *__restrict__ binbase, const uint8_t *__restrict__ comp, uint32_t wide, uint32_t Tp1,
                                                       uint32_t Tp, uint32_t *__restrict__ preT, uint16_t *__restrict__ startT, uint64_t *__restrict__ tile_addr) {
    const uint32_t b = blockIdx.x, cnt = binbase[b + 1] - binbase[b];
    for (uint32_t t = threadIdx.x; t <= (uint32_t)F1; t += BLOCK) {
        preT[(size_t)b * Tp1 + t] = t <= b ? 0u : cnt;
        if (t < (uint32_t)F1) startT[(size_t)b * Tp + t] = 0;
        if (b == 0 && t < (uint32_t)F1) tile_addr[t] = reinterpret_cast<uint64_t>(comp + (uint64_t)binbase[t] * tuple_bytes(wide)) | (wide ? 1u : 0u);
    }
}

// tot[p] = tuples of partition p of the second level's output.
__global__ __launch_bounds__(BLOCK) void k_part_totals(GridSeg sg, uint32_t nparts, uint32_t *__restrict__ tot) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= nparts) return;
    tot[p] = sg.cnt ? sg.cnt[p] : sg.off[p + 1] - sg.off[p];
}

// out[0..n] = exclusive prefix of in[0..n) (out[n] = the sum); one workgroup, any n.
__global__ __launch_bounds__(1024) void k_excl_scan_u32(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = (n + 1023) / 1024;  // a contiguous piece per thread
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += in[i];
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += s_wave[w];
    uint32_t run = wave_off + incl - sum;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = in[i];
        out[i] = run;
        run += v;
    }
    if (threadIdx.x == 1023) s_carry = wave_off + incl;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = s_carry;
}

__global__ __launch_bounds__(1024) void k_excl_scan_u64(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t n) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += in[i];
    uint64_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t up = __shfl_up((unsigned long long)incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += s_wave[w];
    uint64_t run = wave_off + incl - sum;
    for (uint32_t i = lo; i < hi; i++) {
        const uint64_t v = in[i];
        out[i] = run;
        run += v;
    }
    if (threadIdx.x == 1023) s_carry = wave_off + incl;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = s_carry;
}

// Room for the winners of partition p: never more than its inputs, never more than the LDS table holds.
__global__ __launch_bounds__(BLOCK) void k_winner_room(const uint32_t *__restrict__ tot, const uint32_t *__restrict__ ocount, uint32_t nparts,
                                                       uint32_t limit, uint64_t *__restrict__ room) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= nparts) return;
    const uint64_t in = (uint64_t)tot[p] + (ocount ? ocount[p] : 0);
    room[p] = in < limit ? in : limit;
}

// Exclusive prefix of up to a few hundred thousand u64 in three small launches (a single workgroup walking 200 k
// partitions took 0.2 ms): sums of 4096-element pieces, their prefix, the pieces again.
constexpr int SCAN_PIECE = 4096;
__global__ __launch_bounds__(1024) void k_scan_piece_sums(const uint64_t *__restrict__ in, uint32_t n, uint64_t *__restrict__ sums) {
    __shared__ uint64_t s_wave[16];
    const uint32_t base = blockIdx.x * SCAN_PIECE;
    uint64_t v = 0;
    for (int k = 0; k < SCAN_PIECE / 1024; k++) {
        const uint32_t i = base + k * 1024 + threadIdx.x;
        v += i < n ? in[i] : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down((unsigned long long)v, off, 64);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < 16; w++) t += s_wave[w];
        sums[blockIdx.x] = t;
    }
}
// out[i] = piece_prefix[piece] + exclusive prefix inside the piece; thread t owns 4 consecutive elements; out[n] = total
__global__ __launch_bounds__(1024) void k_scan_pieces(const uint64_t *__restrict__ in, uint32_t n, const uint64_t *__restrict__ piece_prefix,
                                                      uint64_t *__restrict__ out) {
    __shared__ uint64_t s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i0 = blockIdx.x * SCAN_PIECE + threadIdx.x * 4;
    uint64_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = i0 + k < n ? in[i0 + k] : 0;
        sum += v[k];
    }
    uint64_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t up = __shfl_up((unsigned long long)incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t run = piece_prefix[blockIdx.x] + incl - sum;
    for (int w = 0; w < wave; w++) run += s_wave[w];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (i0 + k < n) out[i0 + k] = run;
        run += v[k];
        if (i0 + k + 1 == n) out[n] = run;
    }
}

// Distinct cells among the tuples of the first PROBE_BINS level-1 bins (a global hash set; one thread per fragment).
__global__ __launch_bounds__(BLOCK) void k_probe_distinct(BinSrc S, EntryRef entries, DevGrid g, uint64_t *__restrict__ set, uint64_t mask,
                                                          unsigned long long *__restrict__ distinct) {
    uint32_t mine = 0;
    const uint32_t nfrag = S.T * PROBE_BINS;
    for (uint32_t q = blockIdx.x * BLOCK + threadIdx.x; q < nfrag; q += gridDim.x * BLOCK) {
        const uint32_t bin = q / S.T, f = q % S.T;
        const uint32_t lo = ldg(S.preT + (size_t)bin * S.Tp1 + f), hi = ldg(S.preT + (size_t)bin * S.Tp1 + f + 1);
        if (lo == hi) continue;
        const uint64_t a = frag_addr(S, bin, f);
        const bool wide = a & 1;
        const uint8_t *p = reinterpret_cast<const uint8_t *>(a & ~1ull);
        for (uint32_t i = 0; i < hi - lo; i++) {
            const GridTuple t = ld_tuple(p + (uint64_t)i * tuple_bytes(wide), wide);
            const uint64_t key = eval_tuple(g, entries, t).key;
            uint64_t h = hash64(key) & mask;
            bool is_new = true;  // (the set is sized from an estimate: after 64 probes in a crowded one a key counts as new — an estimate either way)
            for (int probes = 0; probes < 64; probes++) {
                const uint64_t k = __hip_atomic_load(&set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k == key) {
                    is_new = false;
                    break;
                }
                if (k == PCQ_EMPTY_KEY) {
                    const uint64_t prev = atomicCAS((unsigned long long *)&set[h], (unsigned long long)PCQ_EMPTY_KEY, (unsigned long long)key);
                    if (prev == PCQ_EMPTY_KEY) break;
                    if (prev == key) {
                        is_new = false;
                        break;
                    }
                }
                h = (h + 1) & mask;
            }
            mine += is_new ? 1 : 0;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(distinct, (unsigned long long)mine);
}

// ---------------------------------------------------------------------------------------------------------------
// second partition level: one workgroup per level-1 bin cuts the bin's tuples (and, when the fan-out changes, the
// earlier winners of the bin) into f2 partitions by the next bits of hash(key).
// ---------------------------------------------------------------------------------------------------------------
struct Level2Params {
    BinSrc src;
    EntryRef entries;
    DevGrid g;
    uint32_t f2;
    const uint32_t *binbase;  // [F1 + 1] tuples in front of each bin; nullptr: tuples are not moved
    uint8_t *out;
    uint32_t wide;            // the output's tuples are 24 bytes (some run carries colour)
    uint32_t *off2;           // [F1 * f2 + 1]
    uint32_t *cnt2;           // k_level2: [F1 * f2] tuples per sub-partition; cap = the room each of them has
    uint32_t cap;
    unsigned long long *stats;  // k_level2: [5] += 1 when a sub-partition outgrew its room
    // earlier winners, re-cut from f2old partitions per bin into f2 (nullptr: not moved)
    const uint64_t *okeys;
    RecArr orecs;
    const uint64_t *obase;    // [F1 * f2old + 1]
    const uint32_t *ocount;   // [F1 * f2old]
    uint32_t f2old;
    const uint32_t *obinbase; // [F1 + 1] earlier winners in front of each bin
    uint64_t *okeys2;
    RecArr orecs2;
    uint32_t *ooff2;          // [F1 * f2 + 1]
};

// The exact form: a histogram pass over the bin's tuples, then the scatter, sub-partitions back to back (off2 only).  Both
// passes compute the tuple's cell.  For more than 1024 sub-partitions per bin, and when k_level2's regions did not hold.
__global__ __launch_bounds__(L2_NT) void k_level2_direct(Level2Params P) {
    __shared__ uint32_t s_hist[F2_MAX], s_cur[F2_MAX], s_ohist[F2_MAX], s_ocur[F2_MAX];
    __shared__ uint32_t s_pre[L2_FB + 1];
    __shared__ uint64_t s_addr[L2_FB];
    const uint32_t bin = xcd_order(blockIdx.x, F1), f2 = P.f2;
    for (uint32_t t = threadIdx.x; t < F2_MAX; t += L2_NT) s_hist[t] = 0, s_ohist[t] = 0;
    __syncthreads();
    if (P.binbase)
        bin_for_each<L2_NT, L2_FB, L2_UNROLL>(P.src, bin, s_pre, s_addr, [&](const GridTuple &t) {
            atomicAdd(&s_hist[tuple_sub(P.g, P.entries, t, f2)], 1u);
        });
    if (P.okeys)
        for (uint32_t q = bin * P.f2old; q < (bin + 1) * P.f2old; q++) {
            const uint64_t base = P.obase[q];
            const uint32_t n = P.ocount[q];
            for (uint32_t i = threadIdx.x; i < n; i += L2_NT) atomicAdd(&s_ohist[sub_of(cell_hash(P.okeys[base + i]), f2)], 1u);
        }
    __syncthreads();
    if (threadIdx.x == 0) {  // a serial prefix is a few hundred to a few thousand LDS reads
        uint32_t run = P.binbase ? P.binbase[bin] : 0, orun = P.okeys ? P.obinbase[bin] : 0;
        for (uint32_t s = 0; s < f2; s++) {
            if (P.binbase) P.off2[bin * f2 + s] = run;
            s_cur[s] = run;
            run += s_hist[s];
            if (P.okeys) P.ooff2[bin * f2 + s] = orun;
            s_ocur[s] = orun;
            orun += s_ohist[s];
        }
        if (bin == F1 - 1) {
            if (P.binbase) P.off2[F1 * f2] = run;
            if (P.okeys) P.ooff2[F1 * f2] = orun;
        }
    }
    __syncthreads();
    if (P.binbase) {
        const bool wide = P.wide;
        bin_for_each<L2_NT, L2_FB, L2_UNROLL>(P.src, bin, s_pre, s_addr, [&](const GridTuple &t) {
            const uint32_t pos = atomicAdd(&s_cur[tuple_sub(P.g, P.entries, t, f2)], 1u);
            st_tuple(P.out + (uint64_t)pos * tuple_bytes(wide), t, wide);
        });
    }
    if (P.okeys)
        for (uint32_t q = bin * P.f2old; q < (bin + 1) * P.f2old; q++) {
            const uint64_t base = P.obase[q];
            const uint32_t n = P.ocount[q];
            for (uint32_t i = threadIdx.x; i < n; i += L2_NT) {
                const uint64_t key = P.okeys[base + i];
                const uint32_t pos = atomicAdd(&s_ocur[sub_of(cell_hash(key), f2)], 1u);
                P.okeys2[pos] = key;
                *P.orecs2.a(pos) = *P.orecs.a(base + i);
                *P.orecs2.b(pos) = *P.orecs.b(base + i);
            }
        }
}

// The staged form (f2 <= 1024), ONE pass over the bin's tuples: a tile of 4096 tuples is sorted by sub-partition in LDS and
// leaves as runs (the direct form's scattered stores reached HBM as 6.4 GB for 3.9 GB of tuples).  There is no histogram
// pass in front: sub-partition p owns the fixed region out[p * cap .. (p + 1) * cap), cap = 1.3 x the mean partition + 64
// — the cell keys are hashed, a partition's tuple count is the mean +- a few per cent unless single cells hold hundreds
// of points — and reports off2[p] = p * cap, cnt2[p] = its tuples.  A partition that outgrows its region raises stats[5]:
// the host then takes the exact form (k_level2_direct), which counts first.
// The bin arrives through the fragment reader: a window of 2048 fragments (about five tiles' worth of tuples) in LDS,
// whole tiles out of it — the next window starts at the fragment the last whole tile ended in.
__global__ __launch_bounds__(L2S_NT) void k_level2(Level2Params P) {
    constexpr int BPT = L2_STAGED_F2 / L2S_NT;  // sub-partitions per thread when the cursors move on
    static_assert(BPT >= 1 && BPT * L2S_NT == L2_STAGED_F2, "whole sub-partitions per thread");
    __shared__ uint32_t s_cur[L2_STAGED_F2], s_ohist[L2_STAGED_F2], s_ocur[L2_STAGED_F2];
    __shared__ uint32_t s_cnt[L2_STAGED_F2], s_base[L2_STAGED_F2];
    __shared__ uint4 s_xyzi[L2S_TILE];   // the tile's tuples, sorted by sub-partition: x, y, z, idx
    __shared__ uint2 s_attr[L2S_TILE];   //                                                 w0, w1
    __shared__ uint32_t s_tpos[L2S_TILE];
    __shared__ uint32_t s_pre[L2S_FB + 1];
    __shared__ uint64_t s_addr[L2S_FB];
    __shared__ uint32_t s_total, s_overflow;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t bin = xcd_order(blockIdx.x, F1), f2 = P.f2, cap = P.cap;
    for (uint32_t t = threadIdx.x; t < L2_STAGED_F2; t += L2S_NT) s_cur[t] = 0, s_ohist[t] = 0, s_cnt[t] = 0;
    if (threadIdx.x == 0) s_overflow = 0;
    __syncthreads();
    if (P.okeys)
        for (uint32_t q = bin * P.f2old; q < (bin + 1) * P.f2old; q++) {
            const uint64_t base = P.obase[q];
            const uint32_t n = P.ocount[q];
            for (uint32_t i = threadIdx.x; i < n; i += L2S_NT) atomicAdd(&s_ohist[sub_of(cell_hash(P.okeys[base + i]), f2)], 1u);
        }
    __syncthreads();
    if (threadIdx.x == 0 && P.okeys) {  // f2 <= 1024: a serial prefix is a thousand LDS reads
        uint32_t orun = P.obinbase[bin];
        for (uint32_t s = 0; s < f2; s++) {
            P.ooff2[bin * f2 + s] = orun;
            s_ocur[s] = orun;
            orun += s_ohist[s];
        }
        if (bin == F1 - 1) P.ooff2[F1 * f2] = orun;
    }
    __syncthreads();
    if (P.out) {
        const BinSrc &S = P.src;
        const bool wide_out = P.wide;
        const uint32_t ts_out = tuple_bytes(wide_out);
        const uint64_t region0 = (uint64_t)bin * f2 * cap;  // this bin's sub-partition s: out[region0 + s * cap ...)
        const uint32_t total_in = uni32(ldg(S.preT + (size_t)bin * S.Tp1 + S.T));
        uint32_t f_lo = 0, j0 = 0;
        while (j0 < total_in) {  // one window of the bin's fragment list per round (the same for every thread)
            const uint32_t nfr = S.T - f_lo < (uint32_t)L2S_FB ? S.T - f_lo : (uint32_t)L2S_FB;
            frag_window_fill<L2S_NT>(S, bin, f_lo, nfr, s_pre, s_addr);
            __syncthreads();
            const uint32_t wend = s_pre[nfr];
            // whole tiles out of the window; what is left starts the next window — unless the window ends the bin or holds
            // less than a tile (a sparse bin), then everything
            uint32_t hi = wend;
            if (wend != total_in && wend - j0 >= (uint32_t)L2S_TILE) hi = j0 + (wend - j0) / L2S_TILE * L2S_TILE;
            if (hi > j0) {
                GridTuple t[L2S_ITEMS], tn[L2S_ITEMS];
#pragma unroll
                for (int j = 0; j < L2S_ITEMS; j++) {
                    const uint32_t i = j0 + j * L2S_NT + threadIdx.x;
                    t[j] = frag_ld_tuple(s_pre, s_addr, nfr, i < hi ? i : hi - 1);
                }
#pragma unroll
                for (int j = 0; j < L2S_ITEMS; j++)  // (arrived: see k_p0_part on the one counter for loads and stores)
                    asm volatile("" ::"v"(t[j].x), "v"(t[j].w0), "v"(t[j].w1));
                for (uint32_t base = j0; base < hi; base += L2S_TILE) {
                    if (base + L2S_TILE < hi) {  // the next tile is on its way while this one is sorted
#pragma unroll
                        for (int j = 0; j < L2S_ITEMS; j++) {
                            const uint32_t i = base + L2S_TILE + j * L2S_NT + threadIdx.x;
                            tn[j] = frag_ld_tuple(s_pre, s_addr, nfr, i < hi ? i : hi - 1);
                        }
                    }
                    uint32_t subs[L2S_ITEMS], ranks[L2S_ITEMS];
                    bool valid[L2S_ITEMS];
#pragma unroll
                    for (int j = 0; j < L2S_ITEMS; j++) {
                        valid[j] = base + j * L2S_NT + threadIdx.x < hi;
                        subs[j] = tuple_sub(P.g, P.entries, t[j], f2);
                        ranks[j] = 0;
                        if (valid[j]) ranks[j] = atomicAdd(&s_cnt[subs[j]], 1u);
                    }
                    __syncthreads();
                    if (wave == 0) {  // exclusive scan of the tile's counts over the sub-partitions, by ONE wave (lane l = entries 16 l ..): no barrier inside
                        constexpr int BPL = L2_STAGED_F2 / 64;
                        uint32_t v[BPL], mine = 0;
#pragma unroll
                        for (int q = 0; q < BPL; q++) v[q] = s_cnt[lane * BPL + q], mine += v[q];
                        uint32_t incl = mine;
#pragma unroll
                        for (int off = 1; off < 64; off <<= 1) {
                            const uint32_t up = __shfl_up(incl, off, 64);
                            if (lane >= off) incl += up;
                        }
                        uint32_t before = incl - mine;
#pragma unroll
                        for (int q = 0; q < BPL; q++) {
                            s_base[lane * BPL + q] = before;
                            s_cnt[lane * BPL + q] = 0;
                            before += v[q];
                        }
                        if (lane == 63) s_total = incl;
                    }
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < L2S_ITEMS; j++) {
                        if (!valid[j]) continue;
                        const uint32_t at = s_base[subs[j]] + ranks[j];
                        s_xyzi[at] = make_uint4((uint32_t)t[j].x, (uint32_t)t[j].y, (uint32_t)t[j].z, t[j].idx);
                        s_attr[at] = make_uint2(t[j].w0, t[j].w1);
                        const uint32_t within = s_cur[subs[j]] + ranks[j];  // place in the sub-partition's region
                        s_tpos[at] = within < cap ? subs[j] * cap + within : 0xffffffffu;
                    }
#pragma unroll
                    for (int j = 0; j < L2S_ITEMS; j++) {  // the next tile has arrived — before this tile's stores are issued
                        t[j] = tn[j];
                        asm volatile("" ::"v"(t[j].x), "v"(t[j].w0), "v"(t[j].w1));
                    }
                    __syncthreads();
                    {  // the cursors move on (a thread's sub-partitions: their tile counts are s_base differences)
                        const uint32_t s0 = threadIdx.x * BPT, total = s_total;
#pragma unroll
                        for (int q = 0; q < BPT; q++) {
                            const uint32_t lo_b = s_base[s0 + q], hi_b = s0 + q + 1 < L2_STAGED_F2 ? s_base[s0 + q + 1] : total;
                            s_cur[s0 + q] += hi_b - lo_b;
                            if (s_cur[s0 + q] > cap) s_overflow = 1;
                        }
                        for (uint32_t k = threadIdx.x; k < total; k += L2S_NT) {
                            const uint32_t tp = s_tpos[k];
                            if (tp == 0xffffffffu) continue;  // beyond the region: the fold's result will not be used
                            const uint4 a = s_xyzi[k];
                            const uint2 b = s_attr[k];
                            uint8_t *q = P.out + (region0 + tp) * ts_out;
                            u32x4_a4 va = {a.x, a.y, a.z, a.w};
                            *(PCQ_GLOBAL u32x4_a4 *)q = va;
                            *(PCQ_GLOBAL uint32_t *)(q + 16) = b.x;
                            if (wide_out) *(PCQ_GLOBAL uint32_t *)(q + 20) = b.y;
                        }
                    }
                    // (no barrier at the end of a tile: the next tile writes the sorted image and the cursors only behind its
                    // first two barriers, and nobody passes those before everybody has left this copy-out)
                }
            }
            // the next window: at the fragment tuple `hi` of the bin lies in
            if (hi == wend) f_lo += nfr;
            else f_lo += frag_find(s_pre, nfr, hi);
            j0 = hi;
            __syncthreads();  // the window is rewritten
        }
        for (uint32_t sp = threadIdx.x; sp < f2; sp += L2S_NT) {
            const uint32_t n = s_cur[sp];
            P.off2[bin * f2 + sp] = (uint32_t)(region0 + (uint64_t)sp * cap);
            P.cnt2[bin * f2 + sp] = n < cap ? n : cap;
        }
        if (threadIdx.x == 0 && s_overflow) atomicAdd(&P.stats[5], 1ull);
    }
    if (P.okeys)
        for (uint32_t q = bin * P.f2old; q < (bin + 1) * P.f2old; q++) {
            const uint64_t base = P.obase[q];
            const uint32_t n = P.ocount[q];
            for (uint32_t i = threadIdx.x; i < n; i += L2S_NT) {
                const uint64_t key = P.okeys[base + i];
                const uint32_t pos = atomicAdd(&s_ocur[sub_of(cell_hash(key), f2)], 1u);
                P.okeys2[pos] = key;
                *P.orecs2.a(pos) = *P.orecs.a(base + i);
                *P.orecs2.b(pos) = *P.orecs.b(base + i);
            }
        }
}

// ocount2[p] = ooff2[p + 1] - ooff2[p], obase2[p] = ooff2[p]   (the re-cut winners are packed)
__global__ __launch_bounds__(BLOCK) void k_unpack_old_dir(const uint32_t *__restrict__ ooff2, uint32_t nparts, uint64_t *__restrict__ obase2,
                                                          uint32_t *__restrict__ ocount2) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    if (p > nparts) return;
    obase2[p] = ooff2[p];
    if (p < nparts) ocount2[p] = ooff2[p + 1] - ooff2[p];
}

// earlier winners per level-1 bin: obin[b] = sum of ocount over the bin's f2old partitions
__global__ __launch_bounds__(BLOCK) void k_old_per_bin(const uint32_t *__restrict__ ocount, uint32_t f2old, uint32_t *__restrict__ obin) {
    const uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= F1) return;
    uint32_t t = 0;
    for (uint32_t q = b * f2old; q < (b + 1) * f2old; q++) t += ocount[q];
    obin[b] = t;
}

// ---------------------------------------------------------------------------------------------------------------
// the fold: one workgroup per partition, open-addressing table in LDS
// ---------------------------------------------------------------------------------------------------------------
struct FoldParams {
    BinSrc src;                    // BINS: partition p = level-1 bin p of pass 0's output
    GridSeg seg;                   // otherwise: partition p of the second level's output
    EntryRef entries;
    GridRef g;
    // earlier winners by partition (okeys == nullptr: none)
    const uint64_t *okeys;
    RecArr orecs;
    const uint64_t *obase;
    const uint32_t *ocount;
    // output
    uint64_t *wkeys;
    RecArr wrecs;
    const uint64_t *wbase;
    uint32_t *wcount;
    uint32_t *palias;              // [P] 1: the partition holds aliased keys
    uint32_t *pay_scratch;         // the parked payloads, 5 words per slot and resident workgroup
    unsigned long long *stats;     // [0] winners, [1] partitions that overflowed the LDS table, [2] partitions with aliased keys,
                                   // [3] partitions k_fold_dense left to k_fold
    uint32_t *defer_list;          // k_fold_dense: the partitions it leaves; k_fold: fold these (stats[3] of them) instead of 0..nparts
};

// Slot of `key` in the LDS table, inserting it if absent; -1 when the table is full (LIMIT cells).
template <int NSLOT, int LIMIT>
__device__ __forceinline__ int lds_find_or_insert(uint64_t *s_key, uint64_t key, uint64_t h, uint32_t *s_ncell) {
    uint32_t s = slot_of<NSLOT>(h);
    for (int probes = 0; probes < NSLOT; probes++) {
        const uint64_t k = __hip_atomic_load(&s_key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == key) return (int)s;
        if (k == PCQ_EMPTY_KEY) {
            const uint64_t prev = atomicCAS((unsigned long long *)&s_key[s], (unsigned long long)PCQ_EMPTY_KEY, (unsigned long long)key);
            if (prev == PCQ_EMPTY_KEY) return atomicAdd(s_ncell, 1u) >= (uint32_t)LIMIT ? -1 : (int)s;
            if (prev == key) return (int)s;
        }
        s = s + 1 == NSLOT ? 0 : s + 1;
    }
    return -1;
}

// One chunk of a partition's tuples into the table (the general path of k_fold): tuple k * NT + thread of the chunk is tu[k].
//   phase 1  cells and their minimum distance: atomicMin on the f64 bits after a plain read — a tuple above the minimum it
//            sees is out (the minimum only falls), which is nearly all of a coarse grid's
//   phase 2  among the tuples at the minimum, the earliest in file order
//   phase 3  a winner from this chunk parks its payload
// No barrier behind phase 3: the next chunk's phase 1 can only make its test fail for a slot whose winner is about to be
// replaced, and every thread passes the next barrier before anyone parks again.
template <int NSLOT, int NT, int FOLD_K, int LIMIT>
__device__ __forceinline__ void fold_chunk(const FoldParams &P, const GridTuple (&tu)[FOLD_K], uint32_t cnt, uint64_t *s_key, uint64_t *s_dist,
                                           uint64_t *s_ord, uint32_t *s_aliasbits, uint32_t *s_oldbits, uint32_t *s_ncell, uint32_t *s_over,
                                           uint32_t *pay) {
    uint64_t dbits[FOLD_K];
    int slot[FOLD_K];
#pragma unroll
    for (int k = 0; k < FOLD_K; k++) {
        const uint32_t i = k * NT + threadIdx.x;
        slot[k] = -1;
        dbits[k] = 0;
        if (i >= cnt) continue;
        const TupleEval ev = eval_tuple(P.g, P.entries, tu[k]);
        dbits[k] = ev.dbits;
        const int s = lds_find_or_insert<NSLOT, LIMIT>(s_key, ev.key, cell_hash(ev.key), s_ncell);
        if (s < 0) {
            *s_over = 1;
            continue;
        }
        slot[k] = s;
        if (ev.alias) atomicOr(&s_aliasbits[s >> 5], 1u << (s & 31));
        const uint64_t seen = __hip_atomic_load(&s_dist[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (a stale value is only too large)
        if (ev.dbits < seen) {
            const uint64_t old = atomicMin((unsigned long long *)&s_dist[s], (unsigned long long)ev.dbits);
            if (ev.dbits < old) s_ord[s] = ~0ull;  // a new minimum: whoever held the cell is out (racing writers store the same value)
        } else if (ev.dbits > seen) {
            slot[k] = -1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FOLD_K; k++) {
        if (slot[k] < 0) continue;
        if (dbits[k] == s_dist[slot[k]]) atomicMin((unsigned long long *)&s_ord[slot[k]], (unsigned long long)ord_of(tu[k]));
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FOLD_K; k++) {
        const int s = slot[k];
        if (s < 0) continue;
        if (dbits[k] == s_dist[s] && s_ord[s] == ord_of(tu[k])) {
            pay[s * 5] = (uint32_t)tu[k].x, pay[s * 5 + 1] = (uint32_t)tu[k].y, pay[s * 5 + 2] = (uint32_t)tu[k].z;
            pay[s * 5 + 3] = tu[k].w0, pay[s * 5 + 4] = tu[k].w1;
            atomicAnd(&s_oldbits[s >> 5], ~(1u << (s & 31)));
        }
    }
}

// Workgroups are persistent: each folds the partitions blockIdx.x, blockIdx.x + gridDim.x, ... (or the partitions
// k_fold_dense left on its list).  BINS (the big shape): the partition is a level-1 bin read through the fragment window;
// the window of the next chunk is asked for (into registers) while the current chunk is folded.  Otherwise the partition
// is a piece of the second level's output, and its range is loaded while the partition before it is folded.
template <int NSLOT, int NT, int FOLD_K, int LIMIT, bool BINS, bool DIRECT, int MIN_WAVES>
__global__ __launch_bounds__(NT, MIN_WAVES) void k_fold(FoldParams P, uint32_t nparts) {
    constexpr int SPT = (NSLOT + NT - 1) / NT;   // slots per thread in the compaction
    constexpr int CHUNK = NT * FOLD_K;
    constexpr int FB = BINS ? (NT > BIG_FB ? BIG_FB : NT - 64) : 1;  // one window entry per thread
    __shared__ uint64_t s_key[NSLOT];
    __shared__ uint64_t s_dist[NSLOT];   // f64 bits of the best squared distance (monotone for d >= 0)
    __shared__ uint64_t s_ord[NSLOT];    // file order of the winner: 0 = an earlier fold's winner, ~0 = none yet
    __shared__ uint32_t s_aliasbits[(NSLOT + 31) / 32], s_oldbits[(NSLOT + 31) / 32];
    __shared__ uint32_t s_pre[FB + 1];
    __shared__ uint64_t s_addr[FB];
    __shared__ uint32_t s_ncell, s_over, s_wsum[NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const GridSeg sg = P.seg;
    const bool seg_wide = sg.wide;
    const uint32_t seg_ts = tuple_bytes(seg_wide);

    // pipeline state (second-level partitions): the range and the output base of the current and the next partition
    uint32_t cur_lo = 0, cur_cnt = 0, nxt_lo = 0, nxt_cnt = 0;
    uint64_t cur_out = 0, nxt_out = 0;
    if (P.defer_list) nparts = (uint32_t)P.stats[3];  // only what k_fold_dense left
    uint32_t it = blockIdx.x, p = 0, p_next = 0;
    if (it < nparts) {
        p_next = P.defer_list ? P.defer_list[it] : (BINS ? xcd_order(it, nparts) : it);
        if (!BINS) cur_lo = sg.off[p_next], cur_cnt = sg.cnt ? sg.cnt[p_next] : sg.off[p_next + 1] - cur_lo;
        cur_out = P.wbase[p_next];
    }
    for (; it < nparts; it += gridDim.x) {
        p = p_next;
        const uint32_t pn = it + gridDim.x;
        if (pn < nparts) {
            p_next = P.defer_list ? P.defer_list[pn] : (BINS ? xcd_order(pn, nparts) : pn);
            if (!BINS) nxt_lo = sg.off[p_next], nxt_cnt = sg.cnt ? sg.cnt[p_next] : sg.off[p_next + 1] - nxt_lo;
            nxt_out = P.wbase[p_next];
        }
        uint32_t *pay = P.pay_scratch + (size_t)blockIdx.x * NSLOT * 5;  // HBM scratch of this workgroup
        const uint32_t n_old = P.okeys ? P.ocount[p] : 0;
        const uint64_t old_base = P.okeys ? P.obase[p] : 0;
        const uint64_t out_base = cur_out;
        for (int t = threadIdx.x; t < NSLOT; t += NT) s_key[t] = PCQ_EMPTY_KEY, s_dist[t] = ~0ull, s_ord[t] = ~0ull;
        for (int t = threadIdx.x; t < (NSLOT + 31) / 32; t += NT) s_aliasbits[t] = 0, s_oldbits[t] = 0;
        if (threadIdx.x == 0) s_ncell = 0, s_over = 0;
        __syncthreads();

        // earlier winners first: their distance is recomputed from the record (same f64 expressions, same bits)
        for (uint32_t i = threadIdx.x; i < n_old; i += NT) {
            const uint64_t key = P.okeys[old_base + i];
            const int s = lds_find_or_insert<NSLOT, LIMIT>(s_key, key, cell_hash(key), &s_ncell);
            if (s < 0) {
                s_over = 1;
                continue;
            }
            const uint4 ra = *P.orecs.a(old_base + i), rb = *P.orecs.b(old_base + i);
            atomicOr(&s_oldbits[s >> 5], 1u << (s & 31));
            pay[s * 5] = (uint32_t)(old_base + i);
            pay[s * 5 + 1] = (uint32_t)((old_base + i) >> 32);
            if (rec_flags(rb) & R_ALIAS) {
                atomicOr(&s_aliasbits[s >> 5], 1u << (s & 31));
            } else {
                const double ox = __longlong_as_double((long long)((uint64_t)ra.x | ((uint64_t)ra.y << 32))),
                             oy = __longlong_as_double((long long)((uint64_t)ra.z | ((uint64_t)ra.w << 32))),
                             oz = __longlong_as_double((long long)((uint64_t)rb.x | ((uint64_t)rb.y << 32)));
                uint64_t cell[3];
                const DevGrid &gf = *P.g.full;
#pragma unroll
                for (int a = 0; a < 3; a++) cell[a] = (key >> gf.shift[a]) & gf.mask[a];  // not aliased: unmasked == masked
                s_dist[s] = (uint64_t)__double_as_longlong(centre_dist(gf, cell, ox, oy, oz));
                s_ord[s] = 0;
            }
        }
        if (n_old) __syncthreads();

        // The common partition of a dense grid: at most one chunk of tuples, no earlier winners.  Every thread
        // still holds its tuples when the winners are known, so the winner of a cell writes its record straight from
        // registers — no payload parked, no sweep over the table's slots.
        if (!BINS && DIRECT && n_old == 0 && cur_cnt <= (uint32_t)CHUNK) {
            const uint32_t cnt = cur_cnt;
            GridTuple tu[FOLD_K];
            uint64_t dbits[FOLD_K];
            int slot[FOLD_K];
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) {
                const uint32_t i = k * NT + threadIdx.x;
                tu[k] = ld_tuple(sg.tuples + (uint64_t)(cur_lo + (i < cnt ? i : (cnt ? cnt - 1 : 0))) * seg_ts, seg_wide);
            }
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) {  // phase 1: cells and their minimum distance
                const uint32_t i = k * NT + threadIdx.x;
                slot[k] = -1;
                dbits[k] = 0;
                if (i >= cnt) continue;
                const TupleEval ev = eval_tuple(P.g, P.entries, tu[k]);
                dbits[k] = ev.dbits;
                const int s = lds_find_or_insert<NSLOT, LIMIT>(s_key, ev.key, cell_hash(ev.key), &s_ncell);
                if (s < 0) {
                    s_over = 1;
                    continue;
                }
                slot[k] = s;
                if (ev.alias) atomicOr(&s_aliasbits[s >> 5], 1u << (s & 31));
                if (ev.dbits < __hip_atomic_load(&s_dist[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                    atomicMin((unsigned long long *)&s_dist[s], (unsigned long long)ev.dbits);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < FOLD_K; k++)  // phase 2: among the tuples at the minimum, the earliest in file order
                if (slot[k] >= 0 && dbits[k] == s_dist[slot[k]]) atomicMin((unsigned long long *)&s_ord[slot[k]], (unsigned long long)ord_of(tu[k]));
            __syncthreads();
            if (s_over) {
                if (threadIdx.x == 0) {
                    P.wcount[p] = 0;
                    atomicAdd(&P.stats[1], 1ull);
                }
            } else {
                // every occupied slot has exactly one tuple at (minimum distance, earliest order): its thread writes the cell
                bool win[FOLD_K];
                uint32_t mine = 0;
#pragma unroll
                for (int k = 0; k < FOLD_K; k++) {
                    win[k] = slot[k] >= 0 && dbits[k] == s_dist[slot[k]] && s_ord[slot[k]] == ord_of(tu[k]);
                    mine += win[k] ? 1 : 0;
                }
                uint32_t incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t up = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += up;
                }
                if (lane == 63) s_wsum[wave] = incl;
                __syncthreads();
                uint32_t before = incl - mine, total = 0;
                for (int w = 0; w < NT / 64; w++) {
                    before += w < wave ? s_wsum[w] : 0;
                    total += s_wsum[w];
                }
                bool any_alias = false;
                uint64_t o = out_base + before;
#pragma unroll
                for (int k = 0; k < FOLD_K; k++) {
                    if (!win[k]) continue;
                    const int sl = slot[k];
                    P.wkeys[o] = s_key[sl];
                    if ((s_aliasbits[sl >> 5] >> (sl & 31)) & 1) {  // left to the exact replay: no point yet, the flag
                        any_alias = true;
                        *P.wrecs.a(o) = make_uint4(0, 0, 0, 0);
                        *P.wrecs.b(o) = make_uint4(0, 0, 0, (uint32_t)R_ALIAS << 24);
                    } else {
                        st_record(P.wrecs, o, P.entries.get((tu[k].w0 >> 8) & 0xff), tu[k].x, tu[k].y, tu[k].z, tu[k].w0, tu[k].w1, R_HAS);
                    }
                    o++;
                }
                if (__syncthreads_or(any_alias) && threadIdx.x == 0) {
                    P.palias[p] = 1;
                    atomicAdd(&P.stats[2], 1ull);
                }
                if (threadIdx.x == 0) {
                    P.wcount[p] = total;
                    if (total) atomicAdd(&P.stats[0], (unsigned long long)total);
                }
            }
        } else {
            if (BINS) {
                // The bin through the fragment window, one round per chunk, software-pipelined: a round publishes the window
                // that covers the NEXT chunk (its entries were asked for, into registers, a round earlier), asks for that
                // chunk's tuples and for the window behind it, and only then folds the chunk whose tuples the round before
                // asked for — the search and the memory round trip of a chunk (4.8 of 12 us per chunk when they came first) run
                // under the fold of the chunk before.
                const BinSrc &S = P.src;
                const uint32_t total_in = uni32(ldg(S.preT + (size_t)p * S.Tp1 + S.T));
                uint32_t f_lo = 0, j0 = 0;  // the next chunk to plan starts at tuple j0 of the bin, in fragment f_lo or behind
                uint32_t nfr = S.T < (uint32_t)FB ? S.T : (uint32_t)FB;
                uint32_t reg_pre = 0;
                uint64_t reg_addr = 0;
                if (threadIdx.x <= nfr) reg_pre = ldg(S.preT + (size_t)p * S.Tp1 + threadIdx.x);
                if (threadIdx.x < nfr) reg_addr = frag_addr(S, p, threadIdx.x);
                GridTuple tu[FOLD_K], tn[FOLD_K];
                uint32_t cnt_n = 0;
#pragma unroll
                for (int k = 0; k < FOLD_K; k++) tn[k] = GridTuple{0, 0, 0, 0, 0, 0, 0};
                for (;;) {  // (every condition below is the same for the whole workgroup)
#pragma unroll
                    for (int k = 0; k < FOLD_K; k++) tu[k] = tn[k];
                    const uint32_t cnt = cnt_n;
                    const bool more = j0 < total_in;
                    if (more) {
                        if (threadIdx.x <= nfr) s_pre[threadIdx.x] = reg_pre;
                        if (threadIdx.x < nfr) s_addr[threadIdx.x] = reg_addr;
                    }
                    __syncthreads();
                    cnt_n = 0;
                    if (more) {
                        const uint32_t wend = uni32(s_pre[nfr]);
                        cnt_n = wend - j0 < (uint32_t)CHUNK ? wend - j0 : (uint32_t)CHUNK;
                        const uint32_t j1 = j0 + cnt_n;
                        if (cnt_n) {
#pragma unroll
                            for (int k = 0; k < FOLD_K; k++) {
                                const uint32_t i = k * NT + threadIdx.x;
                                tn[k] = frag_ld_tuple(s_pre, s_addr, nfr, j0 + (i < cnt_n ? i : cnt_n - 1));
                            }
                        }
                        // the window behind: from the fragment tuple j1 lies in
                        const uint32_t f_next = j1 == wend ? f_lo + nfr : f_lo + uni32(frag_find(s_pre, nfr, j1));
                        const uint32_t nfr_next = S.T - f_next < (uint32_t)FB ? S.T - f_next : (uint32_t)FB;
                        if (j1 < total_in) {
                            if (threadIdx.x <= nfr_next) reg_pre = ldg(S.preT + (size_t)p * S.Tp1 + f_next + threadIdx.x);
                            if (threadIdx.x < nfr_next) reg_addr = frag_addr(S, p, f_next + threadIdx.x);
                        }
                        f_lo = f_next, nfr = nfr_next, j0 = j1;
                    }
                    if (cnt) {
                        fold_chunk<NSLOT, NT, FOLD_K, LIMIT>(P, tu, cnt, s_key, s_dist, s_ord, s_aliasbits, s_oldbits, &s_ncell, &s_over, pay);
                    } else {
                        if (!more) break;
                        __syncthreads();  // (nothing folded this round: everyone has read the window before the next round rewrites it)
                    }
                }
            } else {
                for (uint32_t j0 = 0; j0 < cur_cnt; j0 += CHUNK) {
                    const uint32_t cnt = cur_cnt - j0 < (uint32_t)CHUNK ? cur_cnt - j0 : (uint32_t)CHUNK;
                    GridTuple tu[FOLD_K];
#pragma unroll
                    for (int k = 0; k < FOLD_K; k++) {
                        const uint32_t i = k * NT + threadIdx.x;
                        tu[k] = ld_tuple(sg.tuples + (uint64_t)(cur_lo + j0 + (i < cnt ? i : cnt - 1)) * seg_ts, seg_wide);
                    }
                    fold_chunk<NSLOT, NT, FOLD_K, LIMIT>(P, tu, cnt, s_key, s_dist, s_ord, s_aliasbits, s_oldbits, &s_ncell, &s_over, pay);
                }
            }
            // The parked payloads are read back by other threads of THIS workgroup: a workgroup-scope fence (the stores have
            // left the wave; all waves of a workgroup share one L1).  The device-scope fence that stood here made every wave
            // write the L2's dirty lines back (buffer_wbl2) — 16 times per partition and CU, with the winners of all
            // partitions in flight.
            __threadfence_block();
            __syncthreads();
            if (s_over) {  // more cells than the table holds: the host repeats the fold with more partitions
                if (threadIdx.x == 0) {
                    P.wcount[p] = 0;
                    atomicAdd(&P.stats[1], 1ull);
                }
            } else {
                // compaction: thread t owns slots [t * SPT, ...): winners leave in slot order.  The parked payloads were written
                // by other threads of this workgroup; every wave has fenced (its stores are in the L2, the CU's L1 holds nothing
                // of the scratch) before the barrier above, so they are read with plain loads — all of a thread's slots asked for
                // together: as relaxed atomic loads, slot after slot, this sweep was a chain of 7 x 5 memory round trips per
                // partition, a fifth of the big fold's time.
                uint32_t mine = 0;
                const int s0 = threadIdx.x * SPT;
                uint64_t keys[SPT];
                u32x4_a4 wa[SPT];
                uint32_t wb[SPT];
#pragma unroll
                for (int j = 0; j < SPT; j++) {
                    const int s = s0 + j < NSLOT ? s0 + j : NSLOT - 1;
                    keys[j] = s0 + j < NSLOT ? s_key[s] : PCQ_EMPTY_KEY;
                    wa[j] = *(const PCQ_GLOBAL u32x4_a4 *)(pay + s * 5);
                    wb[j] = *(const PCQ_GLOBAL uint32_t *)(pay + s * 5 + 4);
                    mine += keys[j] != PCQ_EMPTY_KEY ? 1 : 0;
                }
                uint32_t incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t up = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += up;
                }
                if (lane == 63) s_wsum[wave] = incl;
                __syncthreads();
                uint32_t before = incl - mine, total = 0;
                for (int w = 0; w < NT / 64; w++) {
                    before += w < wave ? s_wsum[w] : 0;
                    total += s_wsum[w];
                }
                bool any_alias = false;
                uint64_t o = out_base + before;
#pragma unroll
                for (int j = 0; j < SPT; j++) {
                    const int s = s0 + j;
                    const uint64_t key = keys[j];
                    if (key == PCQ_EMPTY_KEY) continue;
                    P.wkeys[o] = key;
                    const bool alias = (s_aliasbits[s >> 5] >> (s & 31)) & 1, old = (s_oldbits[s >> 5] >> (s & 31)) & 1;
                    if (alias) {  // left to the exact replay: the state before this fold (the earlier winner, if there is one) + the flag
                        any_alias = true;
                        uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, (uint32_t)R_ALIAS << 24);
                        for (uint32_t i = 0; i < n_old; i++)
                            if (P.okeys[old_base + i] == key) {
                                a = *P.orecs.a(old_base + i), b = *P.orecs.b(old_base + i);
                                b.w |= (uint32_t)R_ALIAS << 24;
                                break;
                            }
                        *P.wrecs.a(o) = a;
                        *P.wrecs.b(o) = b;
                    } else if (old) {
                        const uint64_t oi = (uint64_t)wa[j].x | ((uint64_t)wa[j].y << 32);
                        *P.wrecs.a(o) = *P.orecs.a(oi);
                        *P.wrecs.b(o) = *P.orecs.b(oi);
                    } else {
                        st_record(P.wrecs, o, P.entries.get((wa[j].w >> 8) & 0xff), (int32_t)wa[j].x, (int32_t)wa[j].y, (int32_t)wa[j].z, wa[j].w, wb[j], R_HAS);
                    }
                    o++;
                }
                if (__syncthreads_or(any_alias) && threadIdx.x == 0) {
                    P.palias[p] = 1;
                    atomicAdd(&P.stats[2], 1ull);
                }
                if (threadIdx.x == 0) {
                    P.wcount[p] = total;
                    if (total) atomicAdd(&P.stats[0], (unsigned long long)total);
                }
            }
        }
        cur_lo = nxt_lo, cur_cnt = nxt_cnt, cur_out = nxt_out;
        __syncthreads();  // the table is cleared for the next partition
    }
}

// The fold of a dense grid's partitions, on its own: one segment (the second level's output), no earlier winners, the
// partition's tuples in one chunk of registers.  k_fold handles every case and pays for it in registers (168, three waves
// per SIMD) — and a small partition is a chain of latencies (its offsets, its tuples, three barriers, the stores), so the
// waves per CU decide its speed — as long as nothing is spilled.  This kernel keeps only the common case: 512 threads x 3 tuples,
// two workgroups per CU (three needed 80 registers and spilled 92 bytes per lane and partition: 4.8 GB of scratch each way).
//  * every thread holds its tuples from the load to the end: the winner of a cell writes the record from registers;
//  * the table is cleared once: every occupied slot has exactly one winner, which resets the slot behind itself;
//  * the tuples are loaded and evaluated BEFORE the barrier that separates the partitions;
//  * a partition with more tuples than a chunk goes on stats[3] / defer_list for k_fold.
// k_fold_dense's insert: the compare-and-swap IS the probe (a cell's first tuple — three of four in a dense grid — takes
// one LDS round trip instead of a read and then the swap), and the probe sequence is double hashing: with linear probing
// the 64 lanes of a wave leave the loop together, after the longest cluster any of them ran into.  The table can never
// fill up (at most a chunk of 1536 tuples goes into 2048 slots), so the loop ends; the cells are counted per wave.
template <int NSLOT>
__device__ __forceinline__ uint32_t lds_insert_dense(uint64_t *s_key, uint64_t key, uint64_t h, bool *fresh) {
    static_assert((NSLOT & (NSLOT - 1)) == 0, "the step below visits every slot of a power-of-two table");
    uint32_t s = slot_of<NSLOT>(h);
    const uint32_t step = ((uint32_t)(h >> 15) & (NSLOT - 1)) | 1u;
    *fresh = false;
    for (;;) {
        const uint64_t prev = atomicCAS((unsigned long long *)&s_key[s], (unsigned long long)PCQ_EMPTY_KEY, (unsigned long long)key);
        if (prev == PCQ_EMPTY_KEY) {
            *fresh = true;
            return s;
        }
        if (prev == key) return s;
        s = (s + step) & (NSLOT - 1);
    }
}

struct DenseParams {
    const uint8_t *tuples;         // the second level's output: partition p = tuples off[p] .. off[p] + cnt[p]
    uint32_t wide;                 // of 24 bytes (20 otherwise)
    const uint32_t *off;           // (cnt == nullptr: .. off[p + 1])
    const uint32_t *cnt;
    EntryRef entries;
    GridRef g;                     // (full: device copy of the whole grid, for the exact computation next to a cell boundary)
    uint64_t *wkeys;
    RecArr wrecs;
    const uint64_t *wbase;
    uint32_t *wcount;
    uint32_t *palias;
    unsigned long long *stats;
    uint32_t *defer_list;
};
template <int NSLOT, int NT, int FOLD_K, int LIMIT, int MIN_WAVES>
__global__ __launch_bounds__(NT, MIN_WAVES) void k_fold_dense(DenseParams P, uint32_t nparts) {
    constexpr int CHUNK = NT * FOLD_K;
    __shared__ uint64_t s_key[NSLOT];
    __shared__ uint64_t s_dist[NSLOT];
    __shared__ uint64_t s_ord[NSLOT];
    __shared__ uint32_t s_aliasbits[(NSLOT + 31) / 32];
    __shared__ uint32_t s_ncell, s_wsum[NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *tuples = P.tuples;
    const bool wide = P.wide;
    const uint32_t ts = tuple_bytes(wide);
    const uint32_t *off = P.off;
    for (int t = threadIdx.x; t < NSLOT; t += NT) s_key[t] = PCQ_EMPTY_KEY, s_dist[t] = ~0ull, s_ord[t] = ~0ull;
    for (int t = threadIdx.x; t < (NSLOT + 31) / 32; t += NT) s_aliasbits[t] = 0;
    if (threadIdx.x == 0) s_ncell = 0;
    unsigned long long winners = 0;  // thread 0: this workgroup's winners

    uint32_t cur_lo = 0, cur_cnt = 0, nxt_lo = 0, nxt_cnt = 0;
    uint64_t cur_out = 0, nxt_out = 0;
    uint32_t p = blockIdx.x;
    // (the partition's range and output base are the same for the whole workgroup: scalar registers)
    const uint32_t *cntp = P.cnt;
    if (p < nparts) cur_lo = uni32(off[p]), cur_cnt = cntp ? uni32(cntp[p]) : uni32(off[p + 1]) - cur_lo, cur_out = uni64(P.wbase[p]);
    for (; p < nparts; p += gridDim.x) {
        const uint32_t pn = p + gridDim.x;
        if (pn < nparts) nxt_lo = uni32(off[pn]), nxt_cnt = cntp ? uni32(cntp[pn]) : uni32(off[pn + 1]) - nxt_lo, nxt_out = uni64(P.wbase[pn]);
        const uint32_t cnt = cur_cnt;
        GridTuple tu[FOLD_K];
        if (cnt > (uint32_t)CHUNK) {  // (the same for every thread of the workgroup)
            if (threadIdx.x == 0) P.defer_list[atomicAdd(&P.stats[3], 1ull)] = p;
            cur_lo = nxt_lo, cur_cnt = nxt_cnt, cur_out = nxt_out;
            continue;
        }
        uint64_t key[FOLD_K], dbits[FOLD_K];
        bool alias[FOLD_K];
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) {
            const uint32_t i = k * NT + threadIdx.x;
            tu[k] = ld_tuple(tuples + (uint64_t)(cur_lo + (i < cnt ? i : (cnt ? cnt - 1 : 0))) * ts, wide);
        }
        uint32_t inexact = 0;  // bit k: tuple k is next to a cell boundary (or outside the short computation's range)
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) {
            const GridEntryDev e = P.entries.get((tu[k].w0 >> 8) & 0xff);
            const double px = world(tu[k].x, e.scale[0], e.offset[0]), py = world(tu[k].y, e.scale[1], e.offset[1]),
                         pz = world(tu[k].z, e.scale[2], e.offset[2]);
            const CellFast cf = cell_fast(P.g.f, px, py, pz);
            key[k] = key_fast(P.g.f, cf, &alias[k]);
            dbits[k] = (uint64_t)__double_as_longlong(centre_dist_fast(P.g.f, cf, px, py, pz));
            inexact |= cf.ok ? 0u : 1u << k;
        }
        if (__any(inexact != 0)) {  // rare: one copy of the exact computation, off the common path
#pragma unroll 1
            for (int kk = 0; kk < FOLD_K; kk++) {
                if (!((inexact >> kk) & 1)) continue;
                int32_t x = tu[0].x, y = tu[0].y, z = tu[0].z;
                uint32_t w0 = tu[0].w0;
#pragma unroll
                for (int j = 1; j < FOLD_K; j++)
                    if (j == kk) x = tu[j].x, y = tu[j].y, z = tu[j].z, w0 = tu[j].w0;
                const GridEntryDev e = P.entries.get((w0 >> 8) & 0xff);
                const TupleEval ev = eval_exact(*P.g.full, world(x, e.scale[0], e.offset[0]), world(y, e.scale[1], e.offset[1]), world(z, e.scale[2], e.offset[2]));
#pragma unroll
                for (int j = 0; j < FOLD_K; j++)
                    if (j == kk) key[j] = ev.key, dbits[j] = ev.dbits, alias[j] = ev.alias;
            }
        }
        __syncthreads();  // the table is clean: the previous partition's winners have reset their slots
        int slot[FOLD_K];
        uint32_t fresh_cells = 0;
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) {  // phase 1: cells and their minimum distance
            slot[k] = -1;
            if ((uint32_t)(k * NT) + threadIdx.x >= cnt) continue;
            bool fresh;
            const uint32_t sl = lds_insert_dense<NSLOT>(s_key, key[k], cell_hash(key[k]), &fresh);
            fresh_cells += fresh ? 1 : 0;
            slot[k] = (int)sl;
            if (alias[k]) atomicOr(&s_aliasbits[sl >> 5], 1u << (sl & 31));
            atomicMin((unsigned long long *)&s_dist[sl], (unsigned long long)dbits[k]);
        }
        bool over = false;
        {  // cells of the partition so far, counted per wave; beyond LIMIT the partition is given up (like k_fold: the same fan-out rule)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) fresh_cells += __shfl_xor(fresh_cells, o, 64);
            if (lane == 0 && fresh_cells) over = atomicAdd(&s_ncell, fresh_cells) + fresh_cells > (uint32_t)LIMIT;
        }
        if (__syncthreads_or(over)) {  // more cells than the table holds: the host repeats the fold with more partitions
            for (int t = threadIdx.x; t < NSLOT; t += NT) s_key[t] = PCQ_EMPTY_KEY, s_dist[t] = ~0ull, s_ord[t] = ~0ull;
            for (int t = threadIdx.x; t < (NSLOT + 31) / 32; t += NT) s_aliasbits[t] = 0;
            if (threadIdx.x == 0) {
                s_ncell = 0;
                P.wcount[p] = 0;
                atomicAdd(&P.stats[1], 1ull);
            }
            cur_lo = nxt_lo, cur_cnt = nxt_cnt, cur_out = nxt_out;
            continue;
        }
        bool cand[FOLD_K];
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) {  // phase 2: among the tuples at the minimum, the earliest in file order
            cand[k] = slot[k] >= 0 && dbits[k] == s_dist[slot[k]];
            if (cand[k]) atomicMin((unsigned long long *)&s_ord[slot[k]], (unsigned long long)ord_of(tu[k]));
        }
        __syncthreads();
        // Every occupied slot has exactly one tuple at (minimum distance, earliest order): its thread writes the cell.  The
        // winners leave wave by wave and, inside a wave, tuple slot by tuple slot (k), in lane order: the lanes of ONE store
        // instruction then write one contiguous run of keys (8 bytes each) and of records — with a per-thread order the same
        // instruction wrote every second or third record of a 4 KiB span, and the 8-byte key stores reached the memory side as
        // partial writes (counted: 6.7 GB written and 1.9 GB fetched beyond the tuples for 4.9 GB of winners).
        uint32_t cnt_k[FOLD_K], rank_k[FOLD_K], wave_total = 0;
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) {
            cand[k] = cand[k] && s_ord[slot[k]] == ord_of(tu[k]);
            const unsigned long long m = __ballot(cand[k]);
            cnt_k[k] = (uint32_t)__popcll(m);
            rank_k[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            wave_total += cnt_k[k];
        }
        if (lane == 0) s_wsum[wave] = wave_total;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) {
            before += w < wave ? s_wsum[w] : 0;
            total += s_wsum[w];
        }
        uint64_t run_base = cur_out + before;  // the first place of this wave's winners of tuple slot k
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) {
            const uint64_t o = run_base + rank_k[k];
            run_base += cnt_k[k];
            if (!cand[k]) continue;
            const int sl = slot[k];
            P.wkeys[o] = s_key[sl];  // (= key[k]: read back instead of kept in two registers per tuple across the barriers)
            const uint32_t abit = 1u << (sl & 31);
            if (s_aliasbits[sl >> 5] & abit) {  // left to the exact replay: no point yet, the flag
                *P.wrecs.a(o) = make_uint4(0, 0, 0, 0);
                *P.wrecs.b(o) = make_uint4(0, 0, 0, (uint32_t)R_ALIAS << 24);
                atomicAnd(&s_aliasbits[sl >> 5], ~abit);
                if (atomicExch(&P.palias[p], 1u) == 0) atomicAdd(&P.stats[2], 1ull);
            } else {
                // (the record's coordinates are computed again from the integers: kept from the evaluation above they would
                // cost nine registers per tuple across the three barriers — the compiler spills them if it sees the same
                // expression, hence the opaque copies)
                int32_t x = tu[k].x, y = tu[k].y, z = tu[k].z;
                uint32_t w0 = tu[k].w0;
                asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w0));
                st_record(P.wrecs, o, P.entries.get((w0 >> 8) & 0xff), x, y, z, w0, tu[k].w1, R_HAS);
            }
            s_key[sl] = PCQ_EMPTY_KEY, s_dist[sl] = ~0ull, s_ord[sl] = ~0ull;
        }
        if (threadIdx.x == 0) {
            s_ncell = 0;
            P.wcount[p] = total;
            winners += total;
        }
        cur_lo = nxt_lo, cur_cnt = nxt_cnt, cur_out = nxt_out;
    }
    if (threadIdx.x == 0 && winners) atomicAdd(&P.stats[0], winners);
}


// ---------------------------------------------------------------------------------------------------------------
// aliased keys: exact sequential replay (grid_sampling.rs:72-103), rare
// ---------------------------------------------------------------------------------------------------------------
// The tuples of this fold that belong to aliased keys: counted (EMIT = false) or appended to `list` (EMIT = true).
// BINS: the partitions are pass 0's bins (no second level), otherwise pieces of the second level's output.
template <bool EMIT, bool BINS>
__global__ __launch_bounds__(L2_NT) void k_alias_gather(FoldParams P, AliasItem *__restrict__ list, unsigned long long *__restrict__ cursor) {
    __shared__ uint64_t s_akeys[BIG_LIMIT];
    __shared__ uint32_t s_pre[L2_FB + 1];
    __shared__ uint64_t s_addr[L2_FB];
    __shared__ uint32_t s_n;
    const uint32_t p = blockIdx.x;
    if (!P.palias[p]) return;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t wb = P.wbase[p];
    const uint32_t wn = P.wcount[p];
    for (uint32_t i = threadIdx.x; i < wn; i += L2_NT)
        if (rec_flags(*P.wrecs.b(wb + i)) & R_ALIAS) s_akeys[atomicAdd(&s_n, 1u)] = P.wkeys[wb + i];
    __syncthreads();
    const uint32_t na = s_n;
    uint32_t mine = 0;
    auto visit = [&](const GridTuple &t) {
        const uint64_t key = eval_tuple(P.g, P.entries, t).key;
        bool hit = false;
        for (uint32_t q = 0; q < na && !hit; q++) hit = s_akeys[q] == key;
        if (!hit) return;
        if (EMIT) {
            AliasItem it;
            it.key = key, it.ord = ord_of(t);
            it.x = t.x, it.y = t.y, it.z = t.z, it.w0 = t.w0, it.w1 = t.w1, it._pad = 0;
            list[atomicAdd(cursor, 1ull)] = it;
        } else {
            mine++;
        }
    };
    if (BINS) {
        bin_for_each<L2_NT, L2_FB, 1>(P.src, p, s_pre, s_addr, visit);
    } else {
        const GridSeg sg = P.seg;
        const bool wide = sg.wide;
        const uint32_t lo = sg.off[p], hi = lo + (sg.cnt ? sg.cnt[p] : sg.off[p + 1] - lo);
        for (uint32_t i = lo + threadIdx.x; i < hi; i += L2_NT) visit(ld_tuple(sg.tuples + (uint64_t)i * tuple_bytes(wide), wide));
    }
    if (!EMIT && mine) atomicAdd(cursor, (unsigned long long)mine);
}

// sorted[rank] = list[e], rank = number of items in front of it by (key, file order); quadratic, the list is short
__global__ __launch_bounds__(BLOCK) void k_alias_rank(const AliasItem *__restrict__ list, uint64_t n, AliasItem *__restrict__ sorted) {
    const uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    const AliasItem me = list[e];
    uint64_t rank = 0;
    for (uint64_t j = 0; j < n; j++) {
        const uint64_t k = list[j].key, o = list[j].ord;
        rank += (k < me.key || (k == me.key && o < me.ord)) ? 1 : 0;
    }
    sorted[rank] = me;
}

// The thread of the FIRST item of a key owns that key: it applies insert_point to the key's items in file order,
// starting from the state the fold left in the winner record.
__global__ __launch_bounds__(BLOCK) void k_alias_replay(const AliasItem *__restrict__ sorted, uint64_t n, FoldParams P, uint32_t f2) {
    const uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    const uint64_t key = sorted[e].key;
    if (e > 0 && sorted[e - 1].key == key) return;
    const uint64_t h = cell_hash(key);
    const uint32_t p = bin_of(h) * f2 + sub_of(h, f2);
    const uint64_t wb = P.wbase[p];
    const uint32_t wn = P.wcount[p];
    uint64_t o = ~0ull;
    for (uint32_t i = 0; i < wn && o == ~0ull; i++)
        if (P.wkeys[wb + i] == key) o = wb + i;
    if (o == ~0ull) return;  // cannot happen: the fold wrote a record for every key it saw
    const uint4 ra = *P.wrecs.a(o), rb = *P.wrecs.b(o);
    bool has = rec_flags(rb) & R_HAS;
    double cx = __longlong_as_double((long long)((uint64_t)ra.x | ((uint64_t)ra.y << 32))),
           cy = __longlong_as_double((long long)((uint64_t)ra.z | ((uint64_t)ra.w << 32))),
           cz = __longlong_as_double((long long)((uint64_t)rb.x | ((uint64_t)rb.y << 32)));
    bool changed = false;
    AliasItem best = sorted[e];
    for (uint64_t q = e; q < n && sorted[q].key == key; q++) {
        const AliasItem it = sorted[q];
        const GridEntryDev en = P.entries.get((it.w0 >> 8) & 0xff);
        const double px = world(it.x, en.scale[0], en.offset[0]), py = world(it.y, en.scale[1], en.offset[1]), pz = world(it.z, en.scale[2], en.offset[2]);
        bool take;
        if (!has) {
            take = true;  // grid_sampling.rs:73-76
        } else {          // :77-103 — both distances against the NEW point's (unmasked) cell centre
            const DevGrid &gf = *P.g.full;
            const CellInfo ci = cell_of(gf, px, py, pz);
            take = centre_dist(gf, ci.cell, px, py, pz) < centre_dist(gf, ci.cell, cx, cy, cz);
        }
        if (take) {
            best = it, cx = px, cy = py, cz = pz;
            has = true;
            changed = true;
        }
    }
    if (changed) st_record(P.wrecs, o, P.entries.get((best.w0 >> 8) & 0xff), best.x, best.y, best.z, best.w0, best.w1, R_HAS | R_ALIAS);
}

// ---------------------------------------------------------------------------------------------------------------
// drain: the winners of partition p, packed to 31-byte points at out31[dpre[p] ...] and their keys
// ---------------------------------------------------------------------------------------------------------------
constexpr int DRAIN_RECS = 1024;  // records per LDS image
constexpr int DRAIN_STAGE = DRAIN_RECS * 31 + 16;

__global__ __launch_bounds__(BLOCK) void k_drain(const uint64_t *__restrict__ wkeys, RecArr wrecs, const uint64_t *__restrict__ wbase,
                                                 const uint32_t *__restrict__ wcount, const uint32_t *__restrict__ dpre, uint8_t *__restrict__ out31,
                                                 uint64_t *__restrict__ keys_out) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[DRAIN_STAGE];
    const uint32_t p = blockIdx.x, n = wcount[p];
    if (n == 0) return;
    const uint64_t src = wbase[p], dst = dpre[p];
    if (keys_out)
        for (uint32_t i = threadIdx.x; i < n; i += BLOCK) keys_out[dst + i] = wkeys[src + i];
    if (!out31) return;
    // the 31-byte records are assembled in LDS congruent (mod 16) to their place in the output and leave as 16-byte
    // stores; only the two ragged ends use byte stores, so neighbouring pieces never write the same 16 bytes
    for (uint32_t c0 = 0; c0 < n; c0 += DRAIN_RECS) {
        const uint32_t m = n - c0 < DRAIN_RECS ? n - c0 : DRAIN_RECS;
        const uint64_t gbyte0 = (dst + c0) * 31ull;
        const uint32_t pad = (uint32_t)(gbyte0 & 15);
        for (uint32_t i = threadIdx.x; i < m; i += BLOCK) {
            const uint4 a = *wrecs.a(src + c0 + i), b = *wrecs.b(src + c0 + i);
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint8_t *dp = s_stage + pad + 31u * i;
#pragma unroll
            for (int k = 0; k < 31; k++) dp[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
        }
        __syncthreads();
        const uint32_t total = pad + m * 31u;
        uint8_t *gdst = out31 + (gbyte0 - pad);
        for (uint32_t b0 = threadIdx.x * 16u; b0 < total; b0 += BLOCK * 16u) {
            const uint32_t b1 = b0 + 16u;
            if (b0 >= pad && b1 <= total) {
                *reinterpret_cast<uint4 *>(gdst + b0) = *reinterpret_cast<const uint4 *>(s_stage + b0);
            } else {
                const uint32_t lo = b0 > pad ? b0 : pad, hi = b1 < total ? b1 : total;
                for (uint32_t k = lo; k < hi; k++) gdst[k] = s_stage[k];
            }
        }
        __syncthreads();
    }
}


}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct GridRun {       // the output of one pass-0 launch
    uint8_t *tuples;   // ntiles blocks of P0_TILE tuples
    uint16_t *dir;     // ntiles directory rows
    uint32_t ntiles;
    uint32_t wide;     // 24-byte tuples (the scan had a colour column)
};

struct GridState {
    // pending: pass-0 runs not folded yet
    std::vector<GridRun> runs;
    std::vector<GridEntryDev> entries;
    std::vector<uint64_t> entry_base;  // first file-order index of each entry
    uint64_t entry_end = 0;            // index behind the last point scanned into the last entry
    std::vector<void *> slabs;         // pool blocks holding the blocks and directory rows
    uint8_t *slab_cur = nullptr;
    size_t slab_left = 0;
    uint64_t pending_cap = 0;          // points scanned into the pending runs (= the most tuples they can hold)
    uint64_t pending_tiles = 0;
    bool any_wide = false;
    // folded winners, grouped by partition
    uint64_t *wkeys = nullptr;
    uint8_t *wrecs = nullptr;    // two arrays of 16-byte halves (RecArr), wrec_cap records each
    uint64_t wrec_cap = 0;
    uint64_t *wbase = nullptr;   // [P + 1]
    uint32_t *wcount = nullptr;  // [P]
    uint32_t f2 = 1;
    uint64_t wtotal = 0;
};

static void grid_free_pending(pcq_ctx *ctx, GridState *gs) {
    for (void *p : gs->slabs) pcq_pool_free(ctx, p);
    gs->slabs.clear();
    gs->slab_cur = nullptr;
    gs->slab_left = 0;
    gs->pending_cap = 0;
    gs->pending_tiles = 0;
    gs->any_wide = false;
    gs->runs.clear();
    gs->entries.clear();
    gs->entry_base.clear();
    gs->entry_end = 0;
}

static void grid_free_winners(pcq_ctx *ctx, GridState *gs) {
    pcq_pool_free(ctx, gs->wkeys);
    pcq_pool_free(ctx, gs->wrecs);
    pcq_pool_free(ctx, gs->wbase);
    pcq_pool_free(ctx, gs->wcount);
    gs->wkeys = nullptr, gs->wrecs = nullptr, gs->wbase = nullptr, gs->wcount = nullptr;
    gs->wrec_cap = 0;
    gs->wtotal = 0;
    gs->f2 = 1;
}

void pcq_grid_release(pcq_collector *c) {
    if (!c->gs) return;
    grid_free_pending(c->ctx, c->gs);
    grid_free_winners(c->ctx, c->gs);
    delete c->gs;
    c->gs = nullptr;
}

// a scratch list of pool blocks released together
struct Scratch {
    pcq_ctx *ctx;
    std::vector<void *> blocks;
    explicit Scratch(pcq_ctx *c) : ctx(c) {}
    ~Scratch() {
        for (void *p : blocks) pcq_pool_free(ctx, p);
    }
    template <typename T>
    int get(size_t count, T **out) {
        void *p = nullptr;
        const int rc = pcq_pool_alloc(ctx, count * sizeof(T), &p);
        if (rc) return rc;
        blocks.push_back(p);
        *out = (T *)p;
        return PCQ_OK;
    }
    void keep(void *p) { blocks.erase(std::remove(blocks.begin(), blocks.end(), p), blocks.end()); }
};
// Declared BEHIND a Scratch: whichever way the scope is left, the stream has drained before the Scratch hands its blocks
// back to the pool ("a block may be freed only when the work that used it has completed").
struct StreamDrainOnExit {
    hipStream_t s;
    explicit StreamDrainOnExit(hipStream_t stream) : s(stream) {}
    ~StreamDrainOnExit() { (void)hipStreamSynchronize(s); }
};

static int grid_fold(pcq_ctx *ctx, pcq_collector *c);

// `bytes` of the pending slabs, 256-byte aligned
static int grid_room(pcq_ctx *ctx, GridState *gs, size_t bytes, void **out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > gs->slab_left) {
        size_t slab = 256ull << 20;
        if (slab < bytes) slab = bytes;
        void *p = nullptr;
        const int rc = pcq_pool_alloc(ctx, slab, &p);
        if (rc) return rc;
        gs->slabs.push_back(p);
        gs->slab_cur = (uint8_t *)p;
        gs->slab_left = slab;
    }
    *out = gs->slab_cur;
    gs->slab_cur += bytes;
    gs->slab_left -= bytes;
    return PCQ_OK;
}

int pcq_grid_scan(pcq_ctx *ctx, pcq_collector *c, const DevCols &cols_in, const DevPred &pred, hipStream_t s) {
    if (cols_in.n == 0) return PCQ_OK;
    if (!c->gs) c->gs = new GridState();
    GridState *gs = c->gs;
    uint64_t budget = ctx->grid_pending_budget > 0 ? (uint64_t)ctx->grid_pending_budget : (384ull << 20);  // points: 7.7 - 9.2 GB of tuples
    if (budget > PENDING_MAX) budget = PENDING_MAX;  // (a fold's tuple counts and offsets are 32-bit)
    for (uint64_t first = 0; first < cols_in.n; first += RUN_POINTS) {
        DevCols cols = cols_in;
        cols.n = cols_in.n - first < RUN_POINTS ? cols_in.n - first : RUN_POINTS;
        cols.first_index = cols_in.first_index + first;
        cols.xyz = cols_in.xyz ? cols_in.xyz + first * cols_in.xyz_stride : nullptr;
        cols.cls = cols_in.cls ? cols_in.cls + first * cols_in.cls_stride : nullptr;
        cols.rgb = cols_in.rgb ? cols_in.rgb + first * cols_in.rgb_stride : nullptr;
        const uint32_t ntiles = (uint32_t)((cols.n + P0_TILE - 1) / P0_TILE);
        const bool wide = cols.rgb != nullptr;
        const size_t tuple_room = (size_t)ntiles * P0_TILE * tuple_bytes(wide) + 64, dir_room = (size_t)ntiles * DIR_STRIDE * sizeof(uint16_t);
        // the entry: scans of one file share it (same scale / offset, indices within 32 bits of its base)
        auto needs_entry = [&]() {
            if (gs->entries.empty()) return true;
            const GridEntryDev &e = gs->entries.back();
            return memcmp(e.scale, cols.scale, sizeof e.scale) != 0 || memcmp(e.offset, cols.offset, sizeof e.offset) != 0 ||
                   cols.first_index < gs->entry_end || cols.first_index + cols.n - gs->entry_base.back() > 0xffffffffull;
        };
        if ((needs_entry() && gs->entries.size() == 255) || gs->runs.size() == (size_t)MAX_RUNS || (gs->pending_cap && gs->pending_cap + cols.n > budget)) {
            c->last_stream = s;
            const int frc = grid_fold(ctx, c);
            if (frc) return frc;
        }
        GridRun run{};
        run.ntiles = ntiles, run.wide = wide;
        auto alloc_run = [&]() {
            void *pt = nullptr, *pd = nullptr;
            int arc = grid_room(ctx, gs, tuple_room, &pt);
            if (!arc) arc = grid_room(ctx, gs, dir_room, &pd);
            run.tuples = (uint8_t *)pt, run.dir = (uint16_t *)pd;
            return arc;
        };
        int rc = alloc_run();
        if (rc == PCQ_ERR_NOMEM && !gs->runs.empty()) {  // no room next to what is pending: fold that first (its slabs go back to the pool)
            c->last_stream = s;
            const int frc = grid_fold(ctx, c);
            if (frc) return frc;
            rc = alloc_run();
        }
        if (rc) return rc;
        if (needs_entry()) {
            GridEntryDev e;
            for (int a = 0; a < 3; a++) e.scale[a] = cols.scale[a], e.offset[a] = cols.offset[a];
            gs->entries.push_back(e);
            gs->entry_base.push_back(cols.first_index);
        }
        gs->entry_end = cols.first_index + cols.n;
        const uint32_t entry = (uint32_t)gs->entries.size() - 1;
        const uint64_t idx_base = cols.first_index - gs->entry_base.back();
        gs->pending_cap += cols.n;
        gs->pending_tiles += ntiles;
        gs->any_wide |= wide;

        const DevGrid &g = c->grid;
        const unsigned nblocks = ntiles < (uint32_t)ctx->num_cus ? ntiles : (unsigned)ctx->num_cus;  // one workgroup per CU is resident (LDS)
        const int agg = ctx->grid_agg;
        // LAST blocks (12-byte positions at an aligned address, a class byte per point, 6-byte colours): the short index arithmetic
        const bool packed = cols.xyz_stride == 12 && ((uintptr_t)cols.xyz & 3) == 0 && (!cols.cls || cols.cls_stride == 1) && (!cols.rgb || cols.rgb_stride == 6);
#define PCQ_P0_LAUNCH(KIND, RGB, PACKED) \
    hipLaunchKernelGGL((k_p0_part<KIND, RGB, PACKED>), dim3(nblocks), dim3(P0_NT), 0, s, cols, pred, g, ntiles, run.tuples, run.dir, entry, idx_base, agg)
#define PCQ_P0(KIND)                                            \
    do {                                                        \
        if (wide && packed) PCQ_P0_LAUNCH(KIND, true, true);    \
        else if (wide) PCQ_P0_LAUNCH(KIND, true, false);        \
        else if (packed) PCQ_P0_LAUNCH(KIND, false, true);      \
        else PCQ_P0_LAUNCH(KIND, false, false);                 \
    } while (0)
        if (pred.kind == PCQ_PRED_BOUNDS) PCQ_P0(PCQ_PRED_BOUNDS);
        else if (pred.kind == PCQ_PRED_CLASS) PCQ_P0(PCQ_PRED_CLASS);
        else PCQ_P0(PCQ_PRED_BOUNDS_F64);
#undef PCQ_P0
#undef PCQ_P0_LAUNCH
        PCQ_HIP(hipGetLastError());
        gs->runs.push_back(run);
    }
    return PCQ_OK;
}

// Folds the pending runs (and the earlier winners) into a new set of winners.  Synchronises.
static int grid_fold(pcq_ctx *ctx, pcq_collector *c) {
    GridState *gs = c->gs;
    if (!gs || gs->runs.empty()) return PCQ_OK;
    hipStream_t s = ctx->stream;
    if (c->last_stream && c->last_stream != s) PCQ_HIP(hipStreamSynchronize(c->last_stream));
    Scratch tmp(ctx);
    StreamDrainOnExit drain_before_tmp(s);
    const int nruns = (int)gs->runs.size();
    const DevGrid &g = c->grid;
    if (gs->pending_cap >= (1ull << 32)) return pcq_fail(PCQ_ERR_UNSUPPORTED, "grid collector: more than 2^32 pending tuples in one fold");

    // run directory, entries, the per-bin fragment lists
    const uint32_t T = (uint32_t)gs->pending_tiles, Tp = (T + 63) & ~63u, Tp1 = (T + 1 + 63) & ~63u;
    std::vector<DevRun> hruns(nruns);
    {
        uint32_t tile0 = 0;
        for (int r = 0; r < nruns; r++) {
            hruns[r] = DevRun{gs->runs[r].tuples, gs->runs[r].dir, tile0, gs->runs[r].ntiles, gs->runs[r].wide, 0};
            tile0 += gs->runs[r].ntiles;
        }
    }
    const bool any_wide = gs->any_wide;
    DevRun *d_runs = nullptr;
    GridEntryDev *d_entries = nullptr;
    uint32_t *d_bintot = nullptr, *d_binbase = nullptr, *d_preT = nullptr;
    uint16_t *d_startT = nullptr;
    uint64_t *d_tile_addr = nullptr;
    unsigned long long *d_stats = nullptr;
    DevGrid *d_grid = nullptr;
    int rc = tmp.get(nruns, &d_runs);
    if (!rc) rc = tmp.get(256, &d_entries);
    if (!rc) rc = tmp.get(F1, &d_bintot);
    if (!rc) rc = tmp.get(F1 + 1, &d_binbase);
    if (!rc) rc = tmp.get(8, &d_stats);
    if (!rc) rc = tmp.get(1, &d_grid);
    if (!rc) rc = tmp.get((size_t)F1 * Tp, &d_startT);
    if (!rc) rc = tmp.get((size_t)F1 * Tp1, &d_preT);
    if (!rc) rc = tmp.get(T, &d_tile_addr);
    if (rc) return rc;
    PCQ_HIP(hipMemcpyAsync(d_grid, &g, sizeof g, hipMemcpyHostToDevice, s));
    GridRef gref{};
    gref.full = d_grid;
    for (int a = 0; a < 3; a++) {
        gref.f.bmin[a] = g.bmin[a], gref.f.qk[a] = g.qk[a], gref.f.qmax[a] = g.qmax[a], gref.f.guard[a] = g.guard[a];
        gref.f.mask[a] = (uint32_t)g.mask[a], gref.f.shift[a] = g.shift[a];
    }
    gref.f.cell_size = g.cell_size;
    PCQ_HIP(hipMemcpyAsync(d_runs, hruns.data(), nruns * sizeof(DevRun), hipMemcpyHostToDevice, s));
    PCQ_HIP(hipMemcpyAsync(d_entries, gs->entries.data(), gs->entries.size() * sizeof(GridEntryDev), hipMemcpyHostToDevice, s));
    EntryRef eref;
    eref.table = d_entries;
    eref.e0 = gs->entries[0];
    hipLaunchKernelGGL(k_dir_transpose, dim3((T + 63) / 64), dim3(BLOCK), 0, s, d_runs, nruns, T, Tp, Tp1, d_startT, d_preT, d_tile_addr);
    hipLaunchKernelGGL(k_bin_prefix, dim3(F1), dim3(1024), 0, s, d_preT, T, Tp1, d_bintot);
    hipLaunchKernelGGL(k_excl_scan_u32, dim3(1), dim3(1024), 0, s, d_bintot, d_binbase, (uint32_t)F1);
    PCQ_HIP(hipGetLastError());
    BinSrc src{d_preT, d_startT, d_tile_addr, T, Tp1, Tp};  // (replaced by the compacted bins below when the fragments are short)
    // How dense is the grid?  The distinct cells of two bins are counted into a global hash set — asked for here, before the
    // host knows how many tuples there are, so that ONE synchronisation brings back the tuple count and the estimate (the
    // set is sized for eight times the mean bin; its probing is bounded).  Not when the bins cannot be large anyway.
    const double old_per_bin = (double)gs->wtotal / F1;
    const bool probed = ctx->grid_f2 <= 0 && (double)gs->pending_cap / F1 + old_per_bin > BIG_DIRECT;
    PCQ_HIP(hipMemsetAsync(d_stats, 0, 64, s));
    if (probed) {
        uint64_t cap = 1024;
        while (cap < 16ull * (gs->pending_cap / F1 + 1) * PROBE_BINS) cap <<= 1;
        if (cap > (1ull << 26)) cap = 1ull << 26;
        uint64_t *d_set = nullptr;
        rc = tmp.get(cap, &d_set);
        if (rc) return rc;
        PCQ_HIP(hipMemsetAsync(d_set, 0xff, cap * 8, s));
        unsigned probe_blocks = (unsigned)(((uint64_t)T * PROBE_BINS + BLOCK - 1) / BLOCK);
        if (probe_blocks > 4096) probe_blocks = 4096;
        hipLaunchKernelGGL(k_probe_distinct, dim3(probe_blocks), dim3(BLOCK), 0, s, src, eref, g, d_set, cap - 1, d_stats);
        PCQ_HIP(hipGetLastError());
    }
    uint32_t h_tuples = 0;
    unsigned long long distinct = 0;
    PCQ_HIP(hipMemcpyAsync(&h_tuples, d_binbase + F1, 4, hipMemcpyDeviceToHost, s));
    PCQ_HIP(hipMemcpyAsync(&distinct, d_stats, 8, hipMemcpyDeviceToHost, s));
    PCQ_HIP(hipStreamSynchronize(s));  // also: the pageable sources above have been read
    const uint32_t h_probe[2] = {0, h_tuples};
    const uint64_t m = h_probe[1], w_old = gs->wtotal;
    ctx->grid_last_tuples = (int64_t)m;
    if (m == 0) {
        grid_free_pending(ctx, gs);
        return PCQ_OK;
    }
    ctx->grid_folds++;
    if (T > 2u * BIG_FB && m < 2ull * T * F1) {  // fragments of less than two tuples on average: copy the bins together first
        const uint32_t Tc = F1, Tcp = F1, Tcp1 = (F1 + 1 + 63) & ~63u;
        uint8_t *d_comp = nullptr;
        uint32_t *d_cpre = nullptr;
        uint16_t *d_cstart = nullptr;
        uint64_t *d_caddr = nullptr;
        rc = tmp.get((size_t)m * tuple_bytes(any_wide) + 64, &d_comp);
        if (!rc) rc = tmp.get((size_t)F1 * Tcp1, &d_cpre);
        if (!rc) rc = tmp.get((size_t)F1 * Tcp, &d_cstart);
        if (!rc) rc = tmp.get(Tc, &d_caddr);
        if (rc) return rc;
        const uint64_t nfrag = (uint64_t)T * F1;
        hipLaunchKernelGGL(k_bin_compact, dim3((unsigned)((nfrag + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, src, d_binbase, d_comp, any_wide ? 1u : 0u);
        hipLaunchKernelGGL(k_compact_dir, dim3(F1), dim3(BLOCK), 0, s, d_binbase, d_comp, any_wide ? 1u : 0u, Tcp1, Tcp, d_cpre, d_cstart, d_caddr);
        PCQ_HIP(hipGetLastError());
        src = BinSrc{d_cpre, d_cstart, d_caddr, Tc, Tcp1, Tcp};
        ctx->grid_compactions++;
    }

    // estimated cells per level-1 bin -> fold the bins directly, or cut them again first
    uint32_t f2 = 1;
    if (ctx->grid_f2 > 0) {
        f2 = (uint32_t)ctx->grid_f2;
    } else if (probed && (double)m / F1 + old_per_bin > BIG_DIRECT) {
        const double est = (double)distinct / PROBE_BINS + old_per_bin;
        if (est > BIG_DIRECT) {
            f2 = (uint32_t)std::ceil(est / SMALL_TARGET);
            if (f2 > F2_MAX) f2 = F2_MAX;
        }
    }

    bool level2_exact = false;
    for (int attempt = 0;; attempt++) {
        const uint32_t nparts = (uint32_t)F1 * f2;
        Scratch att(ctx);
        StreamDrainOnExit drain_before_att(s);
        bool staged_level2 = false;
        GridSeg seg2{};  // the second level's output (f2 > 1)
        const uint32_t *d_tot = d_bintot;
        const uint64_t *obase = gs->wbase;
        const uint32_t *ocount = gs->wcount;
        const uint64_t *okeys = gs->wkeys;
        RecArr orecs{gs->wrecs, gs->wrec_cap};
        const bool recut_old = w_old && gs->f2 != f2;
        if (f2 > 1 || recut_old) {
            Level2Params L{};
            L.src = src, L.entries = eref, L.g = g, L.f2 = f2, L.stats = d_stats, L.wide = any_wide;
            uint8_t *d_t2 = nullptr;
            uint32_t *d_off2 = nullptr, *d_cnt2 = nullptr;
            // one pass into regions with slack (k_level2), unless that failed for this fold or does not apply
            const uint64_t cap = (uint64_t)std::ceil((double)m / nparts * 1.3) + 64;
            const bool staged = f2 <= (uint32_t)L2_STAGED_F2 && !level2_exact && cap * nparts < (1ull << 32);
            if (f2 > 1) {
                rc = att.get((staged ? (size_t)(cap * nparts) : (size_t)m) * tuple_bytes(any_wide) + 64, &d_t2);
                if (!rc) rc = att.get((size_t)nparts + 1, &d_off2);
                if (!rc && staged) rc = att.get((size_t)nparts, &d_cnt2);
                if (rc) return rc;
                L.binbase = d_binbase, L.out = d_t2, L.off2 = d_off2, L.cnt2 = d_cnt2, L.cap = (uint32_t)cap;
            }
            uint64_t *d_okeys2 = nullptr, *d_obase2 = nullptr;
            uint8_t *d_orecs2 = nullptr;
            uint32_t *d_ooff2 = nullptr, *d_ocount2 = nullptr, *d_obin = nullptr, *d_obinbase = nullptr;
            if (recut_old) {
                rc = att.get(w_old, &d_okeys2);
                if (!rc) rc = att.get(w_old * 32, &d_orecs2);
                if (!rc) rc = att.get((size_t)nparts + 1, &d_ooff2);
                if (!rc) rc = att.get((size_t)nparts + 1, &d_obase2);
                if (!rc) rc = att.get((size_t)nparts + 1, &d_ocount2);
                if (!rc) rc = att.get(F1, &d_obin);
                if (!rc) rc = att.get(F1 + 1, &d_obinbase);
                if (rc) return rc;
                hipLaunchKernelGGL(k_old_per_bin, dim3(F1 / BLOCK), dim3(BLOCK), 0, s, gs->wcount, gs->f2, d_obin);
                hipLaunchKernelGGL(k_excl_scan_u32, dim3(1), dim3(1024), 0, s, d_obin, d_obinbase, (uint32_t)F1);
                L.okeys = gs->wkeys, L.orecs = RecArr{gs->wrecs, gs->wrec_cap}, L.obase = gs->wbase, L.ocount = gs->wcount, L.f2old = gs->f2;
                L.obinbase = d_obinbase, L.okeys2 = d_okeys2, L.orecs2 = RecArr{d_orecs2, w_old}, L.ooff2 = d_ooff2;
            }
            PCQ_HIP(hipMemsetAsync(d_stats, 0, 64, s));
            if (staged) hipLaunchKernelGGL(k_level2, dim3(F1), dim3(L2S_NT), 0, s, L);
            else hipLaunchKernelGGL(k_level2_direct, dim3(F1), dim3(L2_NT), 0, s, L);
            PCQ_HIP(hipGetLastError());
            if (f2 > 1) {
                seg2 = GridSeg{d_t2, d_off2, staged ? d_cnt2 : nullptr, any_wide ? 1u : 0u};
                uint32_t *d_tot2 = nullptr;
                rc = att.get(nparts, &d_tot2);
                if (rc) return rc;
                staged_level2 = staged;  // (whether a region was outgrown is read back with the fold's counters: the fold of truncated
                                         // partitions is wasted then, but the common case saves a synchronisation)
                hipLaunchKernelGGL(k_part_totals, dim3((nparts + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, seg2, nparts, d_tot2);
                d_tot = d_tot2;
            }
            if (recut_old) {
                hipLaunchKernelGGL(k_unpack_old_dir, dim3((nparts + 1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, d_ooff2, nparts, d_obase2, d_ocount2);
                okeys = d_okeys2, orecs = RecArr{d_orecs2, w_old}, obase = d_obase2, ocount = d_ocount2;
            }
        }
        // room for the winners
        const bool big = f2 == 1;
        const uint32_t limit = big ? BIG_LIMIT : SMALL_LIMIT;
        uint64_t wcap = m + w_old;
        if (wcap > (uint64_t)nparts * limit) wcap = (uint64_t)nparts * limit;
        if (wcap >= (1ull << 32)) return pcq_fail(PCQ_ERR_UNSUPPORTED, "grid collector: more than 2^32 cells in one fold");
        uint64_t *n_wkeys = nullptr, *n_wbase = nullptr, *d_room = nullptr, *d_pieces = nullptr, *d_piece_pre = nullptr;
        uint8_t *n_wrecs = nullptr;
        uint32_t *n_wcount = nullptr, *d_palias = nullptr, *d_pay = nullptr, *d_defer = nullptr;
        const bool dense = !big && !w_old;  // k_fold_dense first, k_fold for what it leaves
        const uint32_t npieces = (nparts + SCAN_PIECE - 1) / SCAN_PIECE;
        rc = att.get(wcap, &n_wkeys);
        if (!rc) rc = att.get(wcap * 32, &n_wrecs);
        if (!rc) rc = att.get((size_t)nparts + 1, &n_wbase);
        if (!rc) rc = att.get(nparts, &n_wcount);
        if (!rc) rc = att.get(nparts, &d_room);
        if (!rc) rc = att.get(nparts, &d_palias);
        if (!rc) rc = att.get(npieces, &d_pieces);
        if (!rc) rc = att.get((size_t)npieces + 1, &d_piece_pre);
        const uint32_t resident_wgs = (uint32_t)ctx->num_cus * (big ? 1u : 3u);  // what fits the LDS: the rest of the partitions is looped over
        if (!rc) rc = att.get((size_t)resident_wgs * (big ? BIG_SLOTS : SMALL_SLOTS) * 5, &d_pay);  // parked payloads, per resident workgroup
        if (!rc && dense) rc = att.get(nparts, &d_defer);
        if (rc) return rc;
        hipLaunchKernelGGL(k_winner_room, dim3((nparts + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, d_tot, w_old ? ocount : nullptr, nparts, limit, d_room);
        hipLaunchKernelGGL(k_scan_piece_sums, dim3(npieces), dim3(1024), 0, s, d_room, nparts, d_pieces);
        hipLaunchKernelGGL(k_excl_scan_u64, dim3(1), dim3(1024), 0, s, d_pieces, d_piece_pre, npieces);
        hipLaunchKernelGGL(k_scan_pieces, dim3(npieces), dim3(1024), 0, s, d_room, nparts, d_piece_pre, n_wbase);
        PCQ_HIP(hipMemsetAsync(d_palias, 0, (size_t)nparts * 4, s));
        PCQ_HIP(hipMemsetAsync(d_stats, 0, 40, s));  // [0 .. 5): the fold's counters ([5]: the second level's, still to be read)
        if (!(f2 > 1 || recut_old)) PCQ_HIP(hipMemsetAsync(d_stats + 5, 0, 24, s));
        FoldParams F{};
        F.src = src, F.seg = seg2, F.entries = eref, F.g = gref;
        if (w_old) F.okeys = okeys, F.orecs = orecs, F.obase = obase, F.ocount = ocount;
        F.wkeys = n_wkeys, F.wrecs = RecArr{n_wrecs, wcap}, F.wbase = n_wbase, F.wcount = n_wcount, F.palias = d_palias, F.pay_scratch = d_pay, F.stats = d_stats;
        {
            uint32_t resident = resident_wgs;
            if (resident > nparts) resident = nparts;
            if (dense) {
                F.defer_list = d_defer;
                DenseParams D{};
                D.tuples = seg2.tuples, D.wide = seg2.wide, D.off = seg2.off, D.cnt = seg2.cnt, D.entries = eref, D.g = gref;
                D.wkeys = n_wkeys, D.wrecs = RecArr{n_wrecs, wcap}, D.wbase = n_wbase, D.wcount = n_wcount, D.palias = d_palias, D.stats = d_stats, D.defer_list = d_defer;
                // two workgroups per CU (4 waves per SIMD, 103 registers, nothing spilled) — three (6 waves per SIMD, 80 registers)
                // spilled 92 bytes per lane and partition, 4.8 GB of scratch each way per file: 2.58 against 2.20 ms in one process
                uint32_t dense_wgs = (uint32_t)ctx->num_cus * 2u;
                if (dense_wgs > nparts) dense_wgs = nparts;
                hipLaunchKernelGGL((k_fold_dense<SMALL_SLOTS, DENSE_NT, DENSE_K, SMALL_LIMIT, 4>), dim3(dense_wgs), dim3(DENSE_NT), 0, s, D, nparts);
            }
            if (big) hipLaunchKernelGGL((k_fold<BIG_SLOTS, BIG_NT, BIG_K, BIG_LIMIT, true, false, 4>), dim3(resident), dim3(BIG_NT), 0, s, F, nparts);
            else hipLaunchKernelGGL((k_fold<SMALL_SLOTS, SMALL_NT, SMALL_K, SMALL_LIMIT, false, true, 3>), dim3(resident), dim3(SMALL_NT), 0, s, F, nparts);
        }
        PCQ_HIP(hipGetLastError());
        unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        PCQ_HIP(hipMemcpyAsync(st, d_stats, sizeof st, hipMemcpyDeviceToHost, s));
        PCQ_HIP(hipStreamSynchronize(s));
        if (staged_level2 && st[5]) {  // a sub-partition outgrew its region (cells with very many points): count first, then cut
            level2_exact = true;
            ctx->grid_level2_exact++;
            attempt--;
            continue;
        }
        if (f2 > 1) ctx->grid_level2++;
        if (st[1]) {  // a partition held more cells than the LDS table: more partitions
            if (f2 >= F2_MAX || attempt > 8) return pcq_fail(PCQ_ERR_UNSUPPORTED, "grid collector: a partition does not fit the LDS table at the largest fan-out");
            ctx->grid_refolds++;
            f2 = big ? (uint32_t)((BIG_LIMIT * 3 / 2 + SMALL_TARGET - 1) / SMALL_TARGET) : (f2 * 2 > F2_MAX ? F2_MAX : f2 * 2);
            continue;
        }
        if (st[2]) {  // aliased keys: gather their tuples, sort by (key, file order), replay
            PCQ_HIP(hipMemsetAsync(d_stats + 4, 0, 8, s));
            if (big) hipLaunchKernelGGL((k_alias_gather<false, true>), dim3(nparts), dim3(L2_NT), 0, s, F, (AliasItem *)nullptr, d_stats + 4);
            else hipLaunchKernelGGL((k_alias_gather<false, false>), dim3(nparts), dim3(L2_NT), 0, s, F, (AliasItem *)nullptr, d_stats + 4);
            unsigned long long na = 0;
            PCQ_HIP(hipMemcpyAsync(&na, d_stats + 4, 8, hipMemcpyDeviceToHost, s));
            PCQ_HIP(hipStreamSynchronize(s));
            if (na) {
                AliasItem *d_list = nullptr, *d_sorted = nullptr;
                rc = att.get(na, &d_list);
                if (!rc) rc = att.get(na, &d_sorted);
                if (rc) return rc;
                PCQ_HIP(hipMemsetAsync(d_stats + 4, 0, 8, s));
                if (big) hipLaunchKernelGGL((k_alias_gather<true, true>), dim3(nparts), dim3(L2_NT), 0, s, F, d_list, d_stats + 4);
                else hipLaunchKernelGGL((k_alias_gather<true, false>), dim3(nparts), dim3(L2_NT), 0, s, F, d_list, d_stats + 4);
                if (na <= ALIAS_QUADRATIC) {
                    hipLaunchKernelGGL(k_alias_rank, dim3((unsigned)((na + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, d_list, (uint64_t)na, d_sorted);
                } else {  // massive aliasing: a real sort (alias_sort.hip)
                    rc = pcq_sort_by_key_then_order(ctx, d_list, sizeof(AliasItem), na, d_sorted, s);
                    if (rc) return rc;
                }
                hipLaunchKernelGGL(k_alias_replay, dim3((unsigned)((na + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, d_sorted, (uint64_t)na, F, f2);
                PCQ_HIP(hipGetLastError());
                PCQ_HIP(hipStreamSynchronize(s));
            }
        }
        // install
        att.keep(n_wkeys), att.keep(n_wrecs), att.keep(n_wbase), att.keep(n_wcount);
        grid_free_winners(ctx, gs);
        gs->wkeys = n_wkeys, gs->wrecs = n_wrecs, gs->wbase = n_wbase, gs->wcount = n_wcount;
        gs->wrec_cap = wcap;
        gs->f2 = f2;
        gs->wtotal = st[0];
        ctx->grid_last_f2 = f2;
        break;
    }
    grid_free_pending(ctx, gs);
    return PCQ_OK;
}

// Folds what is pending now (the host layer calls it when a file is done, so that a collector kept for later holds its
// winners — a few bytes per cell — instead of a tuple per scanned point).
int pcq_grid_flush(pcq_collector *c) { return c->gs ? grid_fold(c->ctx, c) : PCQ_OK; }

int pcq_grid_drain(pcq_collector *c, pcq_point *out, uint64_t *keys_out, uint64_t cap, uint64_t *out_n) {
    pcq_ctx *ctx = c->ctx;
    hipStream_t s = ctx->stream;
    *out_n = 0;
    GridState *gs = c->gs;
    if (!gs) return PCQ_OK;
    int rc = grid_fold(ctx, c);
    if (rc) return rc;
    const uint64_t n = gs->wtotal;
    *out_n = n;
    if ((!out && !keys_out) || n == 0) return PCQ_OK;
    if (cap < n) return pcq_fail(PCQ_ERR_CAPACITY, "grid collector holds %llu points, capacity %llu", (unsigned long long)n,
                                 (unsigned long long)cap);
    Scratch tmp(ctx);
    const uint32_t nparts = (uint32_t)F1 * gs->f2;
    uint32_t *d_pre = nullptr;
    uint8_t *d_out = nullptr;
    uint64_t *d_keys = nullptr;
    rc = tmp.get((size_t)nparts + 1, &d_pre);
    if (!rc && out) rc = tmp.get(n * 31 + 16, &d_out);
    if (!rc && keys_out) rc = tmp.get(n, &d_keys);
    if (rc) return rc;
    hipLaunchKernelGGL(k_excl_scan_u32, dim3(1), dim3(1024), 0, s, gs->wcount, d_pre, nparts);
    hipLaunchKernelGGL(k_drain, dim3(nparts), dim3(BLOCK), 0, s, gs->wkeys, RecArr{gs->wrecs, gs->wrec_cap}, gs->wbase, gs->wcount, d_pre, d_out, d_keys);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && out) e = hipMemcpyAsync(out, d_out, n * 31, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && keys_out) e = hipMemcpyAsync(keys_out, d_keys, n * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return pcq_fail(PCQ_ERR_HIP, "grid drain failed: %s", hipGetErrorString(e));
    return PCQ_OK;
}
