// grid_common.h — GridSampledCollector / SparseGrid on the device (kernel K4): partition by cell key, fold in LDS.
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
//           BLOCK of tuples — 16 bytes, 16-byte aligned: {x, y, z relative to the query box's low corner, place in the
//           pending stream}, the class byte (and the second level's 16 selector bits) in the coordinates' spare top bytes;
//           24 bytes {x, y, z, place, class | colour} when there is no room or a colour column —, sorted in LDS by the level-1
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
//           32-byte record + key per cell, coalesced.  A coarse grid folds its level-1 bins directly, as a STREAM
//           (k_fold_stream, grid_fold_stream.hip: a CU's whole LDS as one 6400-slot table of {key, best distance}; no
//           window, no barrier inside a bin; a tuple above its cell's minimum is out, the others go to a survivor list
//           and one exact pass picks the earliest at the minimum; k_fold<BIG> — three barriers per chunk — is its
//           fallback for bins whose survivors outgrow the list); a denser grid first gets a second partition level (k_level2:
//           one pass into fixed regions with slack, fan-out chosen from a measured estimate of the distinct cells per
//           bin) and folds the small partitions two workgroups to a CU (k_fold_dense; k_fold<SMALL> for what that
//           leaves: earlier winners, partitions longer than a chunk).
// What the kernels had to learn about gfx950: every pass is bound by vector instructions before
// it is bound by memory unless the cell arithmetic is cut down (cell_fast); loads and stores share one in-order
// counter, so a prefetch must be waited for before the stores behind it are issued; pointers loaded from memory make
// flat loads, which also hold every LDS wait; a device-scope fence writes the L2 back; registers spilled to scratch are
// HBM traffic (the dense fold: 4.8 GB each way per file until it ran with more registers and fewer waves).  Round 4: a wave64
// vector instruction occupies its SIMD for four cycles — the folds are bound by their instruction COUNT (0.5 - 0.75 of the issue
// slots), and a third of it was overhead: scalars parked in vector lanes (karg), a search per tuple (wave_max_scan), 64-bit
// multiplies (cell_hash); a load in a branch, or behind a store, waits for everything in flight.
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
#pragma once
#include <algorithm>
#include <cmath>

#include "dev_common.h"

// Everything the grid collector's translation units share: constants, the tuple and directory layouts, the cell / key /
// distance arithmetic (THE definition every pass uses), the fragment-window reader, and the kernels' parameter blocks.
// The kernels live in grid_pass0.hip (one reading of the points), grid_dir.hip (directory transposition, prefixes, compaction,
// density probe), grid_level2.hip (second partition level), grid_fold_stream.hip (coarse grids: the streaming fold), grid_fold.hip
// (the other LDS folds), grid_finish.hip (exact replay of aliased keys, drain); grid_host.hip drives them.
namespace pcqgrid {
using namespace pcqdev;

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
static_assert((P0_TILE * 16) % 16 == 0 && (P0_TILE * 24) % 16 == 0, "a tile's block starts 16-byte aligned and holds whole 16-byte words");
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
// (k_level2: 1024 or 512 threads per workgroup, tiles of four tuples per thread, a reader's window of two fragments per thread)
constexpr int L2_STAGED_F2 = 1024;     // largest fan-out of the staged form (its per-tile tables live in LDS)
constexpr int PROBE_BINS = 2;          // bins whose distinct cells are counted to estimate the grid's density
constexpr uint64_t ALIAS_QUADRATIC = 8192;  // aliased tuples up to which the replay order comes from the quadratic rank kernel
constexpr int MAX_RUNS = 1024;         // pending pass-0 runs per collector before a fold is forced
constexpr uint64_t RUN_POINTS = 1ull << 30;  // points per pass-0 run
constexpr uint64_t PENDING_MAX = (1ull << 32) - RUN_POINTS - 1;  // tuple counts and offsets of a fold are 32-bit

constexpr uint8_t R_HAS = 1;    // byte 31 of a winner record: the record holds a point
constexpr uint8_t R_ALIAS = 2;  // the key has seen a point whose unmasked cell differs from the masked one (sticky)

// One matched point on its way to the fold, as the kernels hold it.  `idx` is the tuple's place in the pending stream
// (tile among all pending tiles x 5120 + place in the tile): file order among the tuples of one fold, and — through the
// tile — the entry (scale / offset / packing) it came with.
// In memory a tuple is 16 bytes, 16-byte aligned — ONE vector load or store, and the only size at which the pieces the
// partition passes read (a bin's ~10 tuples per tile) and write (a sub-partition's ~17 tuples per staged tile) move at the
// rate of a stream: profiles/r04_gather_bench.log has 160-byte aligned pieces at 4.6 - 5.5 TB/s against 3.8 for 200-byte
// pieces of 20-byte tuples at a 4-byte phase, and 256-byte aligned runs written at 5.1 TB/s against 3.2 for 320-byte runs
// of 20-byte tuples —
//     {x - lo.x, y - lo.y, z - lo.z, idx},  the class byte in the top byte of the coordinate of ONE axis
// which needs an axis on which every match satisfies x - lo < 2^24: the narrowest side of the query box in the file's integer
// coordinates (last.rs:98-109), e.g. 9.3 M of ca13 XL's z range; a class query (every match has THE class) stores no class.
// Otherwise — a box wider than 2^24 units on every axis, a world-space predicate, a scan with a colour column — 24 bytes:
//     {x, y, z, idx, class | red << 16, green | blue << 16}   ("wide").
struct GridTuple {
    int32_t x, y, z;
    uint32_t idx;
    uint32_t w0;  // classification | entry << 8 | red << 16
    uint32_t w1;  // green | blue << 16
};
__host__ __device__ __forceinline__ uint32_t tuple_bytes(bool wide) { return wide ? 24u : 16u; }

// What turns a tuple's integers back into a position: the header scale / offset of the file it came from
// (last.rs:156-160), and how its 16-byte tuples are packed.  Consecutive scans that agree in all of it share one entry.
struct GridEntryDev {
    double scale[3], offset[3];
    int32_t lo[3];       // 16-byte tuples: subtracted from the coordinates (the query box's low corner; 0 for a class query)
    uint32_t cmask[3];   // 16-byte tuples: the bits of each stored word that are coordinate (0x00ffffff where the top byte carries something)
    uint32_t cls_const;  // 16-byte tuples without a class byte: the class of every tuple
    uint32_t fmt;        // 16-byte tuples: what the three top bytes T = x.top | y.top << 8 | z.top << 16 carry — bits 0-7: shift of the
                         // class byte in T (0xff: none, the class is cls_const); bits 8-15 / 16-23: shifts of the low / high byte of
                         // the second level's selector (sel16_of; 0xff: not stored) — pass 0 has the hash at hand, and when all three
                         // sides of the query box are below 2^24 there are two top bytes to spare
};
constexpr uint32_t FMT_NONE = 0xffu;
__host__ __device__ __forceinline__ uint32_t fmt_cls_shift(uint32_t fmt) { return fmt & 0xffu; }
__host__ __device__ __forceinline__ uint32_t fmt_sel_lo_shift(uint32_t fmt) { return (fmt >> 8) & 0xffu; }
__host__ __device__ __forceinline__ uint32_t fmt_sel_hi_shift(uint32_t fmt) { return (fmt >> 16) & 0xffu; }
__host__ __device__ __forceinline__ bool fmt_has_sel(uint32_t fmt) { return fmt_sel_lo_shift(fmt) != FMT_NONE; }

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
    uint32_t keys_wide, _pad;
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

// The partition hash of a cell key.  Bits 63..55 pick the level-1 bin, bits 52..37 the second-level partition, bits 35..18
// the first LDS slot.  Two 32-bit multiplies of the key as a 32-bit word (Fibonacci hashing, twice): the upper word for bin
// and partition — its lower bits, which a product leaves poor, stirred with its upper ones —, the lower word's upper half for
// the slot.  (One 64-bit multiply is four 32-bit ones, each a quarter-rate instruction — 16 of a fold's ~200 vector issue
// slots per tuple, in kernels bound by those; the murmur finaliser it replaced in round 3 cost two of them and three shifts.)
// `wide` (DevGrid::keys_wide, the same for the whole launch): the grid's keys can have more than 32 bits — their upper word
// goes through a multiply of its own first.  (Folded onto the lower word as it is, the few values a narrow z range leaves
// in the upper word met the low bits of x: whole groups of keys with ONE hash, 1.9 probes per insert instead of 1.3.)
// Keys that come out as the same word share bin, partition and probe sequence; the tables compare whole keys, so that
// costs probes, never results.  Probes per insert, simulated on ca13's cells at 10 m (second level of 240) and 100 m:
// 1.28 / 1.49 — the 64-bit multiply's 1.27 / 1.50, random hashing 1.36 / 1.53.
__device__ __forceinline__ uint64_t cell_hash(uint64_t k, uint32_t wide) {
    uint32_t k32 = (uint32_t)k;
    if (wide) k32 ^= (uint32_t)(k >> 32) * 0x27d4eb2fu;
    uint32_t hi = k32 * 0x9e3779b9u;
    const uint32_t lo = k32 * 0x85ebca6bu;
    hi ^= hi >> 15;
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t bin_of(uint64_t h) { return (uint32_t)(h >> (64 - F1_BITS)); }
// the second-level partition comes from the 16 bits under the bin bits
__device__ __forceinline__ uint32_t sel16_of(uint64_t h) { return (uint32_t)(h >> 37) & 0xffffu; }
__device__ __forceinline__ uint32_t sub_from_sel16(uint32_t sel16, uint32_t f2) { return (sel16 * f2) >> 16; }
__device__ __forceinline__ uint32_t sub_of(uint64_t h, uint32_t f2) { return sub_from_sel16(sel16_of(h), f2); }
template <int NSLOT>
__device__ __forceinline__ uint32_t slot_of(uint64_t h) {
    // 18 bits x 13 bits: one full-rate 24-bit multiply.  The product stays below 2^31 on purpose: the compiler takes the
    // 24-bit multiply's result for a signed number that did not overflow (with 19 bits it shifted it arithmetically, and
    // a slot came out negative).
    static_assert(NSLOT < (1 << 13), "the product must fit 31 bits");
    return __umul24((uint32_t)(h >> 18) & 0x3ffffu, (uint32_t)NSLOT) >> 18;
}


// A kernel's by-value arguments, read again where they are used.  The arguments arrive in scalar registers at the
// kernel's start, and a fold kernel has more of them (grid, entry, pointers: 150 dwords) than the 102 scalar registers: the
// compiler parks the surplus in the lanes of a vector register and fetches every scalar back with a v_readlane — a VECTOR
// instruction — each time it is used: 27 of them per tuple in the streaming fold, a quarter of its vector instructions,
// in a kernel bound by exactly those (profiles/r04_grid_progress.txt).  The argument segment is ordinary constant memory:
// karg<T>(offset) loads a T out of it through the scalar cache, and the empty asm hides from the compiler that the
// address is the same in every round of the loop, so the load stays where it is written and its registers are free again
// behind the last use.
typedef const __attribute__((address_space(4))) uint32_t *KArgPtr;
__device__ __forceinline__ KArgPtr karg_base() {
    KArgPtr p = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
template <typename T>
__device__ __forceinline__ T karg(KArgPtr base, size_t offset) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; i++) w[i] = base[offset / 4 + i];
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}

// Cross-lane steps of one wave in the vector pipe itself (DPP: an operand modifier, no trip through the LDS crossbar that
// ds_bpermute takes).
// the value of the next lane (lane 63: `last`)
__device__ __forceinline__ uint32_t wave_next_lane(uint32_t v, uint32_t last) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
// inclusive prefix maximum over the 64 lanes (unsigned; a lane without a source keeps 0, the maximum's identity; a
// broadcast that reaches a row twice does no harm to a maximum)
__device__ __forceinline__ uint32_t wave_max_scan(uint32_t v) {
#define PCQ_DPP_MAX(ctrl) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false))
    PCQ_DPP_MAX(0x111);  // row_shr:1
    PCQ_DPP_MAX(0x112);  // row_shr:2
    PCQ_DPP_MAX(0x114);  // row_shr:4
    PCQ_DPP_MAX(0x118);  // row_shr:8
    PCQ_DPP_MAX(0x142);  // row_bcast:15 — lane 15 of a row to the whole next row
    PCQ_DPP_MAX(0x143);  // row_bcast:31 — lane 31 to the upper half
#undef PCQ_DPP_MAX
    return v;
}

// Pointers that a kernel reads out of a table in memory are "generic" to the compiler: it emits flat loads, and
// a flat load counts on the LDS counter as well — every wait for an LDS operation (each barrier of the tile loops) would
// then also wait for the tuples in flight.  Everything here lives in global memory; these say so.
#define PCQ_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ T ldg(const T *p) {
    return *(const PCQ_GLOBAL T *)p;
}
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte access at a 4-byte aligned address
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));  // a 24-byte tuple's first 16 bytes
typedef uint32_t u32x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef uint32_t u32x4_a16 __attribute__((ext_vector_type(4), aligned(16)));
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// Workgroups are dealt to the 8 XCDs round-robin (workgroup id mod 8), each XCD with an L2 of its own.  Consumers of pass 0's
// bins take them in this order, so that an XCD walks a contiguous eighth of the bins: the fragments of neighbouring bins
// are neighbours in every tile's block, and the 128-byte line two of them share is then fetched by one L2 instead of two
// (counted: the big fold fetched 1.7 x the tuples it read).  A bijection of 0 .. n for any n that is a multiple of 8.
__device__ __forceinline__ uint32_t xcd_order(uint32_t it, uint32_t n) { return n % 8 == 0 ? (it % 8) * (n / 8) + it / 8 : it; }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return (uint64_t)uni32((uint32_t)v) | ((uint64_t)uni32((uint32_t)(v >> 32)) << 32); }

// The entry table as the kernels see it: entry 0 (often the only one) travels in the kernel arguments, so that the
// common case costs no dependent global load.  A tuple does not carry its entry: its place in the pending stream says
// which tile it came from, and tile_entry[tile] which entry that tile was scanned with (read only when there are several).
struct EntryRef {
    const GridEntryDev *table;
    const uint8_t *tile_entry;  // [T]
    uint32_t multi;             // more than one entry in this fold
    uint32_t _pad;
    GridEntryDev e0;
    // (written field by field with explicit global loads: as `id == 0 ? e0 : table[id]` the compiler selects between the
    // two ADDRESSES — kernel argument segment or table — and loads six doubles through flat instructions for every tuple)
    // MULTI = false (the caller KNOWS this fold has one entry): no load at all.  Not a detail: a global load in a branch
    // makes the compiler wait for EVERY outstanding load where the branch joins (one counter, in order), so a kernel that
    // keeps the next chunk of tuples in flight while it computes must not contain one on its hot path — the streaming fold
    // waited for its prefetch at the first decode until the entry lookups became compile-time (profiles/r04_grid_progress.txt).
    template <bool MULTI = true>
    __device__ __forceinline__ GridEntryDev get(uint32_t id) const {
        GridEntryDev e = e0;
        if (MULTI && id != 0) {
            const double *src = reinterpret_cast<const double *>(table + id);
#pragma unroll
            for (int a = 0; a < 3; a++) e.scale[a] = ldg(src + a), e.offset[a] = ldg(src + 3 + a);
        }
        return e;
    }
    template <bool MULTI = true>
    __device__ __forceinline__ uint32_t entry_of(uint32_t idx) const {
        if (!MULTI) return 0u;
        return multi ? (uint32_t)ldg(tile_entry + idx / (uint32_t)P0_TILE) : 0u;
    }
    // how entry `id` packs its 16-byte tuples
    template <bool MULTI = true>
    __device__ __forceinline__ void packing(uint32_t id, int32_t (&lo)[3], uint32_t (&cmask)[3], uint32_t *cls_const, uint32_t *fmt) const {
#pragma unroll
        for (int a = 0; a < 3; a++) lo[a] = e0.lo[a], cmask[a] = e0.cmask[a];
        *cls_const = e0.cls_const;
        *fmt = e0.fmt;
        if (MULTI && id != 0) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(table + id) + offsetof(GridEntryDev, lo));
#pragma unroll
            for (int a = 0; a < 3; a++) lo[a] = (int32_t)ldg(src + a), cmask[a] = ldg(src + 3 + a);
            *cls_const = ldg(src + 6);
            *fmt = ldg(src + 7);
        }
    }
};

// A tuple from memory (16 bytes at a 16-byte aligned address, or 24 at an 8-byte aligned one) into the form the kernels hold.
// (every tuple buffer ends in 64 spare bytes)
struct RawTuple {   // the words as they lie in memory: what a pass that only moves tuples keeps in registers
    u32x4_a16 a;
    u32x2_a8 b;     // wide tuples only
};
__device__ __forceinline__ RawTuple ld_raw(const uint8_t *p, bool wide) {
    RawTuple r;
    if (!wide) {
        r.a = *(const PCQ_GLOBAL u32x4_a16 *)p;
        r.b = (u32x2_a8){0u, 0u};
    } else {
        const u32x4_a8 a = *(const PCQ_GLOBAL u32x4_a8 *)p;
        r.a = (u32x4_a16){a.x, a.y, a.z, a.w};
        r.b = *(const PCQ_GLOBAL u32x2_a8 *)(p + 16);
    }
    return r;
}
template <bool MULTI = true>
__device__ __forceinline__ GridTuple decode16(const u32x4_a16 &a, const EntryRef &entries) {
    GridTuple t;
    const uint32_t id = entries.entry_of<MULTI>(a.w);
    int32_t lo[3];
    uint32_t cm[3], cc, fmt;
    entries.packing<MULTI>(id, lo, cm, &cc, &fmt);
    t.x = (int32_t)(a.x & cm[0]) + lo[0], t.y = (int32_t)(a.y & cm[1]) + lo[1], t.z = (int32_t)(a.z & cm[2]) + lo[2];
    t.idx = a.w;
    const uint32_t T = (a.x >> 24) | ((a.y >> 24) << 8) | ((a.z >> 24) << 16), cs = fmt_cls_shift(fmt);
    t.w0 = (cs == FMT_NONE ? cc : (T >> cs) & 0xffu) | (id << 8);
    t.w1 = 0;
    return t;
}
// The second level's selector of a 16-byte tuple, when its entry stores it (fmt_has_sel): no cell, no hash.
__device__ __forceinline__ uint32_t sel16_of_raw(const u32x4_a16 &a, uint32_t fmt) {
    const uint32_t T = (a.x >> 24) | ((a.y >> 24) << 8) | ((a.z >> 24) << 16);
    return ((T >> fmt_sel_lo_shift(fmt)) & 0xffu) | (((T >> fmt_sel_hi_shift(fmt)) & 0xffu) << 8);
}
// What goes into the top bytes of a 16-byte tuple's coordinates: V = class | sel16 << 8, byte top_shift(axis) / 8 of it.
__host__ __device__ __forceinline__ uint32_t fmt_top_shift(uint32_t fmt, int axis) {
    const uint32_t here = 8u * (uint32_t)axis;
    if (fmt_cls_shift(fmt) == here) return 0u;
    if (fmt_sel_lo_shift(fmt) == here) return 8u;
    if (fmt_sel_hi_shift(fmt) == here) return 16u;
    return 24u;  // (nothing: byte 3 of V is zero)
}
template <bool MULTI = true>
__device__ __forceinline__ GridTuple decode_raw(const RawTuple &r, bool wide, const EntryRef &entries) {
    if (!wide) return decode16<MULTI>(r.a, entries);
    GridTuple t;
    t.x = (int32_t)r.a.x, t.y = (int32_t)r.a.y, t.z = (int32_t)r.a.z, t.idx = r.a.w;
    t.w0 = (r.b.x & 0xffff00ffu) | (entries.entry_of<MULTI>(r.a.w) << 8), t.w1 = r.b.y;
    return t;
}
template <bool MULTI = true>
__device__ __forceinline__ GridTuple ld_tuple(const uint8_t *p, bool wide, const EntryRef &entries) { return decode_raw<MULTI>(ld_raw(p, wide), wide, entries); }
// the same for a buffer known to hold 16-byte tuples only (the dense fold's input): no format test, one aligned load
template <bool MULTI = true>
__device__ __forceinline__ GridTuple ld_tuple16(const uint8_t *p, const EntryRef &entries) {
    const u32x4_a16 a = *(const PCQ_GLOBAL u32x4_a16 *)p;
    return decode16<MULTI>(a, entries);
}
// A tuple as it came from memory, written again (the second level, the compaction of sparse bins): the same words when the
// output has the input's size; a 16-byte tuple into a 24-byte output (some other run of the fold is wide) is decoded first.
// (A 24-byte tuple never goes into a 16-byte output: the output is wide as soon as one pending run is.)
template <bool MULTI = true>
__device__ __forceinline__ void st_raw_as(uint8_t *p, const RawTuple &r, bool wide_in, bool wide_out, const EntryRef &entries) {
    if (!wide_out) {
        *(PCQ_GLOBAL u32x4_a16 *)p = r.a;
        return;
    }
    u32x4_a8 a = {r.a.x, r.a.y, r.a.z, r.a.w};
    u32x2_a8 b = r.b;
    if (!wide_in) {
        const GridTuple t = decode16<MULTI>(r.a, entries);
        a = (u32x4_a8){(uint32_t)t.x, (uint32_t)t.y, (uint32_t)t.z, t.idx};
        b = (u32x2_a8){t.w0 & 0xffff00ffu, 0u};
    }
    *(PCQ_GLOBAL u32x4_a8 *)p = a;
    *(PCQ_GLOBAL u32x2_a8 *)(p + 16) = b;
}

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
__device__ __forceinline__ uint32_t keys_wide_of(const GridRef &g) { return g.f.keys_wide; }
__device__ __forceinline__ uint32_t keys_wide_of(const DevGrid &g) { return g.keys_wide; }
__device__ __forceinline__ uint32_t keys_wide_of(const DevGridFast &g) { return g.keys_wide; }
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
template <bool MULTI = true, typename G>
__device__ __forceinline__ TupleEval eval_tuple(const G &g, const EntryRef &entries, const GridTuple &t) {
    const GridEntryDev e = entries.get<MULTI>((t.w0 >> 8) & 0xff);
    return eval_world(g, world(t.x, e.scale[0], e.offset[0]), world(t.y, e.scale[1], e.offset[1]), world(t.z, e.scale[2], e.offset[2]));
}
// the second-level partition of a tuple, from its cell (a 16-byte tuple has no room for the 16 hash bits pass 0 had at hand)
template <bool MULTI = true, typename G>
__device__ __forceinline__ uint32_t tuple_sub(const G &g, const EntryRef &entries, const GridTuple &t, uint32_t f2) {
    return sub_of(cell_hash(eval_tuple<MULTI>(g, entries, t).key, keys_wide_of(g)), f2);
}
// The same for a tuple still in its memory form: the selector pass 0 left in it when its entry stores one, its cell otherwise.
template <bool MULTI = true, typename G>
__device__ __forceinline__ uint32_t raw_sub(const G &g, const EntryRef &entries, const RawTuple &r, bool wide, uint32_t f2) {
    if (!wide) {
        int32_t lo[3];
        uint32_t cm[3], cc, fmt;
        entries.packing<MULTI>(entries.entry_of<MULTI>(r.a.w), lo, cm, &cc, &fmt);
        if (fmt_has_sel(fmt)) return sub_from_sel16(sel16_of_raw(r.a, fmt), f2);
    }
    return tuple_sub<MULTI>(g, entries, decode_raw<MULTI>(r, wide, entries), f2);
}
// file order among the tuples of a fold: the place in the pending stream; 0 is reserved for an earlier fold's winner
__device__ __forceinline__ uint64_t ord_of(const GridTuple &t) { return (uint64_t)t.idx + 1; }

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
template <bool ANYWIDE = true>
__device__ __forceinline__ RawTuple frag_ld_raw(const uint32_t *s_pre, const uint64_t *s_addr, uint32_t nfr, uint32_t j, bool *wide) {
    const uint32_t f = frag_find(s_pre, nfr, j);
    const uint64_t a = s_addr[f];
    *wide = ANYWIDE && (a & 1);
    const uint8_t *src = reinterpret_cast<const uint8_t *>(a & ~1ull) + (uint64_t)(j - s_pre[f]) * tuple_bytes(*wide);
    if (ANYWIDE) return ld_raw(src, *wide);
    RawTuple r;
    r.a = *(const PCQ_GLOBAL u32x4_a16 *)src;
    r.b = (u32x2_a8){0u, 0u};
    return r;
}
__device__ __forceinline__ GridTuple frag_ld_tuple(const uint32_t *s_pre, const uint64_t *s_addr, uint32_t nfr, uint32_t j, const EntryRef &entries) {
    const uint32_t f = frag_find(s_pre, nfr, j);
    const uint64_t a = s_addr[f];
    const bool wide = a & 1;
    return ld_tuple(reinterpret_cast<const uint8_t *>(a & ~1ull) + (uint64_t)(j - s_pre[f]) * tuple_bytes(wide), wide, entries);
}

// body(tuple, its words in memory, whether those are 24 bytes) for every tuple of bin `bin`, some thread each, no particular
// order; whole workgroup, ends on a barrier.
template <int NT, int FB, int UNROLL, typename F>
__device__ __forceinline__ void bin_for_each(const BinSrc &S, const EntryRef &entries, uint32_t bin, uint32_t *s_pre, uint64_t *s_addr, F &&body) {
    const uint32_t total = uni32(ldg(S.preT + (size_t)bin * S.Tp1 + S.T));
    uint32_t f_lo = 0, j0 = 0;
    while (j0 < total) {  // (the same for every thread)
        const uint32_t nfr = S.T - f_lo < (uint32_t)FB ? S.T - f_lo : (uint32_t)FB;
        frag_window_fill<NT>(S, bin, f_lo, nfr, s_pre, s_addr);
        __syncthreads();
        const uint32_t wend = s_pre[nfr];
        for (uint32_t i0 = j0 + threadIdx.x; i0 < wend; i0 += NT * UNROLL) {  // the loads of UNROLL steps are issued together
            RawTuple r[UNROLL];
            bool rw[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const uint32_t j = i0 + u * NT;
                r[u] = frag_ld_raw(s_pre, s_addr, nfr, j < wend ? j : wend - 1, &rw[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
                if (i0 + u * NT < wend) body(decode_raw(r[u], rw[u], entries), r[u], rw[u]);
        }
        j0 = wend, f_lo += nfr;
        __syncthreads();  // the window is rewritten
    }
}


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


// Timing stamps (libpcq_stamps.so only: make -C csrc stamps, loaded by the tools under PCQ_LAB=stamps): cycles per phase, accumulated per wave
// and added to stats[16 + i] by lane 0; stats[31] counts the waves.
#ifdef PCQ_STAMPS
#define ST_DECL uint64_t st_last_ = __builtin_amdgcn_s_memtime(), st_acc_[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define ST(i)                                                  \
    do {                                                       \
        const uint64_t st_t_ = __builtin_amdgcn_s_memtime();   \
        st_acc_[i] += st_t_ - st_last_;                        \
        st_last_ = st_t_;                                      \
    } while (0)
#define ST_FLUSH(stats)                                                                            \
    do {                                                                                           \
        if ((threadIdx.x & 63) == 0) {                                                             \
            for (int i_ = 0; i_ < 15; i_++) atomicAdd(&(stats)[16 + i_], (unsigned long long)st_acc_[i_]); \
            atomicAdd(&(stats)[31], 1ull);                                                         \
        }                                                                                          \
    } while (0)
#else
#define ST_DECL
#define ST(i)
#define ST_FLUSH(stats)
#endif

// ---------------------------------------------------------------------------------------------------------------
// parameter blocks and kernels (defined in the grid_*.hip files, launched by grid_host.hip)
// ---------------------------------------------------------------------------------------------------------------
struct P0Pack {
    int32_t lo[3];
    uint32_t cmask[3];
    uint32_t top_shift[3];  // which byte of V = class | sel16 << 8 rides in each coordinate's top byte (fmt_top_shift)
    uint32_t block_bytes;   // from one tile's block to the next
};
// pass 0's arguments (one struct: the kernel reads its parts out of the argument segment by offset — karg())
struct P0Args {
    DevCols c;
    DevPred pr;
    DevGrid g;
    P0Pack pk16;
    uint8_t *out;      // the tile blocks
    uint16_t *dir;     // the directory rows
    uint32_t ntiles, tile0;
    int agg_mode;
};

struct DevRun {        // one pending pass-0 run
    const uint8_t *tuples;
    const uint16_t *dir;
    uint32_t tile0;    // its first tile among all pending tiles
    uint32_t ntiles;
    uint32_t wide;
    uint32_t entry;    // the entry its tiles were scanned with
    uint32_t block_bytes, _pad;  // from one tile's block to the next
};

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

constexpr int SCAN_PIECE = 4096;

// grid_pass0.hip
template <int KIND, bool RGB, bool PACKED, bool WIDE>
__global__ void k_p0_part(P0Args A);
// grid_dir.hip
__global__ void k_dir_transpose(const DevRun *__restrict__ runs, int nruns, uint32_t T, uint32_t Tp, uint32_t Tp1, uint16_t *__restrict__ startT,
                                uint32_t *__restrict__ preT, uint64_t *__restrict__ tile_addr, uint8_t *__restrict__ tile_entry);
__global__ void k_bin_prefix(uint32_t *__restrict__ preT, uint32_t T, uint32_t Tp1, uint32_t *__restrict__ bintot);
__global__ void k_bin_compact(BinSrc S, EntryRef entries, const uint32_t *__restrict__ binbase, uint8_t *__restrict__ comp, uint32_t wide_out);
__global__ void k_compact_dir(const uint32_t *__restrict__ binbase, const uint8_t *__restrict__ comp, uint32_t wide, uint32_t Tp1, uint32_t Tp,
                              uint32_t *__restrict__ preT, uint16_t *__restrict__ startT, uint64_t *__restrict__ tile_addr);
__global__ void k_part_totals(GridSeg sg, uint32_t nparts, uint32_t *__restrict__ tot);
__global__ void k_excl_scan_u32(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n);
__global__ void k_excl_scan_u64(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t n);
__global__ void k_winner_room(const uint32_t *__restrict__ tot, const uint32_t *__restrict__ ocount, uint32_t nparts, uint32_t limit,
                              uint64_t *__restrict__ room);
__global__ void k_scan_piece_sums(const uint64_t *__restrict__ in, uint32_t n, uint64_t *__restrict__ sums);
__global__ void k_scan_pieces(const uint64_t *__restrict__ in, uint32_t n, const uint64_t *__restrict__ piece_prefix, uint64_t *__restrict__ out);
__global__ void k_probe_distinct(BinSrc S, EntryRef entries, DevGrid g, uint64_t *__restrict__ set, uint64_t mask, unsigned long long *__restrict__ distinct);
__global__ void k_fold_numbers(const uint32_t *__restrict__ tuples, const unsigned long long *__restrict__ stats, uint64_t *__restrict__ h_out);
__global__ void k_words_out(const uint64_t *__restrict__ src, uint64_t *__restrict__ h_out, uint32_t n);
// grid_level2.hip
__global__ void k_level2_direct(Level2Params P);
template <bool ANYWIDE, bool MULTI, int NT>
__global__ void k_level2(Level2Params P);
__global__ void k_unpack_old_dir(const uint32_t *__restrict__ ooff2, uint32_t nparts, uint64_t *__restrict__ obase2, uint32_t *__restrict__ ocount2);
__global__ void k_old_per_bin(const uint32_t *__restrict__ ocount, uint32_t f2old, uint32_t *__restrict__ obin);
// grid_fold.hip
template <int NSLOT, int NT, int FOLD_K, int LIMIT, bool BINS, bool DIRECT, int MIN_WAVES>
__global__ void k_fold(FoldParams P, uint32_t nparts);
template <int NSLOT, int NT, int FOLD_K, int LIMIT, int MIN_WAVES, bool WIDE, bool MULTI>
__global__ void k_fold_dense(DenseParams P, uint32_t nparts);
// grid_fold_stream.hip
template <int NSLOT, int NT, int LIMIT, int U, bool ANYWIDE, bool MULTI>
__global__ void k_fold_stream(FoldParams P, uint32_t nparts, uint32_t surv_cap, uint4 *__restrict__ surv_scratch);
// grid_finish.hip
template <bool EMIT, bool BINS>
__global__ void k_alias_gather(FoldParams P, AliasItem *__restrict__ list, unsigned long long *__restrict__ cursor);
__global__ void k_alias_rank(const AliasItem *__restrict__ list, uint64_t n, AliasItem *__restrict__ sorted);
__global__ void k_alias_replay(const AliasItem *__restrict__ sorted, uint64_t n, FoldParams P, uint32_t f2);
__global__ void k_drain(const uint64_t *__restrict__ wkeys, RecArr wrecs, const uint64_t *__restrict__ wbase, const uint32_t *__restrict__ wcount,
                        const uint32_t *__restrict__ dpre, uint8_t *__restrict__ out31, uint64_t *__restrict__ keys_out);

}  // namespace pcqgrid
