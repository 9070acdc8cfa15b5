// grid.hip — GridSampledCollector / SparseGrid on the device (kernel K4).
//
// Restates query/src/grid_sampling.rs:49-105 (SparseGrid::insert_point) for a massively parallel
// device.  The reference folds points sequentially into a HashMap<u64, Point>: a cell keeps the point
// closest to the cell centre, replaced only when a later point is STRICTLY closer, so the earliest
// point in file order wins ties.  For every cell key whose points all fall into the same unmasked
// cell (always, except the mask-aliasing case below) that fold is the lexicographic arg-min of
// (squared distance, file-order index), computed here in three order-independent passes over the
// matched points against an open-addressing hash table in HBM:
// (one 32-byte slot per cell: key, distance, winner index, flags — one random access per probe):
//   A  insert key; atomicMin of the f64 distance bits (monotone for d >= 0); a slot whose minimum
//      was lowered in this scan has its winner index reset;
//   B  points whose distance equals the slot minimum: atomicMin of the file-order index; the first
//      such point of a cell materialises its 31-byte Point at once (it is almost always the only one);
//   C  only when pass B saw an exact distance tie: the winners are re-derived from the final indices.
// Scans into one collector are issued in file order with increasing `first_index`, which keeps
// "first seen wins" across chunks and across files (sequential mode, main.rs:129-133).
//
// Mask aliasing (grid_sampling.rs:62-82): the key masks each axis to `bits`, but the cell centre
// uses the UNMASKED cell, so a cell >= 2^bits folds onto another key while comparing against a
// different centre.  For such keys the result depends on the visiting order; they are flagged in
// pass A and re-folded exactly, in file order, by pass R (one thread per flagged key).
//
// HBM-bound random access (one 8-byte atomic + probes per matched point); not reshaped into GEMMs.
#include "dev_common.h"

using namespace pcqdev;

int pcq_emit_prepare(pcq_ctx *ctx, const DevCols &cols, const DevPred &pred, uint64_t *matches, hipStream_t s);

namespace {

// The count of occupied slots is kept in OCC_SHARDS counters on separate 128-byte lines (one
// same-address atomic per inserting wave would serialise at ~88 atomics/us: 122 M new cells = 2 M waves).
constexpr int OCC_SHARDS = 256;
constexpr int OCC_STRIDE = 16;  // u64 units between shards
constexpr size_t OCC_WORDS = (size_t)OCC_SHARDS * OCC_STRIDE + 16;  // + n_alias and padding

constexpr uint8_t F_HAS_POINT = 1;  // pts[slot] holds a materialised winner
constexpr uint8_t F_ALIAS = 2;      // key has seen a point whose unmasked cell differs from the masked one

struct CellInfo {
    uint64_t key;
    uint64_t cell[3];  // unmasked
    bool alias;
};

// grid_sampling.rs:51-70
__device__ __forceinline__ CellInfo cell_of(const DevGrid &g, double px, double py, double pz) {
    const double p[3] = {px, py, pz};
    CellInfo ci;
    ci.alias = false;
    ci.key = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double num = (p[a] - g.bmin[a]) * g.dims_f[a];
        const double r = num / (g.bmax[a] - g.bmin[a]);
        const uint64_t cell = f64_as_u64(r);
        ci.cell[a] = cell;
        const uint64_t masked = cell & g.mask[a];
        ci.alias |= masked != cell;
        ci.key |= masked << g.shift[a];
    }
    return ci;
}

// grid_sampling.rs:78-95 — squared distance of (px,py,pz) to the centre of the unmasked cell.
__device__ __forceinline__ double centre_dist(const DevGrid &g, const uint64_t (&cell)[3], double px, double py,
                                              double pz) {
    const double cx = ((double)cell[0] + 0.5) * g.cell_size + g.bmin[0];
    const double cy = ((double)cell[1] + 0.5) * g.cell_size + g.bmin[1];
    const double cz = ((double)cell[2] + 0.5) * g.cell_size + g.bmin[2];
    const double dx = px - cx, dy = py - cy, dz = pz - cz;
    const double a = dx * dx, b = dy * dy, c = dz * dz;
    return (a + b) + c;
}

// A table sized from a guess (see pcq_grid_scan) may fill up.  An insert that finds its occupancy shard beyond
// `shard_limit` (the shards fill evenly, so this is "load factor beyond ~0.6"), or that has probed `probe_limit`
// slots, or sees that another insert has given up, raises the overflow word (n_alias[2]) and gives up — the host
// then re-runs the pass against a table of the guaranteed size (both limits 2^64-1 there: never gives up).
__device__ __forceinline__ uint64_t find_or_insert(const DevGridTable &t, uint64_t key, uint64_t probe_limit, uint64_t shard_limit) {
    const uint64_t m = t.cap - 1;
    uint64_t h = hash64(key) & m;
    uint64_t probes = 0;
    for (;;) {
        if ((++probes & 63) == 0 &&
            (probes >= probe_limit || __hip_atomic_load(t.n_alias + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(t.n_alias + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return PCQ_NO_INDEX;
        }
        uint64_t k = __hip_atomic_load(&t.slots[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == key) return h;
        if (k == PCQ_EMPTY_KEY) {
            const uint64_t prev = atomicCAS((unsigned long long *)&t.slots[h].key, (unsigned long long)PCQ_EMPTY_KEY,
                                            (unsigned long long)key);
            if (prev == PCQ_EMPTY_KEY) {
                const uint64_t in_shard = atomicAdd((unsigned long long *)&t.occupied[(blockIdx.x & (OCC_SHARDS - 1)) * OCC_STRIDE], 1ull);
                if (in_shard >= shard_limit) {
                    __hip_atomic_store(t.n_alias + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return PCQ_NO_INDEX;  // the key stays inserted; the re-run finds it
                }
                return h;
            }
            if (prev == key) return h;
        }
        h = (h + 1) & m;
    }
}

__device__ __forceinline__ uint64_t find_slot(const DevGridTable &t, uint64_t key) {
    const uint64_t m = t.cap - 1;
    uint64_t h = hash64(key) & m;
    for (;;) {
        const uint64_t k = t.slots[h].key;
        if (k == key) return h;
        if (k == PCQ_EMPTY_KEY) return PCQ_NO_INDEX;  // cannot happen after pass A
        h = (h + 1) & m;
    }
}

struct Matched {
    bool pass;
    double px, py, pz;
    RawPoint rp;
};

__device__ __forceinline__ Matched match_point(const DevCols &c, const DevPred &pr, uint64_t i) {
    Matched m;
    bool have = false;
    m.pass = i < c.n && eval_pred(c, pr, i, m.rp, have);
    if (m.pass) {
        if (!have) m.rp = ld_xyz(c, i);
        m.px = world(m.rp.x, c.scale[0], c.offset[0]);
        m.py = world(m.rp.y, c.scale[1], c.offset[1]);
        m.pz = world(m.rp.z, c.scale[2], c.offset[2]);
    }
    return m;
}

// The three passes stream the whole scan range but usually work on few of its points, so each thread takes
// GRID_BATCH points per step: the predicate inputs of all of them are loaded together (kind fixed at compile
// time: straight-line code), then the matches are folded one by one.
constexpr int GRID_BATCH = 4;

template <int KIND, typename F>
__device__ __forceinline__ void for_each_match(const DevCols &c, const DevPred &pr, F &&body) {
    const uint64_t step = (uint64_t)gridDim.x * BLOCK * GRID_BATCH;
    for (uint64_t base = (uint64_t)blockIdx.x * BLOCK * GRID_BATCH + threadIdx.x; base < c.n; base += step) {
        RawPoint rps[GRID_BATCH];
        bool passes[GRID_BATCH];
#pragma unroll
        for (int j = 0; j < GRID_BATCH; j++) {
            const uint64_t i = base + (uint64_t)j * BLOCK;
            passes[j] = eval_pred_kind<KIND>(c, pr, i < c.n ? i : c.n - 1, rps[j]) & (i < c.n);
        }
#pragma unroll
        for (int j = 0; j < GRID_BATCH; j++) {
            if (!passes[j]) continue;
            const uint64_t i = base + (uint64_t)j * BLOCK;
            if (KIND == PCQ_PRED_CLASS) rps[j] = ld_xyz(c, i);
            if (!body(i, rps[j])) return;
        }
    }
}

// Pass A.  Besides folding the distances it writes the scan's CANDIDATE bitmap (bit i%64 of word i/64): a matched
// point whose distance was <= the slot minimum it saw.  The minimum only falls, so every point that ends at the
// final minimum is a candidate; for a coarse grid that is a few points per cell (the harmonic number of the cell's
// population), and pass B reads only those.
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_grid_pass_a(DevCols c, DevPred pr, DevGrid g, DevGridTable t, uint64_t probe_limit,
                                                       uint64_t shard_limit, uint64_t *__restrict__ cand) {
    const uint64_t step = (uint64_t)gridDim.x * BLOCK * GRID_BATCH;
    const int lane = threadIdx.x & 63;
    // wave-uniform trip count (the wave's first index decides): the ballots below need the whole wave
    for (uint64_t base = (uint64_t)blockIdx.x * BLOCK * GRID_BATCH + threadIdx.x; base - lane < c.n; base += step) {
        RawPoint rps[GRID_BATCH];
        bool passes[GRID_BATCH];
#pragma unroll
        for (int j = 0; j < GRID_BATCH; j++) {
            const uint64_t i = base + (uint64_t)j * BLOCK;
            passes[j] = eval_pred_kind<KIND>(c, pr, i < c.n ? i : c.n - 1, rps[j]) & (i < c.n);
        }
#pragma unroll
        for (int j = 0; j < GRID_BATCH; j++) {
            const uint64_t i = base + (uint64_t)j * BLOCK;
            const bool pass = passes[j];
            bool candidate = false;
            if (pass) {
                if (KIND == PCQ_PRED_CLASS) rps[j] = ld_xyz(c, i);
                const RawPoint rp = rps[j];
                const double px = world(rp.x, c.scale[0], c.offset[0]), py = world(rp.y, c.scale[1], c.offset[1]),
                             pz = world(rp.z, c.scale[2], c.offset[2]);
                const CellInfo ci = cell_of(g, px, py, pz);
                const uint64_t db = (uint64_t)__double_as_longlong(centre_dist(g, ci.cell, px, py, pz));
                const uint64_t h = find_or_insert(t, ci.key, probe_limit, shard_limit);
                if (h == PCQ_NO_INDEX) return;  // table full: the pass is re-run (bitmap included) after the table has grown
                // the slot's line was just read for the key; a point that cannot lower the minimum (most points of a
                // coarse grid) skips the atomic.  A stale value can only be too large: an atomic more, never a miss.
                const uint64_t seen = __hip_atomic_load(&t.slots[h].dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                candidate = db <= seen;
                if (db < seen) {
                    const uint64_t old = atomicMin((unsigned long long *)&t.slots[h].dist, (unsigned long long)db);
                    if (db < old) t.slots[h].widx = PCQ_NO_INDEX;
                }
                if (ci.alias) {
                    t.slots[h].nflags &= ~F_ALIAS;  // racing writers all clear the same bit; bit0 is not written in pass A
                    atomicAdd((unsigned long long *)t.n_alias, 1ull);
                }
            }
            const uint64_t word = __ballot(candidate);
            if (lane == 0 && i < c.n) cand[i >> 6] = word;  // i of lane 0 is the wave's first index, a multiple of 64
        }
    }
}

// Pass B: the candidates whose distance equals the slot's final minimum race for the lowest file-order index.  It
// also materialises: the FIRST such point of a cell in this scan (atomicMin found "no index") writes its record
// right away — for almost every cell it is the only point at the minimum distance.  A second one (an exact
// distance tie, or a tie with the winner of an earlier scan) only bumps `ties`; pass C, which re-derives the
// winners from the final indices, then runs for that scan only.  One wave per bitmap word, GRID_BATCH words a step.
__global__ __launch_bounds__(BLOCK) void k_grid_pass_b(DevCols c, DevGrid g, DevGridTable t, const uint64_t *__restrict__ cand) {
    const int lane = threadIdx.x & 63;
    const uint64_t words = (c.n + 63) >> 6;
    const uint64_t wave = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nwaves = (uint64_t)gridDim.x * WAVES;
    for (uint64_t w0 = wave * GRID_BATCH; w0 < words; w0 += nwaves * GRID_BATCH) {
        uint64_t bits[GRID_BATCH];
#pragma unroll
        for (int j = 0; j < GRID_BATCH; j++) bits[j] = w0 + j < words ? cand[w0 + j] : 0;
        RawPoint rps[GRID_BATCH];
#pragma unroll
        for (int j = 0; j < GRID_BATCH; j++) {  // lanes that are not candidates re-read the word's first point
            const bool on = (bits[j] >> lane) & 1;
            rps[j] = ld_xyz(c, on ? (w0 + j) * 64 + lane : (w0 + j < words ? (w0 + j) * 64 : 0));
        }
#pragma unroll
        for (int j = 0; j < GRID_BATCH; j++) {
            if (!((bits[j] >> lane) & 1)) continue;
            const uint64_t i = (w0 + j) * 64 + lane;
            const RawPoint rp = rps[j];
            const double px = world(rp.x, c.scale[0], c.offset[0]), py = world(rp.y, c.scale[1], c.offset[1]),
                         pz = world(rp.z, c.scale[2], c.offset[2]);
            const CellInfo ci = cell_of(g, px, py, pz);
            const uint64_t h = find_slot(t, ci.key);
            if (h == PCQ_NO_INDEX) continue;
            const double d = centre_dist(g, ci.cell, px, py, pz);
            if ((uint64_t)__double_as_longlong(d) != t.slots[h].dist) continue;
            const uint64_t prev = atomicMin((unsigned long long *)&t.slots[h].widx, (unsigned long long)(c.first_index + i));
            if (prev == PCQ_NO_INDEX) {
                if (t.slots[h].nflags & F_ALIAS) {  // inverted flag: set = NOT aliased (aliased keys belong to pass R)
                    pcq_point pt;
                    make_point(c, i, rp, pt);
                    store_point_slot32(t.pts + h * 32, pt);
                    t.slots[h].nflags &= ~F_HAS_POINT;
                }
            } else {
                atomicAdd((unsigned long long *)(t.n_alias + 1), 1ull);
            }
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_grid_pass_c(DevCols c, DevPred pr, DevGrid g, DevGridTable t) {
    for_each_match<KIND>(c, pr, [&](uint64_t i, const RawPoint &rp) {
        const double px = world(rp.x, c.scale[0], c.offset[0]), py = world(rp.y, c.scale[1], c.offset[1]),
                     pz = world(rp.z, c.scale[2], c.offset[2]);
        const CellInfo ci = cell_of(g, px, py, pz);
        const uint64_t h = find_slot(t, ci.key);
        if (h == PCQ_NO_INDEX) return true;
        if (!(t.slots[h].nflags & F_ALIAS)) return true;  // aliased key: resolved by pass R
        if (t.slots[h].widx == c.first_index + i) {
            pcq_point pt;
            make_point(c, i, rp, pt);
            store_point_slot32(t.pts + h * 32, pt);
            t.slots[h].nflags &= ~F_HAS_POINT;
        }
        return true;
    });
}

// Pass R: exact sequential fold for aliased keys.  `list` / `lkeys` hold, in file order, the local
// index and the key of this scan's matched points whose key is flagged.  The thread of the FIRST list
// entry of a key owns that key: it walks the rest of the list and applies insert_point
// (grid_sampling.rs:72-103) to the entries of its key, starting from the slot's state before the scan.
__global__ __launch_bounds__(BLOCK) void k_grid_pass_r(DevCols c, DevGrid g, DevGridTable t,
                                                       const uint64_t *__restrict__ list,
                                                       const uint64_t *__restrict__ lkeys, uint64_t nlist) {
    const uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= nlist) return;
    const uint64_t key = lkeys[e];
    for (uint64_t q = 0; q < e; q++)
        if (lkeys[q] == key) return;  // an earlier entry owns this key
    const uint64_t h = find_slot(t, key);
    if (h == PCQ_NO_INDEX) return;
    bool has = !(t.slots[h].nflags & F_HAS_POINT);
    pcq_point cur;
    if (has) {
        const uint8_t *s = t.pts + h * 32;
        uint8_t *d = reinterpret_cast<uint8_t *>(&cur);
        for (int k = 0; k < 31; k++) d[k] = s[k];
    }
    for (uint64_t q = e; q < nlist; q++) {
        if (lkeys[q] != key) continue;
        const uint64_t i = list[q];
        const RawPoint rp = ld_xyz(c, i);
        const double px = world(rp.x, c.scale[0], c.offset[0]), py = world(rp.y, c.scale[1], c.offset[1]),
                     pz = world(rp.z, c.scale[2], c.offset[2]);
        bool take;
        if (!has) {
            take = true;  // grid_sampling.rs:73-76
        } else {          // :77-103 — both distances against the NEW point's (unmasked) cell centre
            const CellInfo ci = cell_of(g, px, py, pz);
            const double cur_d = centre_dist(g, ci.cell, cur.x, cur.y, cur.z);
            const double new_d = centre_dist(g, ci.cell, px, py, pz);
            take = new_d < cur_d;
        }
        if (take) {
            make_point(c, i, rp, cur);
            has = true;
        }
    }
    if (has) {
        store_point31(t.pts + h * 32, cur);
        t.slots[h].nflags &= ~F_HAS_POINT;
    }
}

// Selector for pass R's list: matched && key flagged (two-pass stable compaction of local indices).
__global__ __launch_bounds__(BLOCK) void k_alias_tile_counts(DevCols c, DevPred pr, DevGrid g, DevGridTable t,
                                                             uint64_t *__restrict__ counts) {
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    uint32_t cnt = 0;
    for (int j = 0; j < ITEMS; j++) {
        const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
        const Matched m = match_point(c, pr, i);
        bool sel = false;
        if (m.pass) {
            const CellInfo ci = cell_of(g, m.px, m.py, m.pz);
            const uint64_t h = find_slot(t, ci.key);
            sel = h != PCQ_NO_INDEX && !(t.slots[h].nflags & F_ALIAS);
        }
        cnt += (uint32_t)__popcll(__ballot(sel));
    }
    __shared__ uint32_t s_w[WAVES];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tt = 0;
        for (int i = 0; i < WAVES; i++) tt += s_w[i];
        counts[blockIdx.x] = tt;
    }
}

__global__ __launch_bounds__(BLOCK) void k_alias_emit(DevCols c, DevPred pr, DevGrid g, DevGridTable t,
                                                      const uint64_t *__restrict__ offsets, uint64_t *__restrict__ list,
                                                      uint64_t *__restrict__ lkeys) {
    __shared__ uint32_t s_w[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    uint64_t run = offsets[blockIdx.x];
    for (int j = 0; j < ITEMS; j++) {
        const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
        const Matched m = match_point(c, pr, i);
        bool sel = false;
        uint64_t key = 0;
        if (m.pass) {
            const CellInfo ci = cell_of(g, m.px, m.py, m.pz);
            const uint64_t h = find_slot(t, ci.key);
            sel = h != PCQ_NO_INDEX && !(t.slots[h].nflags & F_ALIAS);
            key = ci.key;
        }
        const uint64_t mask = __ballot(sel);
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w = 0; w < WAVES; w++) {
            const uint32_t v = s_w[w];
            before += w < wave ? v : 0;
            all += v;
        }
        if (sel) {
            const uint64_t pos = run + before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            list[pos] = i;
            lkeys[pos] = key;
        }
        run += all;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_scan_u64(uint64_t *__restrict__ counts, uint64_t n, uint64_t *__restrict__ total_out) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < n ? counts[i] : 0;
        uint64_t incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t up = __shfl_up((unsigned long long)incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint64_t wave_off = 0;
        for (int w = 0; w < wave; w++) wave_off += s_wave[w];
        const uint64_t carry = s_carry;
        if (i < n) counts[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = s_carry;
}

// out[0] = occupied slots (sum of the shards), out[1] = aliased points, out[2] = distance ties of the last scan,
// out[3] = overflow word of the last pass A.
__global__ __launch_bounds__(OCC_SHARDS) void k_sum_occupied(const uint64_t *__restrict__ occ, const uint64_t *__restrict__ n_alias,
                                                             uint64_t *__restrict__ out) {
    __shared__ uint64_t s[OCC_SHARDS];
    s[threadIdx.x] = occ[threadIdx.x * OCC_STRIDE];
    __syncthreads();
    for (int off = OCC_SHARDS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = s[0];
        out[1] = n_alias[0];
        out[2] = n_alias[1];  // distance ties seen by pass B
        out[3] = n_alias[2];  // pass A gave up: table full
    }
}

// Re-insert every used slot of `src` into `dst` (table growth).
__global__ __launch_bounds__(BLOCK) void k_grid_rehash(DevGridTable src, DevGridTable dst) {
    const uint64_t nthreads = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t h = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; h < src.cap; h += nthreads) {
        const GridSlot sl = src.slots[h];
        const uint64_t key = sl.key;
        if (key == PCQ_EMPTY_KEY) continue;
        const uint64_t m = dst.cap - 1;
        uint64_t d = hash64(key) & m;
        for (;;) {
            const uint64_t prev = atomicCAS((unsigned long long *)&dst.slots[d].key, (unsigned long long)PCQ_EMPTY_KEY,
                                            (unsigned long long)key);
            if (prev == PCQ_EMPTY_KEY) break;
            d = (d + 1) & m;
        }
        dst.slots[d].dist = sl.dist;
        dst.slots[d].widx = sl.widx;
        dst.slots[d].nflags = sl.nflags;
        const uint4 *sp = reinterpret_cast<const uint4 *>(src.pts + h * 32);
        uint4 *dp = reinterpret_cast<uint4 *>(dst.pts + d * 32);
        dp[0] = sp[0];
        dp[1] = sp[1];
    }
}

// Drain: deterministic slot-order compaction of the used slots into packed 31-byte points + keys
// (tile counts -> exclusive scan -> emit; same ballot-rank scheme as the buffer collector).
__global__ __launch_bounds__(BLOCK) void k_drain_tile_counts(DevGridTable t, uint64_t *__restrict__ counts) {
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    uint32_t cnt = 0;
    for (int j = 0; j < ITEMS; j++) {
        const uint64_t h = base + (uint64_t)j * BLOCK + threadIdx.x;
        const bool used = h < t.cap && t.slots[h].key != PCQ_EMPTY_KEY && !(t.slots[h].nflags & F_HAS_POINT);
        cnt += (uint32_t)__popcll(__ballot(used));
    }
    __shared__ uint32_t s_w[WAVES];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tt = 0;
        for (int i = 0; i < WAVES; i++) tt += s_w[i];
        counts[blockIdx.x] = tt;
    }
}

__global__ __launch_bounds__(BLOCK) void k_drain_emit(DevGridTable t, const uint64_t *__restrict__ offsets,
                                                      uint8_t *__restrict__ out31, uint64_t *__restrict__ keys_out) {
    __shared__ uint32_t s_w[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    uint64_t run = offsets[blockIdx.x];
    for (int j = 0; j < ITEMS; j++) {
        const uint64_t h = base + (uint64_t)j * BLOCK + threadIdx.x;
        const bool used = h < t.cap && t.slots[h].key != PCQ_EMPTY_KEY && !(t.slots[h].nflags & F_HAS_POINT);
        const uint64_t mask = __ballot(used);
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w = 0; w < WAVES; w++) {
            const uint32_t v = s_w[w];
            before += w < wave ? v : 0;
            all += v;
        }
        if (used) {
            const uint64_t pos = run + before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (out31) {
                const uint8_t *sp = t.pts + h * 32;
                uint8_t *dp = out31 + pos * 31;
                for (int k = 0; k < 31; k++) dp[k] = sp[k];
            }
            if (keys_out) keys_out[pos] = t.slots[h].key;
        }
        run += all;
        __syncthreads();
    }
}

}  // namespace

static void table_free(DevGridTable &t) {
    if (t.slots) (void)hipFree(t.slots);
    if (t.pts) (void)hipFree(t.pts);
    if (t.occupied) (void)hipFree(t.occupied);
    t = DevGridTable{};
}

void pcq_grid_cache_clear(pcq_ctx *ctx) {
    for (DevGridTable &e : ctx->grid_cache) table_free(e);
}

// Retires the collector's table.  The context keeps up to two retired tables (device allocations of this size cost
// from tens of milliseconds to more than a second): a process that alternates coarse and dense grids finds both
// its small and its large table again.  A third size replaces the smaller of the two if it is larger.
void pcq_grid_release(pcq_collector *c) {
    DevGridTable &t = c->table;
    pcq_ctx *ctx = c->ctx;
    if (!t.slots) return;
    if (ctx) {
        DevGridTable *victim = nullptr;
        for (DevGridTable &e : ctx->grid_cache) {
            if (!e.slots) {
                victim = &e;
                break;
            }
            if (!victim || e.cap < victim->cap) victim = &e;
        }
        if (!victim->slots || victim->cap < t.cap) {
            table_free(*victim);
            *victim = t;
            t = DevGridTable{};
            return;
        }
    }
    table_free(t);
}

static int table_alloc(pcq_ctx *ctx, DevGridTable *t, uint64_t cap, hipStream_t s) {
    *t = DevGridTable{};
    DevGridTable *hit = nullptr;  // the smallest retired table that is large enough, and not wastefully larger
    for (DevGridTable &e : ctx->grid_cache)
        if (e.slots && e.cap >= cap && e.cap <= 4 * cap && (!hit || e.cap < hit->cap)) hit = &e;
    if (hit) {
        *t = *hit;  // reuse (a larger table only lowers the load factor)
        *hit = DevGridTable{};
        cap = t->cap;
    } else {
        t->cap = cap;
        PCQ_HIP(hipMalloc((void **)&t->slots, cap * sizeof(GridSlot)));
        PCQ_HIP(hipMalloc((void **)&t->pts, cap * 32));
        PCQ_HIP(hipMalloc((void **)&t->occupied, OCC_WORDS * 8));
        t->n_alias = t->occupied + (size_t)OCC_SHARDS * OCC_STRIDE;
    }
    PCQ_HIP(hipMemsetAsync(t->slots, 0xff, cap * sizeof(GridSlot), s));  // all-ones = empty (flags are inverted)
    PCQ_HIP(hipMemsetAsync(t->occupied, 0, OCC_WORDS * 8, s));
    return PCQ_OK;
}

static uint64_t next_pow2(uint64_t v) {
    uint64_t p = 1024;
    while (p < v) p <<= 1;
    return p;
}

static int grid_blocks(pcq_ctx *ctx, uint64_t n) {
    uint64_t want = (n + BLOCK * GRID_BATCH - 1) / (BLOCK * GRID_BATCH);
    const uint64_t cap = (uint64_t)ctx->num_cus * 16;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

// Make room for `additional` new cells: load factor <= 1/2.
static int grid_reserve(pcq_ctx *ctx, pcq_collector *c, uint64_t additional, hipStream_t s) {
    const uint64_t need = next_pow2(2 * (c->table_used_bound + additional) + 1);
    if (c->table.slots && c->table.cap >= need) return PCQ_OK;
    if (!c->table.slots) return table_alloc(ctx, &c->table, need, s);
    DevGridTable nt;
    int rc = table_alloc(ctx, &nt, need, s);
    if (rc) return rc;
    // carry the counters over, then re-insert
    PCQ_HIP(hipMemcpyAsync(nt.occupied, c->table.occupied, OCC_WORDS * 8, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_grid_rehash, dim3(grid_blocks(ctx, c->table.cap)), dim3(BLOCK), 0, s, c->table, nt);
    PCQ_HIP(hipGetLastError());
    PCQ_HIP(hipStreamSynchronize(s));
    pcq_grid_release(c);
    c->table = nt;
    return PCQ_OK;
}

int pcq_grid_scan(pcq_ctx *ctx, pcq_collector *c, const DevCols &cols, const DevPred &pred,
                  uint64_t matches_upper_bound, hipStream_t s) {
    if (cols.n == 0 || matches_upper_bound == 0) return PCQ_OK;
    // new cells <= matches, and <= the number of distinct keys the bit layout can express
    const uint64_t bitsum = c->bits[0] + c->bits[1] + c->bits[2];
    uint64_t additional = matches_upper_bound;
    if (bitsum < 62) {
        const uint64_t keyspace = 1ull << bitsum;
        const uint64_t room = keyspace > c->table_used_bound ? keyspace - c->table_used_bound : 0;
        if (additional > room) additional = room;
    }
    // Table size.  `additional` is the guaranteed bound, but a coarse grid (the paper's 100 m cells: one cell per
    // ~90 points) would then spread a few million cells over a table of gigabytes: every probe an HBM miss and a
    // memset of the whole table per file.  So the first size is a guess — an eighth of the bound — unless this
    // collector, or the previous scan of this context, has shown the grid to be dense; pass A raises the overflow
    // word if the guess was too small and is then re-run (it is idempotent) on a table of the guaranteed size.
    const uint64_t used_before = c->table_used_bound;
    bool guessing = ctx->grid_guess && !c->grid_dense && !ctx->grid_dense_hint && additional > (1ull << 20);
    int rc = grid_reserve(ctx, c, guessing ? additional / 8 : additional, s);
    if (rc) return rc;
    const int grid = grid_blocks(ctx, cols.n);
    const DevGrid &g = c->grid;
    const uint64_t cand_words = (cols.n + 63) / 64;  // candidate bitmap of this scan (context scratch, grow-only)
    if (cand_words > ctx->cand_words) {
        PCQ_HIP(hipStreamSynchronize(s));
        if (ctx->d_cand) PCQ_HIP(hipFree(ctx->d_cand));
        ctx->d_cand = nullptr;
        ctx->cand_words = 0;
        if (hipMalloc((void **)&ctx->d_cand, cand_words * 8) != hipSuccess) {
            (void)hipGetLastError();
            return pcq_fail(PCQ_ERR_NOMEM, "grid candidate bitmap: %llu bytes", (unsigned long long)(cand_words * 8));
        }
        ctx->cand_words = cand_words;
    }
    for (;;) {
        PCQ_HIP(hipMemsetAsync(c->table.n_alias, 0, 24, s));  // per scan: aliased points, distance ties, overflow
        const uint64_t shard_limit = guessing ? c->table.cap / OCC_SHARDS * 5 / 8 : ~0ull;
        const uint64_t probe_limit = guessing ? 1024ull : ~0ull;
        if (pred.kind == PCQ_PRED_BOUNDS) hipLaunchKernelGGL(k_grid_pass_a<PCQ_PRED_BOUNDS>, dim3(grid), dim3(BLOCK), 0, s, cols, pred, g, c->table, probe_limit, shard_limit, ctx->d_cand);
        else if (pred.kind == PCQ_PRED_CLASS) hipLaunchKernelGGL(k_grid_pass_a<PCQ_PRED_CLASS>, dim3(grid), dim3(BLOCK), 0, s, cols, pred, g, c->table, probe_limit, shard_limit, ctx->d_cand);
        else hipLaunchKernelGGL(k_grid_pass_a<PCQ_PRED_BOUNDS_F64>, dim3(grid), dim3(BLOCK), 0, s, cols, pred, g, c->table, probe_limit, shard_limit, ctx->d_cand);
        PCQ_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_sum_occupied, dim3(1), dim3(OCC_SHARDS), 0, s, c->table.occupied, c->table.n_alias, ctx->d_scalars + 16);
        PCQ_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars + 16, 32, hipMemcpyDeviceToHost, s));
        PCQ_HIP(hipStreamSynchronize(s));
        if (!ctx->h_scalars[3]) break;
        if (!guessing) return pcq_fail(PCQ_ERR_HIP, "grid table of the guaranteed size overflowed");
        guessing = false;
        ctx->grid_overflows++;
        c->grid_dense = true;
        c->table_used_bound = ctx->h_scalars[0];  // the cells inserted so far stay (re-inserting finds them)
        const uint64_t still = additional > ctx->h_scalars[0] - used_before ? additional - (ctx->h_scalars[0] - used_before) : 0;
        rc = grid_reserve(ctx, c, still, s);
        if (rc) return rc;
    }
    c->table_used_bound = ctx->h_scalars[0];
    if (ctx->h_scalars[1]) c->grid_has_alias = true;  // sticky: flagged keys stay flagged
    // the next scan of this context starts from a guess again only if this one would have fitted it
    ctx->grid_dense_hint = (c->table_used_bound - used_before) > additional / 8;
    if (2 * c->table_used_bound >= c->table.cap) {  // only after a guess: back to a load factor <= 1/2 for the lookups
        ctx->grid_regrows++;
        rc = grid_reserve(ctx, c, 0, s);
        if (rc) return rc;
    }
    DevGridTable &t = c->table;
    hipLaunchKernelGGL(k_grid_pass_b, dim3(grid), dim3(BLOCK), 0, s, cols, g, t, ctx->d_cand);
    PCQ_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sum_occupied, dim3(1), dim3(OCC_SHARDS), 0, s, t.occupied, t.n_alias, ctx->d_scalars + 16);
    PCQ_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars + 16, 32, hipMemcpyDeviceToHost, s));
    PCQ_HIP(hipStreamSynchronize(s));
    const uint64_t n_ties = ctx->h_scalars[2];
    if (c->grid_has_alias) {
        const uint64_t nblocks = (cols.n + TILE - 1) / TILE;
        rc = pcq_ensure_partials(ctx, (size_t)nblocks);
        if (rc) return rc;
        hipLaunchKernelGGL(k_alias_tile_counts, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, cols, pred, g, t, ctx->d_partials);
        hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s, ctx->d_partials, nblocks, ctx->d_scalars);
        PCQ_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 8, hipMemcpyDeviceToHost, s));
        PCQ_HIP(hipStreamSynchronize(s));
        const uint64_t nlist = ctx->h_scalars[0];
        if (nlist) {
            uint64_t *d_list = nullptr;
            PCQ_HIP(hipMalloc((void **)&d_list, nlist * 16));
            uint64_t *d_lkeys = d_list + nlist;
            hipLaunchKernelGGL(k_alias_emit, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, cols, pred, g, t, ctx->d_partials,
                               d_list, d_lkeys);
            hipLaunchKernelGGL(k_grid_pass_r, dim3((unsigned)((nlist + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, cols, g, t,
                               d_list, d_lkeys, nlist);
            hipError_t e = hipStreamSynchronize(s);
            (void)hipFree(d_list);
            if (e != hipSuccess) return pcq_fail(PCQ_ERR_HIP, "grid pass R failed: %s", hipGetErrorString(e));
        }
    }
    if (n_ties) {
        if (pred.kind == PCQ_PRED_BOUNDS) hipLaunchKernelGGL(k_grid_pass_c<PCQ_PRED_BOUNDS>, dim3(grid), dim3(BLOCK), 0, s, cols, pred, g, t);
        else if (pred.kind == PCQ_PRED_CLASS) hipLaunchKernelGGL(k_grid_pass_c<PCQ_PRED_CLASS>, dim3(grid), dim3(BLOCK), 0, s, cols, pred, g, t);
        else hipLaunchKernelGGL(k_grid_pass_c<PCQ_PRED_BOUNDS_F64>, dim3(grid), dim3(BLOCK), 0, s, cols, pred, g, t);
    }
    PCQ_HIP(hipGetLastError());
    return PCQ_OK;
}

int pcq_grid_drain(pcq_collector *c, pcq_point *out, uint64_t *keys_out, uint64_t cap, uint64_t *out_n) {
    pcq_ctx *ctx = c->ctx;
    hipStream_t s = ctx->stream;
    *out_n = 0;
    if (c->last_stream && c->last_stream != s) PCQ_HIP(hipStreamSynchronize(c->last_stream));
    if (!c->table.slots) return PCQ_OK;
    hipLaunchKernelGGL(k_sum_occupied, dim3(1), dim3(OCC_SHARDS), 0, s, c->table.occupied, c->table.n_alias, ctx->d_scalars + 16);
    PCQ_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars + 16, 8, hipMemcpyDeviceToHost, s));
    PCQ_HIP(hipStreamSynchronize(s));
    const uint64_t n = ctx->h_scalars[0];
    *out_n = n;
    if ((!out && !keys_out) || n == 0) return PCQ_OK;
    if (cap < n) return pcq_fail(PCQ_ERR_CAPACITY, "grid collector holds %llu points, capacity %llu", (unsigned long long)n,
                                 (unsigned long long)cap);
    uint8_t *d_out = nullptr;
    uint64_t *d_keys = nullptr;
    const uint64_t nblocks = (c->table.cap + TILE - 1) / TILE;
    int rc = pcq_ensure_partials(ctx, (size_t)nblocks);
    if (rc) return rc;
    if (out) PCQ_HIP(hipMalloc((void **)&d_out, n * 31));
    if (keys_out) PCQ_HIP(hipMalloc((void **)&d_keys, n * 8));
    hipLaunchKernelGGL(k_drain_tile_counts, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, c->table, ctx->d_partials);
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s, ctx->d_partials, nblocks, ctx->d_scalars);
    hipLaunchKernelGGL(k_drain_emit, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, c->table, ctx->d_partials, d_out, d_keys);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && out) e = hipMemcpyAsync(out, d_out, n * 31, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && keys_out) e = hipMemcpyAsync(keys_out, d_keys, n * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (d_out) (void)hipFree(d_out);
    if (d_keys) (void)hipFree(d_keys);
    if (e != hipSuccess) return pcq_fail(PCQ_ERR_HIP, "grid drain failed: %s", hipGetErrorString(e));
    return PCQ_OK;
}
