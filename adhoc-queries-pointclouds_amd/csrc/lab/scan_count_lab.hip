// scan_count.hip — the count-only fast paths of the predicate scan (kernels K1 and K2).
//
// K1 bounds_count restates the loop of search_last_file_by_bounds_optimized
//    (query/src/search/last.rs:117-135) feeding a CountCollector (collect_points.rs:83-85):
//    count of points with lmin <= (x,y,z) <= lmax over the LAST positions block, N x {i32 x,y,z}.
// K2 class_count restates search_last_file_by_classification_optimized (last.rs:253-262):
//    count of classification bytes equal to `cls` over the LAST classification block, N x u8.
//
// Both are pure HBM streaming reads (12 B/point, 1 B/point): no MFMA, no reuse.  Design for gfx950:
//  * every global load is a fully coalesced 16 B/lane access (1 KiB per wave-instruction),
//    issued non-temporal (the stream is read once);
//  * a 12-byte point is not a power of two, so a wave owns a 768-dword tile (= 256 whole points,
//    3 KiB) loaded by three dwordx4 instructions; instead of transposing through LDS, each dword is
//    range-tested against the bound of ITS component ((k + lane + j) mod 3, rotated per lane once)
//    and the 64-bit compare masks (wave64: v_cmp writes an SGPR pair) are combined with scalar
//    shifts/ANDs into "three consecutive dwords pass" bits, popcounted with s_bcnt1 — the VALU sees
//    two instructions per dword, everything else runs on the scalar unit;
//  * persistent grid with a tile-stride loop; per-block partial counts are written with plain stores and
//    folded by a 1-block finishing kernel, so no same-address atomic storm at the tail;
//  * the kernels that run by default (k_bounds_count_w1_pipe, k_bounds_count_batch_pipe,
//    k_class_count_pipe, k_class_count_batch_pipe) are ONE-WAVE workgroups, three (class: four) per CU,
//    software-pipelined by hand: the loads of the next step are in flight (inline-asm global_load_dwordx4
//    nt + counted s_waitcnt) while the current step is evaluated.  The 256-thread kernels they replaced
//    stay selectable (options k1_variant / batch_variant / class_batch_pipe) for the sweeps in tools/.
#include <vector>

#include "pcq_internal.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int TILE_POINTS = 256;  // per wave: 768 dwords = 3 x (64 lanes x 16 B)

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr uint64_t R0 = 0x9249249249249249ull;  // lanes with lane % 3 == 0
constexpr uint64_t R1 = 0x2492492492492492ull;  // lane % 3 == 1
constexpr uint64_t R2 = 0x4924924924924924ull;  // lane % 3 == 2

// lanes l for which dword (k, l, j) of a tile is the first component of a point:
// (k + l + j) % 3 == 0  <=>  l % 3 == (3 - (k + j) % 3) % 3
__device__ __forceinline__ constexpr uint64_t start_lanes(int s) {
    return (s % 3) == 0 ? R0 : ((s % 3) == 1 ? R2 : R1);
}

__device__ __forceinline__ v4i ld_nt(const v4i *p) { return __builtin_nontemporal_load(p); }

struct LaneBox {
    int lo[3];        // lo[(lane%3 + t) % 3], t = 0..2
    uint32_t w[3];
};

__device__ __forceinline__ LaneBox rotate_box(const int32_t (&lo)[3], const uint32_t (&w)[3], int lane) {
    const int r = lane % 3;
    LaneBox b;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int c = (r + t) % 3;
        b.lo[t] = c == 0 ? lo[0] : (c == 1 ? lo[1] : lo[2]);
        b.w[t] = c == 0 ? w[0] : (c == 1 ? w[1] : w[2]);
    }
    return b;
}

// Count of matching points in one 768-dword wave tile, mask-algebra form (wave-uniform result).
__device__ __forceinline__ uint32_t tile_count_regs(const v4i (&v)[3], const LaneBox &b);

template <bool NT = true>
__device__ __forceinline__ uint32_t tile_count_masks(const v4i *tile, int lane, const LaneBox &b) {
    v4i v[3];
    if (NT) {
        v[0] = ld_nt(tile + lane);
        v[1] = ld_nt(tile + 64 + lane);
        v[2] = ld_nt(tile + 128 + lane);
    } else {
        v[0] = tile[lane];
        v[1] = tile[64 + lane];
        v[2] = tile[128 + lane];
    }
    return tile_count_regs(v, b);
}

__device__ __forceinline__ uint32_t tile_count_regs(const v4i (&v)[3], const LaneBox &b) {
    uint64_t m[3][4];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int t = (k + j) % 3;
            m[k][j] = __ballot((uint32_t)(v[k][j] - b.lo[t]) <= b.w[t]);
        }
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint64_t m0 = m[k][0], m1 = m[k][1], m2 = m[k][2], m3 = m[k][3];
        // dwords 4l+4, 4l+5: lane l+1 of this load, or lane 0 of the next one.  The tile ends on a
        // point boundary, so nothing is carried out of k == 2.
        const uint64_t c0 = k < 2 ? m[k < 2 ? k + 1 : k][0] : 0ull;
        const uint64_t c1 = k < 2 ? m[k < 2 ? k + 1 : k][1] : 0ull;
        const uint64_t n0 = (m0 >> 1) | (c0 << 63);
        const uint64_t n1 = (m1 >> 1) | (c1 << 63);
        const uint64_t a = m1 & m2;
        const uint64_t t0 = m0 & a;      // dwords j=0,1,2 of lane l
        const uint64_t t1 = a & m3;      // j=1,2,3
        const uint64_t bb = m3 & n0;
        const uint64_t t2 = m2 & bb;     // j=2,3 and next lane's 0
        const uint64_t t3 = bb & n1;     // j=3 and next lane's 0,1
        const uint64_t s012 = (t0 & start_lanes(k)) | (t1 & start_lanes(k + 1)) | (t2 & start_lanes(k + 2));
        cnt += (uint32_t)__popcll(s012) + (uint32_t)__popcll(t3 & start_lanes(k + 3));
    }
    return cnt;
}

// One point per lane per load (global_load_dwordx3), 4 loads per tile.
struct __attribute__((packed, aligned(4))) P3 {
    int x, y, z;
};

__device__ __forceinline__ uint32_t tile_count_x3(const v4i *tile, int lane, const int32_t (&lo)[3],
                                                  const uint32_t (&w)[3]) {
    const P3 *pts = reinterpret_cast<const P3 *>(tile);
    P3 p[4];
#pragma unroll
    for (int q = 0; q < 4; q++) p[q] = pts[q * 64 + lane];
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const bool pass = ((uint32_t)(p[q].x - lo[0]) <= w[0]) & ((uint32_t)(p[q].y - lo[1]) <= w[1]) &
                          ((uint32_t)(p[q].z - lo[2]) <= w[2]);
        cnt += (uint32_t)__popcll(__ballot(pass));
    }
    return cnt;
}

// Four whole points per lane: 48 contiguous bytes as three 16-byte loads at a 48-byte lane stride.
__device__ __forceinline__ uint32_t tile_count_lane48(const v4i *tile, int lane, const int32_t (&lo)[3],
                                                      const uint32_t (&w)[3]) {
    const v4i a = ld_nt(tile + 3 * lane), b = ld_nt(tile + 3 * lane + 1), c = ld_nt(tile + 3 * lane + 2);
    const int d[12] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3]};
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const bool pass = ((uint32_t)(d[3 * q] - lo[0]) <= w[0]) & ((uint32_t)(d[3 * q + 1] - lo[1]) <= w[1]) &
                          ((uint32_t)(d[3 * q + 2] - lo[2]) <= w[2]);
        cnt += (uint32_t)__popcll(__ballot(pass));
    }
    return cnt;
}

__device__ __forceinline__ void block_store_partial(uint64_t wave_total, uint64_t *partials) {
    __shared__ uint64_t s_w[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) t += s_w[i];
        partials[blockIdx.x] = t;
    }
}

// K1, 256-thread kernels (k1_variant 0..7).  VARIANT 0: mask algebra · 1: dwordx3 per lane · 2: 48-byte lane stride
//      · 3: mask algebra, two tiles in flight per wave · 4: mask algebra, plain (temporal) loads
//      · 5: mask algebra, each wave owns two ADJACENT tiles (6 KiB contiguous), six loads in flight
//      · 6: mask algebra with software prefetch of the wave's next tile (compiler-scheduled)
//      · 7: the same pipeline with inline-asm loads and counted waits.
template <int VARIANT>
__global__ __launch_bounds__(BLOCK) void k_bounds_count_xyz12(const v4i *__restrict__ base, uint64_t n,
                                                              DevPred pred, uint64_t *__restrict__ partials) {
    const int lane = threadIdx.x & 63;
    const uint64_t tiles = n / TILE_POINTS;
    // readfirstlane: the wave index is uniform, so the tile loop runs on the scalar unit
    const uint64_t wave_id = (uint64_t)blockIdx.x * WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t stride = (uint64_t)gridDim.x * WAVES;
    uint64_t total = 0;  // wave-uniform
    const LaneBox lb = rotate_box(pred.lo, pred.width, lane);
    if (VARIANT == 3) {
        uint64_t t = wave_id;
        for (; t + stride < tiles; t += 2 * stride) {
            total += tile_count_masks(base + t * 192, lane, lb);
            total += tile_count_masks(base + (t + stride) * 192, lane, lb);
        }
        if (t < tiles) total += tile_count_masks(base + t * 192, lane, lb);
    } else if (VARIANT == 6) {
        // software pipeline, ping-pong registers (no copies): while tile A is evaluated the loads of
        // tile B are in flight and vice versa, so each wave keeps 3 KiB outstanding at all times.
        // The prefetch is unconditional (index clamped to the wave's current tile at the tail, an L2
        // hit) so that the compiler can count exactly which loads an evaluation has to wait for.
        if (wave_id < tiles) {
            v4i a[3], b[3];
            uint64_t t = wave_id;
#pragma unroll
            for (int k = 0; k < 3; k++) a[k] = ld_nt(base + t * 192 + 64 * k + lane);
            for (;;) {
                const uint64_t t1 = t + stride;
                const uint64_t t1c = t1 < tiles ? t1 : t;
#pragma unroll
                for (int k = 0; k < 3; k++) b[k] = ld_nt(base + t1c * 192 + 64 * k + lane);
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE the evaluation it overlaps
                total += tile_count_regs(a, lb);
                __builtin_amdgcn_sched_barrier(0);
                if (t1 >= tiles) break;
                const uint64_t t2 = t1 + stride;
                const uint64_t t2c = t2 < tiles ? t2 : t1;
#pragma unroll
                for (int k = 0; k < 3; k++) a[k] = ld_nt(base + t2c * 192 + 64 * k + lane);
                __builtin_amdgcn_sched_barrier(0);
                total += tile_count_regs(b, lb);
                __builtin_amdgcn_sched_barrier(0);
                if (t2 >= tiles) break;
                t = t2;
            }
        }
    } else if (VARIANT == 7) {
        // The same software pipeline with the loads and their waits written as inline asm: hipcc sinks
        // ordinary loads down to their first use (and rotates the loop), which serialises "prefetch B"
        // behind "evaluate A".  asm volatile statements keep their order; the counted s_waitcnt names
        // the three registers it guards as in/out operands so no use can be hoisted above it.
        if (wave_id < tiles) {
            v4i a0, a1, a2, b0, b1, b2;
#define PCQ_LOAD3(r0, r1, r2, tile_index)                                                                       \
    {                                                                                                            \
        const v4i *q_ = base + (tile_index) * 192 + lane;                                                         \
        asm volatile("global_load_dwordx4 %0, %3, off nt\n\tglobal_load_dwordx4 %1, %3, off offset:1024 nt\n\t"   \
                     "global_load_dwordx4 %2, %3, off offset:2048 nt"                                             \
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2)                                                            \
                     : "v"(q_)                                                                                    \
                     : "memory");                                                                                 \
    }
#define PCQ_WAIT3(r0, r1, r2, n) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(r0), "+v"(r1), "+v"(r2)::"memory")
            uint64_t t = wave_id;
            PCQ_LOAD3(a0, a1, a2, t);
            for (;;) {
                const uint64_t t1 = t + stride;
                const uint64_t t1c = t1 < tiles ? t1 : t;
                PCQ_LOAD3(b0, b1, b2, t1c);
                PCQ_WAIT3(a0, a1, a2, 3);  // A has landed, B's three loads stay in flight
                {
                    const v4i va[3] = {a0, a1, a2};
                    total += tile_count_regs(va, lb);
                }
                if (t1 >= tiles) break;
                const uint64_t t2 = t1 + stride;
                const uint64_t t2c = t2 < tiles ? t2 : t1;
                PCQ_LOAD3(a0, a1, a2, t2c);
                PCQ_WAIT3(b0, b1, b2, 3);
                {
                    const v4i vb[3] = {b0, b1, b2};
                    total += tile_count_regs(vb, lb);
                }
                if (t2 >= tiles) break;
                t = t2;
            }
            // the clamped tail prefetch is still in flight: land it before its registers are reused
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2)::"memory");
#undef PCQ_LOAD3
#undef PCQ_WAIT3
        }
    } else if (VARIANT == 5) {
        const uint64_t pairs = tiles / 2;
        for (uint64_t t = wave_id; t < pairs; t += stride) {
            const v4i *tile = base + t * 384;
            v4i a[3], b[3];
#pragma unroll
            for (int k = 0; k < 3; k++) a[k] = ld_nt(tile + 64 * k + lane);
#pragma unroll
            for (int k = 0; k < 3; k++) b[k] = ld_nt(tile + 192 + 64 * k + lane);
            total += tile_count_regs(a, lb);
            total += tile_count_regs(b, lb);
        }
        if ((tiles & 1) && wave_id == 0) total += tile_count_masks(base + (tiles - 1) * 192, lane, lb);
    } else {
        for (uint64_t t = wave_id; t < tiles; t += stride) {
            const v4i *tile = base + t * 192;  // 192 x 16 B = 3 KiB
            if (VARIANT == 0) total += tile_count_masks(tile, lane, lb);
            else if (VARIANT == 4) total += tile_count_masks<false>(tile, lane, lb);
            else if (VARIANT == 1) total += tile_count_x3(tile, lane, pred.lo, pred.width);
            else total += tile_count_lane48(tile, lane, pred.lo, pred.width);
        }
    }
    // ragged tail: fewer than 256 points, one per thread of the first block
    const uint64_t rem_first = tiles * TILE_POINTS;
    if (blockIdx.x == 0) {
        const uint64_t p = rem_first + threadIdx.x;
        bool pass = false;
        if (p < n) {
            const int *q = reinterpret_cast<const int *>(base) + 3 * p;
            pass = ((uint32_t)(q[0] - pred.lo[0]) <= pred.width[0]) & ((uint32_t)(q[1] - pred.lo[1]) <= pred.width[1]) &
                   ((uint32_t)(q[2] - pred.lo[2]) <= pred.width[2]);
        }
        total += (uint64_t)__popcll(__ballot(pass));
    }
    block_store_partial(total, partials);
}

// K1 with one wave per workgroup (variants 8..11): the read-only geometry sweep
// (profiles/r01_hbm_read_geometry_sweep.log) puts 64-thread blocks 1-2 % above 256-thread blocks at the same
// bytes in flight.  A wave owns TILES adjacent 3 KiB tiles per step and issues all 3 * TILES loads before it
// evaluates any of them (variant 8: 1 tile, 9: 2, 10: 3, 11: 4); option "k1_waves_per_cu" sets the number of
// such workgroups per CU.
template <int TILES>
__global__ __launch_bounds__(64) void k_bounds_count_w1(const v4i *__restrict__ base, uint64_t n, DevPred pred,
                                                        uint64_t *__restrict__ partials) {
    const int lane = threadIdx.x;
    const uint64_t tiles = n / TILE_POINTS;
    const LaneBox lb = rotate_box(pred.lo, pred.width, lane);
    uint64_t total = 0;
    const uint64_t groups = tiles / TILES;
    for (uint64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const v4i *tile = base + g * (192 * TILES);
        v4i v[TILES][3];
#pragma unroll
        for (int t = 0; t < TILES; t++)
#pragma unroll
            for (int k = 0; k < 3; k++) v[t][k] = ld_nt(tile + 192 * t + 64 * k + lane);
#pragma unroll
        for (int t = 0; t < TILES; t++) total += tile_count_regs(v[t], lb);
    }
    if (blockIdx.x == 0) {
        for (uint64_t t = groups * TILES; t < tiles; t++) total += tile_count_masks(base + t * 192, lane, lb);  // < TILES leftover tiles
        for (int k = 0; k < 4; k++) {  // ragged tail: fewer than 256 points
            const uint64_t p = tiles * TILE_POINTS + (uint64_t)(64 * k + lane);
            bool pass = false;
            if (p < n) {
                const int *q = reinterpret_cast<const int *>(base) + 3 * p;
                pass = ((uint32_t)(q[0] - pred.lo[0]) <= pred.width[0]) & ((uint32_t)(q[1] - pred.lo[1]) <= pred.width[1]) &
                       ((uint32_t)(q[2] - pred.lo[2]) <= pred.width[2]);
            }
            total += (uint64_t)__popcll(__ballot(pass));
        }
    }
    if (lane == 0) partials[blockIdx.x] = total;
}

// Batched K1: many device-resident LAST position blocks (one per file) in one launch.
__global__ __launch_bounds__(BLOCK) void k_bounds_count_batch(const DevSegment *__restrict__ segs, int nseg,
                                                              uint64_t total_tiles, uint64_t *__restrict__ partials) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave_id = (uint64_t)blockIdx.x * WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t stride = (uint64_t)gridDim.x * WAVES;
    uint64_t total = 0;
    int s = 0;
    uint64_t seg_begin = 0, seg_end = 0;  // tile range of the cached segment
    const v4i *seg_base = nullptr;
    LaneBox lb = {};
    bool seg_empty = true;
    for (uint64_t t = wave_id; t < total_tiles; t += stride) {
        if (t >= seg_end) {
            while (s + 1 < nseg && t >= segs[s + 1].tile_begin) s++;
            seg_begin = segs[s].tile_begin;
            seg_end = seg_begin + segs[s].n / TILE_POINTS;
            seg_base = reinterpret_cast<const v4i *>(segs[s].xyz);
            seg_empty = segs[s].empty != 0;
            lb = rotate_box(segs[s].lo, segs[s].width, lane);
        }
        if (!seg_empty) total += tile_count_masks(seg_base + (t - seg_begin) * 192, lane, lb);
    }
    // ragged tails: segment i's tail belongs to block i % gridDim.x
    for (int i = blockIdx.x; i < nseg; i += gridDim.x) {
        const uint64_t n = segs[i].n;
        const uint64_t p = (n / TILE_POINTS) * TILE_POINTS + threadIdx.x;
        bool pass = false;
        if (p < n && !segs[i].empty) {
            const int *q = reinterpret_cast<const int *>(segs[i].xyz) + 3 * p;
            pass = ((uint32_t)(q[0] - segs[i].lo[0]) <= segs[i].width[0]) &
                   ((uint32_t)(q[1] - segs[i].lo[1]) <= segs[i].width[1]) &
                   ((uint32_t)(q[2] - segs[i].lo[2]) <= segs[i].width[2]);
        }
        total += (uint64_t)__popcll(__ballot(pass));
    }
    block_store_partial(total, partials);
}

// Variants 12..14: one wave per workgroup, TILES adjacent tiles per step (12: 2, 13: 1, 14: 3), software-pipelined
// with inline-asm loads and counted waits (as variant 7): the loads of step i+1 are in flight while step i is
// evaluated, so a wave never has fewer than TILES x 3 KiB outstanding.  asm volatile statements keep their order;
// the empty asm behind each s_waitcnt re-defines the registers it guards, so no use can be hoisted above the wait.
template <int TILES>
struct PipeRegs {
    v4i r[TILES][3];
};
template <int TILES>
__device__ __forceinline__ void pipe_load(PipeRegs<TILES> &R, const v4i *base, uint64_t step, int lane) {
#pragma unroll
    for (int t = 0; t < TILES; t++) {
        const v4i *q = base + (step * TILES + t) * 192 + lane;
        asm volatile("global_load_dwordx4 %0, %3, off nt\n\tglobal_load_dwordx4 %1, %3, off offset:1024 nt\n\t"
                     "global_load_dwordx4 %2, %3, off offset:2048 nt"
                     : "=&v"(R.r[t][0]), "=&v"(R.r[t][1]), "=&v"(R.r[t][2])
                     : "v"(q)
                     : "memory");
    }
}
template <int TILES, int PENDING>
__device__ __forceinline__ void pipe_wait(PipeRegs<TILES> &R) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory");
#pragma unroll
    for (int t = 0; t < TILES; t++) asm volatile("" : "+v"(R.r[t][0]), "+v"(R.r[t][1]), "+v"(R.r[t][2])::"memory");
}
template <int TILES>
__device__ __forceinline__ uint64_t pipe_eval(const PipeRegs<TILES> &R, const LaneBox &lb) {
    uint64_t c = 0;
#pragma unroll
    for (int t = 0; t < TILES; t++) c += tile_count_regs(R.r[t], lb);
    return c;
}

template <int TILES>
__global__ __launch_bounds__(64) void k_bounds_count_w1_pipe(const v4i *__restrict__ base, uint64_t n, DevPred pred,
                                                             uint64_t *__restrict__ partials) {
    const int lane = threadIdx.x;
    const uint64_t tiles = n / TILE_POINTS, steps = tiles / TILES, stride = gridDim.x;
    const LaneBox lb = rotate_box(pred.lo, pred.width, lane);
    uint64_t total = 0;
    if (blockIdx.x < steps) {
        PipeRegs<TILES> A, B;
        uint64_t g = blockIdx.x;
        pipe_load<TILES>(A, base, g, lane);
        for (;;) {
            const uint64_t g1 = g + stride;
            pipe_load<TILES>(B, base, g1 < steps ? g1 : g, lane);  // clamped at the tail: a re-read that hits L2
            pipe_wait<TILES, 3 * TILES>(A);                       // A has landed, B's loads stay in flight
            total += pipe_eval<TILES>(A, lb);
            if (g1 >= steps) break;
            const uint64_t g2 = g1 + stride;
            pipe_load<TILES>(A, base, g2 < steps ? g2 : g1, lane);
            pipe_wait<TILES, 3 * TILES>(B);
            total += pipe_eval<TILES>(B, lb);
            if (g2 >= steps) break;
            g = g2;
        }
        pipe_wait<TILES, 0>(A);  // the clamped tail prefetch is still in flight: land it before the registers die
        pipe_wait<TILES, 0>(B);
    }
    if (blockIdx.x == 0) {
        for (uint64_t t = steps * TILES; t < tiles; t++) total += tile_count_masks(base + t * 192, lane, lb);
        for (int k = 0; k < 4; k++) {
            const uint64_t p = tiles * TILE_POINTS + (uint64_t)(64 * k + lane);
            bool pass = false;
            if (p < n) {
                const int *q = reinterpret_cast<const int *>(base) + 3 * p;
                pass = ((uint32_t)(q[0] - pred.lo[0]) <= pred.width[0]) & ((uint32_t)(q[1] - pred.lo[1]) <= pred.width[1]) &
                       ((uint32_t)(q[2] - pred.lo[2]) <= pred.width[2]);
            }
            total += (uint64_t)__popcll(__ballot(pass));
        }
    }
    if (lane == 0) partials[blockIdx.x] = total;
}

// Batched K1 with one wave per workgroup and TILES adjacent tiles per step (the shape variants 8..11
// measure on one file).  Here `tile_begin` of the segment table counts steps (TILES * 256 points), and the
// fewer-than-a-step leftover of segment i is handled point by point by block i % gridDim.x.
template <int TILES>
__global__ __launch_bounds__(64) void k_bounds_count_batch_w1(const DevSegment *__restrict__ segs, int nseg,
                                                             uint64_t total_steps, uint64_t *__restrict__ partials) {
    constexpr uint64_t STEP_POINTS = (uint64_t)TILES * TILE_POINTS;
    const int lane = threadIdx.x;
    uint64_t total = 0;
    int s = 0;
    uint64_t seg_begin = 0, seg_end = 0;
    const v4i *seg_base = nullptr;
    LaneBox lb = {};
    bool seg_empty = true;
    for (uint64_t u = blockIdx.x; u < total_steps; u += gridDim.x) {
        if (u >= seg_end) {
            while (s + 1 < nseg && u >= segs[s + 1].tile_begin) s++;
            seg_begin = segs[s].tile_begin;
            seg_end = seg_begin + segs[s].n / STEP_POINTS;
            seg_base = reinterpret_cast<const v4i *>(segs[s].xyz);
            seg_empty = segs[s].empty != 0;
            lb = rotate_box(segs[s].lo, segs[s].width, lane);
        }
        if (seg_empty) continue;
        const v4i *tile = seg_base + (u - seg_begin) * (192 * TILES);
        v4i v[TILES][3];
#pragma unroll
        for (int t = 0; t < TILES; t++)
#pragma unroll
            for (int k = 0; k < 3; k++) v[t][k] = ld_nt(tile + 192 * t + 64 * k + lane);
#pragma unroll
        for (int t = 0; t < TILES; t++) total += tile_count_regs(v[t], lb);
    }
    for (int i = blockIdx.x; i < nseg; i += gridDim.x) {
        if (segs[i].empty) continue;
        const uint64_t n = segs[i].n;
        const int *q0 = reinterpret_cast<const int *>(segs[i].xyz);
        for (uint64_t p = (n / STEP_POINTS) * STEP_POINTS + lane; p < ((n + 63) & ~63ull); p += 64) {
            bool pass = false;
            if (p < n) {
                const int *q = q0 + 3 * p;
                pass = ((uint32_t)(q[0] - segs[i].lo[0]) <= segs[i].width[0]) &
                       ((uint32_t)(q[1] - segs[i].lo[1]) <= segs[i].width[1]) &
                       ((uint32_t)(q[2] - segs[i].lo[2]) <= segs[i].width[2]);
            }
            total += (uint64_t)__popcll(__ballot(pass));
        }
    }
    if (lane == 0) partials[blockIdx.x] = total;
}

// Batched K1, one wave per workgroup, TILES tiles per step, software-pipelined like variants 12..14: while the
// tiles of step u are evaluated the loads of step u + stride are in flight.  Steps are numbered across all
// segments (tile_begin counts steps); each of the two register sets remembers the segment its step came from.
struct SegCursor {
    int s;
    uint64_t begin, end;
    const v4i *base;
    LaneBox lb;
    bool empty;
};
template <int TILES>
__device__ __forceinline__ void seg_seek(SegCursor &c, const DevSegment *__restrict__ segs, int nseg, uint64_t u, int lane) {
    if (u < c.end) return;
    while (c.s + 1 < nseg && u >= segs[c.s + 1].tile_begin) c.s++;
    c.begin = segs[c.s].tile_begin;
    c.end = c.begin + segs[c.s].n / ((uint64_t)TILES * TILE_POINTS);
    c.base = reinterpret_cast<const v4i *>(segs[c.s].xyz);
    c.empty = segs[c.s].empty != 0;
    // the box through SGPRs: left to itself the compiler turns "select of table entries" into a per-lane address and
    // a VECTOR load, and the s_waitcnt vmcnt(0) behind that load would drain the prefetched tiles
    int32_t lo[3];
    uint32_t w[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        lo[k] = segs[c.s].lo[k];
        w[k] = segs[c.s].width[k];
        asm volatile("" : "+s"(lo[k]), "+s"(w[k]));
    }
    c.lb = rotate_box(lo, w, lane);
}

template <int TILES>
__global__ __launch_bounds__(64) void k_bounds_count_batch_pipe(const DevSegment *__restrict__ segs, int nseg,
                                                               uint64_t total_steps, uint64_t *__restrict__ partials) {
    constexpr uint64_t STEP_POINTS = (uint64_t)TILES * TILE_POINTS;
    const int lane = threadIdx.x;
    const uint64_t stride = gridDim.x;
    uint64_t total = 0;
    if (blockIdx.x < total_steps) {
        PipeRegs<TILES> A, B;
        SegCursor ca = {0, 0, 0, nullptr, {}, true}, cb;
        uint64_t u = blockIdx.x;
        seg_seek<TILES>(ca, segs, nseg, u, lane);
        pipe_load<TILES>(A, ca.base, u - ca.begin, lane);
        for (;;) {
            const uint64_t u1 = u + stride;
            cb = ca;
            if (u1 < total_steps) seg_seek<TILES>(cb, segs, nseg, u1, lane);
            pipe_load<TILES>(B, cb.base, (u1 < total_steps ? u1 : u) - cb.begin, lane);  // clamped at the tail: an L2 hit
            pipe_wait<TILES, 3 * TILES>(A);
            if (!ca.empty) total += pipe_eval<TILES>(A, ca.lb);
            if (u1 >= total_steps) break;
            const uint64_t u2 = u1 + stride;
            ca = cb;
            if (u2 < total_steps) seg_seek<TILES>(ca, segs, nseg, u2, lane);
            pipe_load<TILES>(A, ca.base, (u2 < total_steps ? u2 : u1) - ca.begin, lane);
            pipe_wait<TILES, 3 * TILES>(B);
            if (!cb.empty) total += pipe_eval<TILES>(B, cb.lb);
            if (u2 >= total_steps) break;
            u = u2;
        }
        pipe_wait<TILES, 0>(A);
        pipe_wait<TILES, 0>(B);
    }
    for (int i = blockIdx.x; i < nseg; i += gridDim.x) {  // fewer-than-a-step leftovers of segment i
        if (segs[i].empty) continue;
        const uint64_t n = segs[i].n;
        const int *q0 = reinterpret_cast<const int *>(segs[i].xyz);
        for (uint64_t p = (n / STEP_POINTS) * STEP_POINTS + lane; p < ((n + 63) & ~63ull); p += 64) {
            bool pass = false;
            if (p < n) {
                const int *q = q0 + 3 * p;
                pass = ((uint32_t)(q[0] - segs[i].lo[0]) <= segs[i].width[0]) &
                       ((uint32_t)(q[1] - segs[i].lo[1]) <= segs[i].width[1]) &
                       ((uint32_t)(q[2] - segs[i].lo[2]) <= segs[i].width[2]);
            }
            total += (uint64_t)__popcll(__ballot(pass));
        }
    }
    if (lane == 0) partials[blockIdx.x] = total;
}

// Bytes of a dword equal to zero -> 0x80 in that byte (exact, no borrow artefacts).
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) {
    const uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    return ~(t | x | 0x7f7f7f7fu);
}

// K2.  `body` is the 16-byte aligned part of the classification block; head/tail bytes are
// handled by the first block.
__global__ __launch_bounds__(BLOCK) void k_class_count_u8(const uint8_t *__restrict__ cls, uint64_t n, uint32_t pat,
                                                          uint64_t head, uint64_t nvec, uint64_t *__restrict__ partials) {
    const v4i *body = reinterpret_cast<const v4i *>(cls + head);
    const uint64_t tid = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * BLOCK;
    uint32_t cnt = 0;
    uint64_t i = tid;
    for (; i + 3 * nthreads < nvec; i += 4 * nthreads) {
        const v4i a = ld_nt(body + i), b = ld_nt(body + i + nthreads), c = ld_nt(body + i + 2 * nthreads),
                  d = ld_nt(body + i + 3 * nthreads);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cnt += __popc(zero_bytes((uint32_t)a[j] ^ pat));
            cnt += __popc(zero_bytes((uint32_t)b[j] ^ pat));
            cnt += __popc(zero_bytes((uint32_t)c[j] ^ pat));
            cnt += __popc(zero_bytes((uint32_t)d[j] ^ pat));
        }
    }
    for (; i < nvec; i += nthreads) {
        const v4i a = ld_nt(body + i);
#pragma unroll
        for (int j = 0; j < 4; j++) cnt += __popc(zero_bytes((uint32_t)a[j] ^ pat));
    }
    if (blockIdx.x == 0) {
        const uint8_t c8 = (uint8_t)(pat & 0xff);
        // head: [0, head)   tail: [head + 16*nvec, n)   (each < 16 bytes)
        if (threadIdx.x < 16) {
            const uint64_t p = threadIdx.x;
            if (p < head && cls[p] == c8) cnt++;
        } else if (threadIdx.x < 32) {
            const uint64_t p = head + 16 * nvec + (threadIdx.x - 16);
            if (p < n && cls[p] == c8) cnt++;
        }
    }
    // wave reduce
    uint64_t w = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down((unsigned long long)w, off, 64);
    block_store_partial(w, partials);
}

// Batched K2: the classification blocks of many files in one launch.  A wave-tile is 256 aligned
// 16-byte vectors (4 KiB, four loads in flight per lane); leftovers (< 256 vectors, head and tail
// bytes) of segment i are handled by block i % gridDim.x.
// The class segments live in the same device table as the bounds segments, at DevSegment pitch.
__device__ __forceinline__ const DevClassSegment &cseg(const DevSegment *raw, int i) {
    return *reinterpret_cast<const DevClassSegment *>(raw + i);
}

__global__ __launch_bounds__(BLOCK) void k_class_count_batch(const DevSegment *__restrict__ raw, int nseg,
                                                             uint64_t total_tiles, uint64_t *__restrict__ partials) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave_id = (uint64_t)blockIdx.x * WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t stride = (uint64_t)gridDim.x * WAVES;
    uint32_t cnt = 0;
    int s = 0;
    uint64_t seg_begin = 0, seg_end = 0;
    const v4i *body = nullptr;
    uint32_t pat = 0;
    for (uint64_t t = wave_id; t < total_tiles; t += stride) {
        if (t >= seg_end) {
            while (s + 1 < nseg && t >= cseg(raw, s + 1).tile_begin) s++;
            seg_begin = cseg(raw, s).tile_begin;
            seg_end = seg_begin + cseg(raw, s).nvec / 256;
            body = reinterpret_cast<const v4i *>(cseg(raw, s).cls + cseg(raw, s).head);
            pat = cseg(raw, s).pat;
        }
        const v4i *tile = body + (t - seg_begin) * 256;
        const v4i a = ld_nt(tile + lane), b = ld_nt(tile + 64 + lane), c = ld_nt(tile + 128 + lane), d = ld_nt(tile + 192 + lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cnt += __popc(zero_bytes((uint32_t)a[j] ^ pat));
            cnt += __popc(zero_bytes((uint32_t)b[j] ^ pat));
            cnt += __popc(zero_bytes((uint32_t)c[j] ^ pat));
            cnt += __popc(zero_bytes((uint32_t)d[j] ^ pat));
        }
    }
    for (int i = blockIdx.x; i < nseg; i += gridDim.x) {
        const DevClassSegment g = cseg(raw, i);
        const uint8_t c8 = (uint8_t)(g.pat & 0xff);
        const v4i *bd = reinterpret_cast<const v4i *>(g.cls + g.head);
        for (uint64_t v = (g.nvec / 256) * 256 + threadIdx.x; v < g.nvec; v += BLOCK) {  // < 256 leftover vectors
            const v4i a = bd[v];
#pragma unroll
            for (int j = 0; j < 4; j++) cnt += __popc(zero_bytes((uint32_t)a[j] ^ g.pat));
        }
        if (threadIdx.x < 16) {
            const uint64_t p = threadIdx.x;
            if (p < g.head && g.cls[p] == c8) cnt++;
        } else if (threadIdx.x < 32) {
            const uint64_t p = g.head + 16 * g.nvec + (threadIdx.x - 16);
            if (p < g.n && g.cls[p] == c8) cnt++;
        }
    }
    uint64_t w = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down((unsigned long long)w, off, 64);
    block_store_partial(w, partials);
}

// Batched K2 with one wave per workgroup and LOADS 1 KiB loads per step (tile_begin counts steps of
// LOADS * 64 vectors); leftovers of segment i by block i % gridDim.x.
template <int LOADS>
__global__ __launch_bounds__(64) void k_class_count_batch_w1(const DevSegment *__restrict__ raw, int nseg, uint64_t total_steps,
                                                            uint64_t *__restrict__ partials) {
    constexpr uint64_t STEP_VEC = 64 * LOADS;
    const int lane = threadIdx.x;
    uint32_t cnt = 0;
    int s = 0;
    uint64_t seg_begin = 0, seg_end = 0;
    const v4i *body = nullptr;
    uint32_t pat = 0;
    for (uint64_t t = blockIdx.x; t < total_steps; t += gridDim.x) {
        if (t >= seg_end) {
            while (s + 1 < nseg && t >= cseg(raw, s + 1).tile_begin) s++;
            seg_begin = cseg(raw, s).tile_begin;
            seg_end = seg_begin + cseg(raw, s).nvec / STEP_VEC;
            body = reinterpret_cast<const v4i *>(cseg(raw, s).cls + cseg(raw, s).head);
            pat = cseg(raw, s).pat;
        }
        const v4i *tile = body + (t - seg_begin) * STEP_VEC;
        v4i v[LOADS];
#pragma unroll
        for (int k = 0; k < LOADS; k++) v[k] = ld_nt(tile + 64 * k + lane);
#pragma unroll
        for (int k = 0; k < LOADS; k++)
#pragma unroll
            for (int j = 0; j < 4; j++) cnt += __popc(zero_bytes((uint32_t)v[k][j] ^ pat));
    }
    for (int i = blockIdx.x; i < nseg; i += gridDim.x) {
        const DevClassSegment g = cseg(raw, i);
        const uint8_t c8 = (uint8_t)(g.pat & 0xff);
        const v4i *bd = reinterpret_cast<const v4i *>(g.cls + g.head);
        for (uint64_t v = (g.nvec / STEP_VEC) * STEP_VEC + lane; v < g.nvec; v += 64) {
            const v4i a = bd[v];
#pragma unroll
            for (int j = 0; j < 4; j++) cnt += __popc(zero_bytes((uint32_t)a[j] ^ g.pat));
        }
        if (lane < 16) {
            const uint64_t p = lane;
            if (p < g.head && g.cls[p] == c8) cnt++;
        } else if (lane < 32) {
            const uint64_t p = g.head + 16 * g.nvec + (lane - 16);
            if (p < g.n && g.cls[p] == c8) cnt++;
        }
    }
    uint64_t w = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down((unsigned long long)w, off, 64);
    if (lane == 0) partials[blockIdx.x] = w;
}

// Batched K2, one wave per workgroup, LOADS 1 KiB loads per step, software-pipelined like k_bounds_count_batch_pipe.
template <int LOADS>
struct ClassRegs {
    v4i r[LOADS];
};
template <int LOADS>
__device__ __forceinline__ void class_load(ClassRegs<LOADS> &R, const v4i *tile, int lane) {
#pragma unroll
    for (int k = 0; k < LOADS; k++) {
        const v4i *q = tile + 64 * k + lane;
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(R.r[k]) : "v"(q) : "memory");
    }
}
template <int LOADS, int PENDING>
__device__ __forceinline__ void class_wait(ClassRegs<LOADS> &R) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory");
#pragma unroll
    for (int k = 0; k < LOADS; k++) asm volatile("" : "+v"(R.r[k])::"memory");
}
template <int LOADS>
__device__ __forceinline__ uint32_t class_eval(const ClassRegs<LOADS> &R, uint32_t pat) {
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < LOADS; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) c += __popc(zero_bytes((uint32_t)R.r[k][j] ^ pat));
    return c;
}
struct ClassCursor {
    int s;
    uint64_t begin, end;
    const v4i *body;
    uint32_t pat;
};
template <int LOADS>
__device__ __forceinline__ void class_seek(ClassCursor &c, const DevSegment *__restrict__ raw, int nseg, uint64_t u) {
    if (u < c.end) return;
    while (c.s + 1 < nseg && u >= cseg(raw, c.s + 1).tile_begin) c.s++;
    c.begin = cseg(raw, c.s).tile_begin;
    c.end = c.begin + cseg(raw, c.s).nvec / (64ull * LOADS);
    c.body = reinterpret_cast<const v4i *>(cseg(raw, c.s).cls + cseg(raw, c.s).head);
    c.pat = cseg(raw, c.s).pat;
}

template <int LOADS>
__global__ __launch_bounds__(64) void k_class_count_batch_pipe(const DevSegment *__restrict__ raw, int nseg, uint64_t total_steps,
                                                              uint64_t *__restrict__ partials) {
    constexpr uint64_t STEP_VEC = 64 * LOADS;
    const int lane = threadIdx.x;
    const uint64_t stride = gridDim.x;
    uint32_t cnt = 0;
    if (blockIdx.x < total_steps) {
        ClassRegs<LOADS> A, B;
        ClassCursor ca = {0, 0, 0, nullptr, 0}, cb;
        uint64_t u = blockIdx.x;
        class_seek<LOADS>(ca, raw, nseg, u);
        class_load<LOADS>(A, ca.body + (u - ca.begin) * STEP_VEC, lane);
        for (;;) {
            const uint64_t u1 = u + stride;
            cb = ca;
            if (u1 < total_steps) class_seek<LOADS>(cb, raw, nseg, u1);
            class_load<LOADS>(B, cb.body + ((u1 < total_steps ? u1 : u) - cb.begin) * STEP_VEC, lane);
            class_wait<LOADS, LOADS>(A);
            cnt += class_eval<LOADS>(A, ca.pat);
            if (u1 >= total_steps) break;
            const uint64_t u2 = u1 + stride;
            ca = cb;
            if (u2 < total_steps) class_seek<LOADS>(ca, raw, nseg, u2);
            class_load<LOADS>(A, ca.body + ((u2 < total_steps ? u2 : u1) - ca.begin) * STEP_VEC, lane);
            class_wait<LOADS, LOADS>(B);
            cnt += class_eval<LOADS>(B, cb.pat);
            if (u2 >= total_steps) break;
            u = u2;
        }
        class_wait<LOADS, 0>(A);
        class_wait<LOADS, 0>(B);
    }
    for (int i = blockIdx.x; i < nseg; i += gridDim.x) {
        const DevClassSegment g = cseg(raw, i);
        const uint8_t c8 = (uint8_t)(g.pat & 0xff);
        const v4i *bd = reinterpret_cast<const v4i *>(g.cls + g.head);
        for (uint64_t v = (g.nvec / STEP_VEC) * STEP_VEC + lane; v < g.nvec; v += 64) {
            const v4i a = bd[v];
#pragma unroll
            for (int j = 0; j < 4; j++) cnt += __popc(zero_bytes((uint32_t)a[j] ^ g.pat));
        }
        if (lane < 16) {
            const uint64_t p = lane;
            if (p < g.head && g.cls[p] == c8) cnt++;
        } else if (lane < 32) {
            const uint64_t p = g.head + 16 * g.nvec + (lane - 16);
            if (p < g.n && g.cls[p] == c8) cnt++;
        }
    }
    uint64_t w = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down((unsigned long long)w, off, 64);
    if (lane == 0) partials[blockIdx.x] = w;
}

// Per-file K2 in the same shape: one wave per workgroup, LOADS KiB per step, software-pipelined.
template <int LOADS>
__global__ __launch_bounds__(64) void k_class_count_pipe(const uint8_t *__restrict__ cls, uint64_t n, uint32_t pat, uint64_t head,
                                                        uint64_t nvec, uint64_t *__restrict__ partials) {
    constexpr uint64_t STEP_VEC = 64 * LOADS;
    const int lane = threadIdx.x;
    const v4i *body = reinterpret_cast<const v4i *>(cls + head);
    const uint64_t steps = nvec / STEP_VEC, stride = gridDim.x;
    uint32_t cnt = 0;
    if (blockIdx.x < steps) {
        ClassRegs<LOADS> A, B;
        uint64_t u = blockIdx.x;
        class_load<LOADS>(A, body + u * STEP_VEC, lane);
        for (;;) {
            const uint64_t u1 = u + stride;
            class_load<LOADS>(B, body + (u1 < steps ? u1 : u) * STEP_VEC, lane);
            class_wait<LOADS, LOADS>(A);
            cnt += class_eval<LOADS>(A, pat);
            if (u1 >= steps) break;
            const uint64_t u2 = u1 + stride;
            class_load<LOADS>(A, body + (u2 < steps ? u2 : u1) * STEP_VEC, lane);
            class_wait<LOADS, LOADS>(B);
            cnt += class_eval<LOADS>(B, pat);
            if (u2 >= steps) break;
            u = u2;
        }
        class_wait<LOADS, 0>(A);
        class_wait<LOADS, 0>(B);
    }
    if (blockIdx.x == 0) {
        const uint8_t c8 = (uint8_t)(pat & 0xff);
        for (uint64_t v = steps * STEP_VEC + lane; v < nvec; v += 64) {  // fewer than a step of leftover vectors
            const v4i a = body[v];
#pragma unroll
            for (int j = 0; j < 4; j++) cnt += __popc(zero_bytes((uint32_t)a[j] ^ pat));
        }
        if (lane < 16) {  // head: [0, head)   tail: [head + 16*nvec, n)   (each < 16 bytes)
            const uint64_t p = lane;
            if (p < head && cls[p] == c8) cnt++;
        } else if (lane < 32) {
            const uint64_t p = head + 16 * nvec + (lane - 16);
            if (p < n && cls[p] == c8) cnt++;
        }
    }
    uint64_t w = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down((unsigned long long)w, off, 64);
    if (lane == 0) partials[blockIdx.x] = w;
}

__global__ __launch_bounds__(BLOCK) void k_finish_count(const uint64_t *__restrict__ partials, int nblocks,
                                                        uint64_t *__restrict__ d_count) {
    __shared__ uint64_t s[BLOCK];
    uint64_t t = 0;
    for (int i = threadIdx.x; i < nblocks; i += BLOCK) t += partials[i];
    s[threadIdx.x] = t;
    __syncthreads();
    for (int off = BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd((unsigned long long *)d_count, (unsigned long long)s[0]);
}

}  // namespace

static int grid_for(pcq_ctx *ctx, uint64_t work_items_per_block_min, uint64_t items, int blocks_per_cu = 0) {
    uint64_t want = (items + work_items_per_block_min - 1) / work_items_per_block_min;
    uint64_t cap = (uint64_t)ctx->num_cus * (uint64_t)(blocks_per_cu ? blocks_per_cu : ctx->grid_blocks_per_cu);
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

int pcq_launch_bounds_count_xyz12(pcq_ctx *ctx, const void *d_xyz, uint64_t n, const DevPred &pred,
                                  uint64_t *d_count, hipStream_t s) {
    if (n == 0 || pred.empty) return PCQ_OK;
    if (((uintptr_t)d_xyz & 15) != 0) return pcq_fail(PCQ_ERR_ARG, "bounds_count_xyz12: positions block must be 16-byte aligned");
    const int grid = grid_for(ctx, (uint64_t)WAVES * TILE_POINTS, n);
    int rc = pcq_ensure_partials(ctx, (size_t)grid);
    if (rc) return rc;
    const v4i *base = reinterpret_cast<const v4i *>(d_xyz);
    if (ctx->k1_variant >= 12 && ctx->k1_variant <= 14) {  // one wave per workgroup, software-pipelined; 12: 2 tiles per step, 13: 1, 14: 3
        const int tps = ctx->k1_variant == 12 ? 2 : (ctx->k1_variant == 13 ? 1 : 3);
        const uint64_t units = n / ((uint64_t)tps * TILE_POINTS) + 1;
        uint64_t g = ctx->k1_grid > 0 ? (uint64_t)ctx->k1_grid : (uint64_t)ctx->num_cus * ctx->k1_waves_per_cu;
        if (g > units) g = units;
        rc = pcq_ensure_partials(ctx, (size_t)g);
        if (rc) return rc;
        if (tps == 1) hipLaunchKernelGGL(k_bounds_count_w1_pipe<1>, dim3((unsigned)g), dim3(64), 0, s, base, n, pred, ctx->d_partials);
        else if (tps == 2) hipLaunchKernelGGL(k_bounds_count_w1_pipe<2>, dim3((unsigned)g), dim3(64), 0, s, base, n, pred, ctx->d_partials);
        else hipLaunchKernelGGL(k_bounds_count_w1_pipe<3>, dim3((unsigned)g), dim3(64), 0, s, base, n, pred, ctx->d_partials);
        hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, (int)g, d_count);
        PCQ_HIP(hipGetLastError());
        return PCQ_OK;
    }
    if (ctx->k1_variant >= 8 && ctx->k1_variant <= 11) {  // one wave per workgroup, TILES tiles per step
        const int tiles_per_step = ctx->k1_variant - 7;
        const uint64_t units = n / ((uint64_t)tiles_per_step * TILE_POINTS) + 1;
        uint64_t g = (uint64_t)ctx->num_cus * ctx->k1_waves_per_cu;
        if (g > units) g = units;
        rc = pcq_ensure_partials(ctx, (size_t)g);
        if (rc) return rc;
        switch (tiles_per_step) {
        case 1: hipLaunchKernelGGL(k_bounds_count_w1<1>, dim3((unsigned)g), dim3(64), 0, s, base, n, pred, ctx->d_partials); break;
        case 2: hipLaunchKernelGGL(k_bounds_count_w1<2>, dim3((unsigned)g), dim3(64), 0, s, base, n, pred, ctx->d_partials); break;
        case 3: hipLaunchKernelGGL(k_bounds_count_w1<3>, dim3((unsigned)g), dim3(64), 0, s, base, n, pred, ctx->d_partials); break;
        default: hipLaunchKernelGGL(k_bounds_count_w1<4>, dim3((unsigned)g), dim3(64), 0, s, base, n, pred, ctx->d_partials); break;
        }
        hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, (int)g, d_count);
        PCQ_HIP(hipGetLastError());
        return PCQ_OK;
    }
    switch (ctx->k1_variant) {
    case 1: hipLaunchKernelGGL(k_bounds_count_xyz12<1>, dim3(grid), dim3(BLOCK), 0, s, base, n, pred, ctx->d_partials); break;
    case 2: hipLaunchKernelGGL(k_bounds_count_xyz12<2>, dim3(grid), dim3(BLOCK), 0, s, base, n, pred, ctx->d_partials); break;
    case 3: hipLaunchKernelGGL(k_bounds_count_xyz12<3>, dim3(grid), dim3(BLOCK), 0, s, base, n, pred, ctx->d_partials); break;
    case 4: hipLaunchKernelGGL(k_bounds_count_xyz12<4>, dim3(grid), dim3(BLOCK), 0, s, base, n, pred, ctx->d_partials); break;
    case 5: hipLaunchKernelGGL(k_bounds_count_xyz12<5>, dim3(grid), dim3(BLOCK), 0, s, base, n, pred, ctx->d_partials); break;
    case 6: hipLaunchKernelGGL(k_bounds_count_xyz12<6>, dim3(grid), dim3(BLOCK), 0, s, base, n, pred, ctx->d_partials); break;
    case 7: hipLaunchKernelGGL(k_bounds_count_xyz12<7>, dim3(grid), dim3(BLOCK), 0, s, base, n, pred, ctx->d_partials); break;
    default: hipLaunchKernelGGL(k_bounds_count_xyz12<0>, dim3(grid), dim3(BLOCK), 0, s, base, n, pred, ctx->d_partials); break;
    }
    hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, grid, d_count);
    PCQ_HIP(hipGetLastError());
    return PCQ_OK;
}

int pcq_launch_class_count_u8(pcq_ctx *ctx, const void *d_cls, uint64_t n, uint8_t cls, uint64_t *d_count,
                              hipStream_t s) {
    if (n == 0) return PCQ_OK;
    uint64_t head = (uint64_t)((16 - ((uintptr_t)d_cls & 15)) & 15);
    if (head > n) head = n;
    const uint64_t nvec = (n - head) / 16;
    const uint32_t pat = 0x01010101u * (uint32_t)cls;
    if (ctx->class_batch_pipe) {  // the same one-wave pipelined shape as the batched K2
        uint64_t g = (uint64_t)ctx->num_cus * ctx->class_batch_waves_per_cu;
        const uint64_t steps = nvec / 256 + 1;
        if (g > steps) g = steps;
        int prc = pcq_ensure_partials(ctx, (size_t)g);
        if (prc) return prc;
        hipLaunchKernelGGL(k_class_count_pipe<4>, dim3((unsigned)g), dim3(64), 0, s, reinterpret_cast<const uint8_t *>(d_cls), n, pat, head,
                           nvec, ctx->d_partials);
        hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, (int)g, d_count);
        PCQ_HIP(hipGetLastError());
        return PCQ_OK;
    }
    const int grid = grid_for(ctx, (uint64_t)BLOCK * 4, nvec ? nvec : 1);
    int rc = pcq_ensure_partials(ctx, (size_t)grid);
    if (rc) return rc;
    hipLaunchKernelGGL(k_class_count_u8, dim3(grid), dim3(BLOCK), 0, s, reinterpret_cast<const uint8_t *>(d_cls), n, pat,
                       head, nvec, ctx->d_partials);
    hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, grid, d_count);
    PCQ_HIP(hipGetLastError());
    return PCQ_OK;
}

// batch_variant: 0 = 256-thread blocks (1 tile per wave step) · 1, 2 = one-wave workgroups with 2, 3 tiles per step ·
// 3 = one-wave workgroups, 2 tiles per step, software-pipelined
static int batch_tiles_per_step(int v) { return v == 0 ? 1 : (v == 2 ? 3 : 2); }

extern "C" int pcq_scan_dev_count_batch(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *preds,
                                        size_t nsegments, uint64_t *device_total, void *stream) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx || (!cols && nsegments) || (!preds && nsegments) || !device_total)
        return pcq_fail(PCQ_ERR_ARG, "pcq_scan_dev_count_batch: null argument");
    if (nsegments == 0) return PCQ_OK;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    {
        const int src = pcq_scratch_stream(ctx, s);
        if (src) return src;
    }
    if (nsegments > ctx->segments_cap) {
        if (ctx->d_segments) (void)hipFree(ctx->d_segments);
        if (ctx->h_segments) (void)hipHostFree(ctx->h_segments);
        ctx->d_segments = nullptr;
        ctx->h_segments = nullptr;
        ctx->segments_cap = 0;
        ctx->segments_uploaded = 0;
        size_t cap = nsegments < 64 ? 64 : nsegments;
        PCQ_HIP(hipMalloc((void **)&ctx->d_segments, cap * sizeof(DevSegment)));
        PCQ_HIP(hipHostMalloc((void **)&ctx->h_segments, cap * sizeof(DevSegment), hipHostMallocDefault));
        ctx->segments_cap = cap;
    }
    static_assert(sizeof(DevClassSegment) <= sizeof(DevSegment), "the two segment tables share one buffer");
    const int kind = preds[0].kind;
    // Build the segment table; it is uploaded only when it differs from the one already in HBM
    // (a repeated query re-launches without touching the pinned buffer, so no host-side wait).
    std::vector<DevSegment> table(nsegments);
    memset(table.data(), 0, nsegments * sizeof(DevSegment));
    uint64_t tiles = 0, points = 0;
    for (size_t i = 0; i < nsegments; i++) {
        if (preds[i].kind != kind) return pcq_fail(PCQ_ERR_ARG, "count_batch: mixed predicate kinds");
        if (kind == PCQ_PRED_CLASS) {
            if (cols[i].cls_stride != 1 || (!cols[i].cls && cols[i].n))
                return pcq_fail(PCQ_ERR_ARG, "count_batch: LAST classification blocks only (stride 1)");
            DevClassSegment g;
            memset(&g, 0, sizeof g);
            g.cls = (const uint8_t *)cols[i].cls;
            g.n = cols[i].n;
            g.head = (uint64_t)((16 - ((uintptr_t)g.cls & 15)) & 15);
            if (g.head > g.n) g.head = g.n;
            g.nvec = (g.n - g.head) / 16;
            g.tile_begin = tiles;
            g.pat = 0x01010101u * (uint32_t)preds[i].cls;
            memcpy(&table[i], &g, sizeof g);
            tiles += g.nvec / (ctx->class_batch_loads ? 64 * (uint64_t)ctx->class_batch_loads : 256);
            points += g.n;
            continue;
        }
        if (kind != PCQ_PRED_BOUNDS) return pcq_fail(PCQ_ERR_ARG, "count_batch: bad predicate kind %d", kind);
        if (cols[i].xyz_stride != 12) return pcq_fail(PCQ_ERR_ARG, "count_batch: LAST positions blocks only (stride 12)");
        if (((uintptr_t)cols[i].xyz & 15) != 0) return pcq_fail(PCQ_ERR_ARG, "count_batch: positions block %zu not 16-byte aligned", i);
        DevPred dp;
        int rc = pcq_make_dev_pred(&preds[i], &dp);
        if (rc) return rc;
        DevSegment &g = table[i];
        g.xyz = reinterpret_cast<const int4 *>(cols[i].xyz);
        g.n = cols[i].n;
        g.tile_begin = tiles;
        for (int a = 0; a < 3; a++) g.lo[a] = dp.lo[a], g.width[a] = dp.width[a];
        g.empty = dp.empty;
        tiles += cols[i].n / ((uint64_t)batch_tiles_per_step(ctx->batch_variant) * TILE_POINTS);
        points += cols[i].n;
    }
    if (ctx->segments_uploaded != nsegments || ctx->segments_kind != kind ||
        memcmp(ctx->h_segments, table.data(), nsegments * sizeof(DevSegment)) != 0) {
        PCQ_HIP(hipStreamSynchronize(s));  // the previous upload from the pinned table must have been consumed
        memcpy(ctx->h_segments, table.data(), nsegments * sizeof(DevSegment));
        PCQ_HIP(hipMemcpyAsync(ctx->d_segments, ctx->h_segments, nsegments * sizeof(DevSegment), hipMemcpyHostToDevice, s));
        ctx->segments_uploaded = nsegments;
        ctx->segments_kind = kind;
    }
    if (kind == PCQ_PRED_CLASS && ctx->class_batch_loads) {  // one wave per workgroup, class_batch_loads KiB per step
        uint64_t g = (uint64_t)ctx->num_cus * ctx->class_batch_waves_per_cu;
        if (g > tiles + nsegments) g = tiles + nsegments;
        int crc = pcq_ensure_partials(ctx, (size_t)g);
        if (crc) return crc;
        if (ctx->class_batch_pipe) {
            switch (ctx->class_batch_loads) {
            case 4: hipLaunchKernelGGL(k_class_count_batch_pipe<4>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles, ctx->d_partials); break;
            case 6: hipLaunchKernelGGL(k_class_count_batch_pipe<6>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles, ctx->d_partials); break;
            case 8: hipLaunchKernelGGL(k_class_count_batch_pipe<8>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles, ctx->d_partials); break;
            default: hipLaunchKernelGGL(k_class_count_batch_pipe<12>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles, ctx->d_partials); break;
            }
            hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, (int)g, device_total);
            PCQ_HIP(hipGetLastError());
            return PCQ_OK;
        }
        switch (ctx->class_batch_loads) {
        case 4: hipLaunchKernelGGL(k_class_count_batch_w1<4>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles, ctx->d_partials); break;
        case 6: hipLaunchKernelGGL(k_class_count_batch_w1<6>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles, ctx->d_partials); break;
        case 8: hipLaunchKernelGGL(k_class_count_batch_w1<8>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles, ctx->d_partials); break;
        default: hipLaunchKernelGGL(k_class_count_batch_w1<12>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles, ctx->d_partials); break;
        }
        hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, (int)g, device_total);
        PCQ_HIP(hipGetLastError());
        return PCQ_OK;
    }
    if (kind == PCQ_PRED_CLASS) {
        // one 4 KiB tile per wave per iteration; the class stream wants more waves in flight than K1
        const int cgrid = grid_for(ctx, (uint64_t)WAVES * 4096, points ? points : 1, ctx->batch_blocks_per_cu + 1);
        int crc = pcq_ensure_partials(ctx, (size_t)cgrid);
        if (crc) return crc;
        hipLaunchKernelGGL(k_class_count_batch, dim3(cgrid), dim3(BLOCK), 0, s, ctx->d_segments, (int)nsegments, tiles,
                           ctx->d_partials);
        hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, cgrid, device_total);
        PCQ_HIP(hipGetLastError());
        return PCQ_OK;
    }
    if (ctx->batch_variant >= 1) {  // one wave per workgroup, 2 (variant 1) or 3 (variant 2) adjacent tiles per step
        uint64_t g = (uint64_t)ctx->num_cus * ctx->batch_waves_per_cu;
        if (g > tiles + nsegments) g = tiles + nsegments;
        int wrc = pcq_ensure_partials(ctx, (size_t)g);
        if (wrc) return wrc;
        if (ctx->batch_variant == 3)
            hipLaunchKernelGGL(k_bounds_count_batch_pipe<2>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles,
                               ctx->d_partials);
        else if (ctx->batch_variant == 1)
            hipLaunchKernelGGL(k_bounds_count_batch_w1<2>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles,
                               ctx->d_partials);
        else
            hipLaunchKernelGGL(k_bounds_count_batch_w1<3>, dim3((unsigned)g), dim3(64), 0, s, ctx->d_segments, (int)nsegments, tiles,
                               ctx->d_partials);
        hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, (int)g, device_total);
        PCQ_HIP(hipGetLastError());
        return PCQ_OK;
    }
    const int grid = grid_for(ctx, (uint64_t)WAVES * TILE_POINTS, points ? points : 1, ctx->batch_blocks_per_cu);
    int rc = pcq_ensure_partials(ctx, (size_t)grid);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bounds_count_batch, dim3(grid), dim3(BLOCK), 0, s, ctx->d_segments, (int)nsegments, tiles,
                       ctx->d_partials);
    hipLaunchKernelGGL(k_finish_count, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, grid, device_total);
    PCQ_HIP(hipGetLastError());
    return PCQ_OK;
}
