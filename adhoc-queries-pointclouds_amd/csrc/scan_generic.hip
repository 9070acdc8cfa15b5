// scan_generic.hip — layout-generic scan kernels (any column strides: LAST column blocks or LAS
// AoS records), used where the count-only fast paths of scan_count.hip do not apply:
//   * strided count            — LAS bounds/class count (las.rs:101-119, :221-231), unaligned LAST
//                                (one record per lane: 4.1-5.9 TB/s of record bytes; an LDS-tiled variant
//                                with 16-byte coalesced loads was measured and was NOT faster,
//                                profiles/r01_las_aos_count_rate.log)
//   * order-preserving emit    — BufferCollector semantics (collect_points.rs:29-31): matches are
//                                appended in file order, as 31-byte Point records built like
//                                last.rs:137-163.
// The emit is a stable stream compaction in three launches on one stream, with nothing coming back to the host:
// (1) k_tile_counts: matches per tile of 2048 points, (2) a three-step exclusive scan of the tile counts on the device,
// (3) k_emit_points: the predicate again, each match written at collector count + tile offset + rank-in-tile — the rank
// from wave64 ballot prefixes (popcount(mask & lanes_below)) and the per-(row, wave) counts in LDS.  The collector's
// point count lives on the device: the emit reads it as its base and stores the new count.
// A single-pass form (decoupled look-back over the tiles, one read of the positions) was built and measured first: on
// this chip a look-back round is a cross-XCD round trip of several microseconds and the kernel took 2.8 ms for a
// 163 M-point file against 1.05 ms for the same kernel with the look-back cut out — the second read of the positions
// (0.3 ms at the streaming rate) is much cheaper than the dependency chain (DESIGN_HISTORY.md D, profiles/r02_emit_lookback.txt).
#include "dev_common.h"

using namespace pcqdev;

namespace {

template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_generic_count(DevCols c, DevPred pr, uint64_t *__restrict__ partials) {
    const uint64_t tid = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * BLOCK;
    uint64_t cnt = 0;
    uint64_t i = tid;
    // four independent points per thread and step: a strided 12-byte (or 1-byte) load keeps only 64 x stride bytes
    // of a wave in flight, so the loads of four steps are issued together (profiles/r01_las_aos_count_rate.log)
    for (; i + 3 * nthreads < c.n; i += 4 * nthreads) {
        bool m[4];
#pragma unroll
        for (int k = 0; k < 4; k++) m[k] = eval_pred_kind<KIND>(c, pr, i + k * nthreads);
#pragma unroll
        for (int k = 0; k < 4; k++) cnt += m[k] ? 1 : 0;
    }
    for (; i < c.n; i += nthreads) cnt += eval_pred_kind<KIND>(c, pr, i) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down((unsigned long long)cnt, off, 64);
    __shared__ uint64_t s_w[WAVES];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int i = 0; i < WAVES; i++) t += s_w[i];
        partials[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(BLOCK) void k_sum_partials(const uint64_t *__restrict__ partials, int nblocks,
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

// 31-byte records at a 31-byte pitch are hostile to per-lane stores (31 byte-stores per match, 16
// cache lines touched per wave-instruction).  The block instead assembles the records of 1024 input
// points in LDS, laid out congruent (mod 16) to their final global position, and then streams the
// contiguous byte range out with 16-byte stores; only the two ragged ends use byte stores, so
// neighbouring blocks never write the same 16-byte chunk.
constexpr int EMIT_ITEMS = 8;                                   // points per thread and tile: 2048-point tiles
constexpr int EMIT_TILE = BLOCK * EMIT_ITEMS;
constexpr int FLUSH_ITEMS = 4;                                  // input rows of 256 points per LDS flush
constexpr int STAGE_BYTES = FLUSH_ITEMS * BLOCK * 31 + 16;      // 31,760 B: five blocks per CU fit in LDS
constexpr int STAGE_WORDS = (STAGE_BYTES + 3) / 4 + 9;          // + the dwords a record's last OR may touch

// A 31-byte record lands at an arbitrary byte offset of the LDS image, and neighbouring records share dwords.  Byte
// stores cost 31 LDS instructions per record (the LDS pipe, not HBM, then bounds the emit); instead the record is
// shifted into place as nine dwords and OR-ed into an image that starts out zero — nine LDS operations, no read.
__device__ __forceinline__ void or_point31(uint32_t *image, uint32_t byte_offset, const pcq_point &p) {
    const uint64_t bx = (uint64_t)__double_as_longlong(p.x), by = (uint64_t)__double_as_longlong(p.y), bz = (uint64_t)__double_as_longlong(p.z);
    const uint32_t src[8] = {(uint32_t)bx, (uint32_t)(bx >> 32), (uint32_t)by, (uint32_t)(by >> 32), (uint32_t)bz, (uint32_t)(bz >> 32),
                             (uint32_t)p.r | ((uint32_t)p.g << 16), (uint32_t)p.b | ((uint32_t)p.classification << 16)};  // byte 31 = 0
    const uint32_t d0 = byte_offset >> 2, back = 32u - 8u * (byte_offset & 3u);  // 32, 24, 16 or 8
    uint32_t prev = 0;
#pragma unroll
    for (int d = 0; d < 9; d++) {
        const uint32_t cur = d < 8 ? src[d] : 0u;
        const uint32_t v = (uint32_t)((((uint64_t)cur << 32) | prev) >> back);
        if (v) atomicOr(&image[d0 + d], v);
        prev = cur;
    }
}

// What a tile's threads need from HBM: the predicate's inputs and — for the emit — the attributes a match's record carries.
template <int KIND, bool ATTRS>
struct TileIn {
    RawPoint rps[EMIT_ITEMS];
    bool passes[EMIT_ITEMS];
    uint32_t attr_cls[EMIT_ITEMS], attr_rg[EMIT_ITEMS], attr_b[EMIT_ITEMS];
};
// Everything is requested together, before the first compare.  (Loading the attributes where the record is built — behind
// the ranks, one dependent round trip per row of 256 points — left the waves waiting 80 % of the time.)
// RGB: the file has a colour block (formats 2, 3, 5).  Compile-time, because a colourless file otherwise pays three masked
// loads per point from one address — 24 of a thread's 40 memory instructions per tile.
template <int KIND, bool ATTRS, bool RGB>
__device__ __forceinline__ void tile_load_and_test(const DevCols &c, const DevPred &pr, uint64_t base, TileIn<KIND, ATTRS> &T) {
    // No branch around a load: with `c.cls ? c.cls[i] : 0` in the unrolled loop every load sat in its own block and was
    // waited for at the block's end (eight serial round trips per tile).  A missing column is read from a valid address
    // with stride 0 and masked instead.
    const bool has_cls = (ATTRS || KIND == PCQ_PRED_CLASS) && c.cls, has_rgb = ATTRS && RGB && c.rgb;
    const uint8_t *fallback = c.xyz ? c.xyz : c.cls;  // (one of the two exists: the predicate reads it)
    const uint8_t *clsp = has_cls ? c.cls : fallback, *rgbp = has_rgb ? c.rgb : fallback;
    const uint64_t cls_stride = has_cls ? c.cls_stride : 0, rgb_stride = has_rgb ? c.rgb_stride : 0;
    const uint32_t cls_mask = has_cls ? 0xffu : 0u, rg_mask = has_rgb ? 0xffffffffu : 0u, b_mask = has_rgb ? 0xffffu : 0u;
    uint32_t raw_cls[EMIT_ITEMS], raw_rg[EMIT_ITEMS], raw_b[EMIT_ITEMS];
#pragma unroll
    for (int j = 0; j < EMIT_ITEMS; j++) {
        const uint64_t i0 = base + (uint64_t)j * BLOCK + threadIdx.x, i = i0 < c.n ? i0 : c.n - 1;
        raw_cls[j] = clsp[i * cls_stride];  // last.rs:138-142
        raw_rg[j] = raw_b[j] = 0;
        if (ATTRS && RGB) {  // last.rs:145-153
            const uint8_t *q = rgbp + i * rgb_stride;
            raw_rg[j] = (uint32_t)ld_u16(q) | ((uint32_t)ld_u16(q + 2) << 16);
            raw_b[j] = ld_u16(q + 4);
        }
        if (KIND != PCQ_PRED_CLASS) T.rps[j] = ld_xyz_stream(c, i);
    }
#pragma unroll
    for (int j = 0; j < EMIT_ITEMS; j++) T.attr_cls[j] = raw_cls[j] & cls_mask, T.attr_rg[j] = raw_rg[j] & rg_mask, T.attr_b[j] = raw_b[j] & b_mask;
#pragma unroll
    for (int j = 0; j < EMIT_ITEMS; j++) {
        const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
        bool pass;
        if (KIND == PCQ_PRED_CLASS) pass = T.attr_cls[j] == pr.cls;
        else if (KIND == PCQ_PRED_BOUNDS)
            pass = (pr.empty == 0) & ((uint32_t)(T.rps[j].x - pr.lo[0]) <= pr.width[0]) & ((uint32_t)(T.rps[j].y - pr.lo[1]) <= pr.width[1]) &
                   ((uint32_t)(T.rps[j].z - pr.lo[2]) <= pr.width[2]);
        else {
            const double wx = c.offset[0] + c.scale[0] * (double)T.rps[j].x, wy = c.offset[1] + c.scale[1] * (double)T.rps[j].y,
                         wz = c.offset[2] + c.scale[2] * (double)T.rps[j].z;
            pass = !((wx < pr.wmin[0]) | (wy < pr.wmin[1]) | (wz < pr.wmin[2]) | (wx > pr.wmax[0]) | (wy > pr.wmax[1]) | (wz > pr.wmax[2]));
        }
        T.passes[j] = pass & (i < c.n);
    }
}

// Launch 1: counts[tile] = matches among the tile's 2048 points.
// It also leaves the tile's MATCH BITS: bits[tile * 32 + k], bit b = point k * 64 + b of the tile matched (row j of wave w
// is the tile's points j * 256 + w * 64 ..: word j * 4 + w — file order).  256 bytes per tile, 1 % of what the pass reads;
// the sparse emit below works from them alone.
// PARK (round 4): a THIN tile — 1 .. park_max matches, a box that keeps a tenth of a file in random order — leaves its matches
// themselves behind, {x, y, z, class} as one 16-byte word each, in file order, at park[tile * park_max ..): the emit then reads
// 16 bytes per MATCH instead of the tile's 13 bytes per POINT a second time (k_emit_parked).  The class bytes of all eight points
// are asked for together (no load in a branch); the ranks come from the tile's 32 mask words, summed by one wave.  Only for
// predicates on the positions (park == nullptr otherwise): a class predicate would have to read positions it does not need.
// RGB: the file has a colour block — a second word {red | green << 16, blue, 0, 0} per match, the colours of the tile asked for
// with the class bytes.
template <int KIND, bool RGB>
__global__ __launch_bounds__(BLOCK) void k_tile_counts(DevCols c, DevPred pr, uint64_t *__restrict__ counts, uint64_t *__restrict__ bits,
                                                       uint4 *__restrict__ park, uint32_t park_max) {
    __shared__ uint32_t s_w[WAVES];
    __shared__ uint64_t s_bits[EMIT_ITEMS * WAVES];
    __shared__ uint32_t s_front[EMIT_ITEMS * WAVES];
    TileIn<KIND, false> T;
    tile_load_and_test<KIND, false, false>(c, pr, (uint64_t)blockIdx.x * EMIT_TILE, T);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t masks[EMIT_ITEMS];
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < EMIT_ITEMS; j++) {
        masks[j] = __ballot(T.passes[j]);  // wave-uniform
        cnt += (uint32_t)__popcll(masks[j]);
        if (lane == 0) s_bits[j * WAVES + wave] = masks[j];
    }
    if (lane == 0) s_w[wave] = cnt;
    __syncthreads();
    if (threadIdx.x < EMIT_ITEMS * WAVES) bits[(uint64_t)blockIdx.x * (EMIT_ITEMS * WAVES) + threadIdx.x] = s_bits[threadIdx.x];
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) total += s_w[w];
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
    if (KIND == PCQ_PRED_CLASS || !park || total == 0 || total > park_max) return;  // (the same for the whole workgroup)
    uint32_t cls[EMIT_ITEMS], rg[EMIT_ITEMS], bl[EMIT_ITEMS];
    {
        const uint8_t *clsp = c.cls ? c.cls : c.xyz;
        const uint64_t stride = c.cls ? c.cls_stride : 0;
        const uint32_t mask = c.cls ? 0xffu : 0u;
#pragma unroll
        for (int j = 0; j < EMIT_ITEMS; j++) {
            const uint64_t i0 = (uint64_t)blockIdx.x * EMIT_TILE + (uint64_t)j * BLOCK + threadIdx.x, i = i0 < c.n ? i0 : c.n - 1;
            cls[j] = clsp[i * stride] & mask;  // last.rs:138-142
            rg[j] = bl[j] = 0;
            if (RGB) {  // last.rs:145-153
                const uint8_t *q = c.rgb + i * c.rgb_stride;
                rg[j] = (uint32_t)ld_u16(q) | ((uint32_t)ld_u16(q + 2) << 16);
                bl[j] = ld_u16(q + 4);
            }
        }
    }
    if (wave == 0) {  // matches in front of (row j, wave w), in file order: word j * WAVES + w
        const uint32_t v = lane < (uint32_t)(EMIT_ITEMS * WAVES) ? (uint32_t)__popcll(s_bits[lane]) : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < EMIT_ITEMS * WAVES; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += up;
        }
        if (lane < (uint32_t)(EMIT_ITEMS * WAVES)) s_front[lane] = incl - v;
    }
    __syncthreads();
    uint4 *dst = park + (uint64_t)blockIdx.x * park_max * (RGB ? 2 : 1);
#pragma unroll
    for (int j = 0; j < EMIT_ITEMS; j++) {
        if (!T.passes[j]) continue;
        const uint32_t rank = s_front[j * WAVES + wave] + (uint32_t)__popcll(masks[j] & ((1ull << lane) - 1ull));
        dst[rank] = make_uint4((uint32_t)T.rps[j].x, (uint32_t)T.rps[j].y, (uint32_t)T.rps[j].z, cls[j]);
        if (RGB) dst[park_max + rank] = make_uint4(rg[j], bl[j], 0u, 0u);
    }
}

// Launch 2 (three small kernels): exclusive prefix of the tile counts; out[n] = the total.
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
// in place over up to 1024 piece sums: sums[i] = sum of the pieces in front of i
__global__ __launch_bounds__(1024) void k_scan_sums(uint64_t *__restrict__ sums, uint32_t n) {
    __shared__ uint64_t s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t v = threadIdx.x < n ? sums[threadIdx.x] : 0;
    uint64_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t up = __shfl_up((unsigned long long)incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t run = incl - v;
    for (int w = 0; w < wave; w++) run += s_wave[w];
    if (threadIdx.x < n) sums[threadIdx.x] = run;
}
// out[i] = piece_prefix[piece] + exclusive prefix inside the piece (thread t owns 4 consecutive elements); out[n] = total
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

// Launch 3: tile t's matches go to records [*d_npoints_in + offsets[t], ...), in file order; tile 0 stores the new count.
template <int KIND, bool RGB>
__global__ __launch_bounds__(BLOCK) void k_emit_points(DevCols c, DevPred pr, const uint64_t *__restrict__ offsets,
                                                       const uint64_t *__restrict__ d_npoints_in, uint64_t *__restrict__ d_npoints_out,
                                                       uint8_t *__restrict__ out31, uint32_t ntiles, uint32_t sparse_max) {
    __shared__ uint32_t s_rw[EMIT_ITEMS][WAVES];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[STAGE_WORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x;
    const uint64_t base_count = *d_npoints_in, before = offsets[tile];
    if (tile == 0 && threadIdx.x == 0) *d_npoints_out = base_count + offsets[ntiles];
    // launch 1 found no match in this tile: nothing of it is read a second time (a box that cuts a flight-line-ordered
    // file leaves most tiles empty; block-uniform, in front of the first load and the first barrier)
    const uint64_t in_tile = offsets[tile + 1] - before;
    if (in_tile <= sparse_max) return;  // (none, or few: k_emit_sparse writes those from the match bits)
    // the image a flush assembles is [pad, pad + 31 * matches of the flush): nothing behind the tile's matches is ever OR-ed
    const uint32_t stage_words = (uint32_t)min((uint64_t)STAGE_WORDS, (16 + 31 * in_tile) / 4 + 10);
    for (uint32_t t = threadIdx.x; t < stage_words; t += BLOCK) s_stage[t] = 0;
    {
        const uint64_t base = (uint64_t)tile * EMIT_TILE;
        TileIn<KIND, true> T;
        tile_load_and_test<KIND, true, RGB>(c, pr, base, T);
        RawPoint (&rps)[EMIT_ITEMS] = T.rps;
        bool (&passes)[EMIT_ITEMS] = T.passes;
        uint32_t (&attr_cls)[EMIT_ITEMS] = T.attr_cls, (&attr_rg)[EMIT_ITEMS] = T.attr_rg, (&attr_b)[EMIT_ITEMS] = T.attr_b;
        if (KIND == PCQ_PRED_CLASS) {  // positions only of the matches (last.rs:265-269), all requested together
#pragma unroll
            for (int j = 0; j < EMIT_ITEMS; j++)
                if (passes[j]) rps[j] = ld_xyz(c, base + (uint64_t)j * BLOCK + threadIdx.x);
        }
        // ranks: the matches of row j in wave w sit behind those of the rows in front and of the waves in front in row j
        uint64_t masks[EMIT_ITEMS];
#pragma unroll
        for (int j = 0; j < EMIT_ITEMS; j++) {
            masks[j] = __ballot(passes[j]);
            if (lane == 0) s_rw[j][wave] = (uint32_t)__popcll(masks[j]);
        }
        __syncthreads();
        uint32_t row_front[EMIT_ITEMS] = {};  // matches of the tile in front of this thread's wave in row j
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < EMIT_ITEMS; j++) {
            uint32_t in_row = 0;
#pragma unroll
            for (int w = 0; w < WAVES; w++) {
                const uint32_t v = s_rw[j][w];
                row_front[j] = w == wave ? total + in_row : row_front[j];
                in_row += v;
            }
            total += in_row;
        }
        uint64_t run = base_count + before;  // record index of the block's next match
        uint32_t flushed = 0;                // matches of the tile already written
#pragma unroll  // rps[] / passes[] must be indexed statically to stay in registers
        for (int h = 0; h < EMIT_ITEMS / FLUSH_ITEMS; h++) {
            const uint64_t gbyte0 = run * 31ull;
            const uint32_t pad = (uint32_t)(gbyte0 & 15);
            uint32_t seg = 0;  // matches of this flush (block-uniform)
#pragma unroll
            for (int jj = 0; jj < FLUSH_ITEMS; jj++) {
                const int j = h * FLUSH_ITEMS + jj;
                uint32_t in_row = 0;
#pragma unroll
                for (int w = 0; w < WAVES; w++) in_row += s_rw[j][w];
                seg += in_row;
                if (!passes[j]) continue;
                const RawPoint rp = rps[j];
                const uint32_t rank = row_front[j] - flushed + (uint32_t)__popcll(masks[j] & ((1ull << lane) - 1ull));
                pcq_point pt;
                pt.x = world(rp.x, c.scale[0], c.offset[0]);  // last.rs:156-160
                pt.y = world(rp.y, c.scale[1], c.offset[1]);
                pt.z = world(rp.z, c.scale[2], c.offset[2]);
                pt.r = (uint16_t)attr_rg[j], pt.g = (uint16_t)(attr_rg[j] >> 16), pt.b = (uint16_t)attr_b[j];
                pt.classification = (uint8_t)attr_cls[j];
                or_point31(s_stage, pad + 31u * rank, pt);
            }
            __syncthreads();
            const uint32_t total_b = pad + seg * 31u;  // staged image is [pad, total_b)
            uint8_t *gdst = out31 + (gbyte0 - pad);     // 16-byte aligned (out31 comes from the pool)
            const uint8_t *stage8 = reinterpret_cast<const uint8_t *>(s_stage);
            for (uint32_t b0 = threadIdx.x * 16u; b0 < total_b; b0 += BLOCK * 16u) {
                const uint32_t b1 = b0 + 16u;
                if (b0 >= pad && b1 <= total_b) {
                    {  // written once, read by nobody on the GPU: streamed past the caches' retention
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const uint4 v = *reinterpret_cast<const uint4 *>(stage8 + b0);
                        u32x4 w = {v.x, v.y, v.z, v.w};
                        __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(gdst + b0));
                    }
                } else {
                    const uint32_t lo = b0 > pad ? b0 : pad, hi = b1 < total_b ? b1 : total_b;
                    for (uint32_t k = lo; k < hi; k++) gdst[k] = stage8[k];
                }
                *reinterpret_cast<uint4 *>(s_stage + b0 / 4) = make_uint4(0, 0, 0, 0);  // the image is zero again for the next flush
            }
            run += seg;
            flushed += seg;
            __syncthreads();  // the stage is reused by the next flush
        }
    }
}

// A PARKED tile (k_tile_counts left its 1 .. park_max <= 256 matches as 16-byte words): thread t builds record t, the records are
// assembled in the LDS image and leave as 16-byte stores, exactly like a flush of k_emit_points — 16 bytes read per match, nothing
// of the tile's 2048 points (RGB: 32).
template <bool RGB>
__global__ __launch_bounds__(BLOCK) void k_emit_parked(DevCols c, const uint64_t *__restrict__ offsets, const uint4 *__restrict__ park,
                                                       const uint64_t *__restrict__ d_npoints_in, uint8_t *__restrict__ out31, uint32_t park_max) {
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[(BLOCK * 31 + 16) / 4 + 16];
    const uint32_t tile = blockIdx.x;
    const uint64_t before = offsets[tile], in_tile = offsets[tile + 1] - before;
    if (in_tile == 0 || in_tile > park_max) return;  // (the same for the whole workgroup)
    uint4 rec = make_uint4(0, 0, 0, 0), col = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < in_tile) {
        const uint4 *src = park + (uint64_t)tile * park_max * (RGB ? 2 : 1);
        rec = src[threadIdx.x];
        if (RGB) col = src[park_max + threadIdx.x];
    }
    const uint64_t gbyte0 = (*d_npoints_in + before) * 31ull;
    const uint32_t pad = (uint32_t)(gbyte0 & 15), total_b = pad + (uint32_t)in_tile * 31u;
    for (uint32_t t = threadIdx.x; t < total_b / 4 + 10; t += BLOCK) s_stage[t] = 0;
    __syncthreads();
    if (threadIdx.x < in_tile) {
        pcq_point pt;
        pt.x = world((int32_t)rec.x, c.scale[0], c.offset[0]);  // last.rs:156-160
        pt.y = world((int32_t)rec.y, c.scale[1], c.offset[1]);
        pt.z = world((int32_t)rec.z, c.scale[2], c.offset[2]);
        pt.r = (uint16_t)col.x, pt.g = (uint16_t)(col.x >> 16), pt.b = (uint16_t)col.y;
        pt.classification = (uint8_t)rec.w;
        or_point31(s_stage, pad + 31u * threadIdx.x, pt);
    }
    __syncthreads();
    uint8_t *gdst = out31 + (gbyte0 - pad);  // 16-byte aligned (out31 comes from the pool)
    const uint8_t *stage8 = reinterpret_cast<const uint8_t *>(s_stage);
    for (uint32_t b0 = threadIdx.x * 16u; b0 < total_b; b0 += BLOCK * 16u) {
        const uint32_t b1 = b0 + 16u;
        if (b0 >= pad && b1 <= total_b) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const uint4 v = *reinterpret_cast<const uint4 *>(stage8 + b0);
            u32x4 w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(gdst + b0));
        } else {
            const uint32_t lo = b0 > pad ? b0 : pad, hi = b1 < total_b ? b1 : total_b;
            for (uint32_t k = lo; k < hi; k++) gdst[k] = stage8[k];
        }
    }
}

// The same for a tile with FEW matches (a box that keeps a per cent of a file in random order: 20 of a tile's 2048 points).
// k_emit_points pays a tile's fixed work whatever it keeps — both readings of all 2048 positions, five barriers, the LDS
// image: 0.72 ms for 1.6 M records against 0.29 ms for the count pass alone.  Here ONE WAVE takes a tile and only its match
// bits (launch 1 left them): lane L owns points 32 L .. 32 L + 31, ranks come from a scan of the lanes' popcounts, and a
// match's position and attributes are loaded and its 31 bytes stored straight from registers.  Nothing else of the tile is read.
template <bool RGB>
__global__ __launch_bounds__(BLOCK) void k_emit_sparse(DevCols c, const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ bits,
                                                       const uint64_t *__restrict__ d_npoints_in, uint8_t *__restrict__ out31, uint32_t ntiles,
                                                       uint32_t parked_max, uint32_t sparse_max) {
    const uint32_t lane = threadIdx.x & 63, tile = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const uint64_t before = offsets[tile], in_tile = offsets[tile + 1] - before;
    if (in_tile <= parked_max || in_tile > sparse_max) return;  // (the same for the whole wave; parked_max = 0: nothing is parked)
    const uint64_t word = bits[(uint64_t)tile * (EMIT_ITEMS * WAVES) + (lane >> 1)];
    uint32_t m = (uint32_t)(word >> (32 * (lane & 1)));
    const uint32_t mine = (uint32_t)__popc(m);
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += up;
    }
    uint64_t rec = *d_npoints_in + before + (incl - mine);  // this lane's first record
    const uint64_t base = (uint64_t)tile * EMIT_TILE + 32u * lane;
    while (m) {
        const uint32_t b = (uint32_t)__ffs((int)m) - 1;
        m &= m - 1;
        const uint64_t i = base + b;
        pcq_point pt;
        make_point(c, i, ld_xyz(c, i), pt);  // last.rs:137-163
        if (!RGB) pt.r = pt.g = pt.b = 0;
        store_point31(out31 + rec * 31ull, pt);
        rec++;
    }
}

}  // namespace

int pcq_launch_generic_count(pcq_ctx *ctx, const DevCols &cols, const DevPred &pred, uint64_t *d_count,
                             hipStream_t s) {
    if (cols.n == 0) return PCQ_OK;
    if (pred.kind == PCQ_PRED_BOUNDS && pred.empty) return PCQ_OK;
    uint64_t want = (cols.n + BLOCK * 4 - 1) / (BLOCK * 4);
    const uint64_t cap = (uint64_t)ctx->num_cus * (uint64_t)ctx->grid_blocks_per_cu;
    const int grid = (int)(want < cap ? want : cap);
    int rc = pcq_ensure_partials(ctx, (size_t)grid);
    if (rc) return rc;
    if (pred.kind == PCQ_PRED_BOUNDS) hipLaunchKernelGGL(k_generic_count<PCQ_PRED_BOUNDS>, dim3(grid), dim3(BLOCK), 0, s, cols, pred, ctx->d_partials);
    else if (pred.kind == PCQ_PRED_CLASS) hipLaunchKernelGGL(k_generic_count<PCQ_PRED_CLASS>, dim3(grid), dim3(BLOCK), 0, s, cols, pred, ctx->d_partials);
    else hipLaunchKernelGGL(k_generic_count<PCQ_PRED_BOUNDS_F64>, dim3(grid), dim3(BLOCK), 0, s, cols, pred, ctx->d_partials);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(BLOCK), 0, s, ctx->d_partials, grid, d_count);
    PCQ_HIP(hipGetLastError());
    return PCQ_OK;
}

// Appends the matches of `cols` to the packed records at d_out31, in file order.  *d_npoints_in (device) is the number
// of records in front of them; the emit stores the new count in *d_npoints_out.  Asynchronous: five launches, no copy.
int pcq_launch_emit_points(pcq_ctx *ctx, const DevCols &cols, const DevPred &pred, uint8_t *d_out31, const uint64_t *d_npoints_in,
                           uint64_t *d_npoints_out, hipStream_t s) {
    if (cols.n == 0) return PCQ_OK;
    const uint64_t ntiles = (cols.n + EMIT_TILE - 1) / EMIT_TILE;
    const uint64_t npieces = (ntiles + SCAN_PIECE - 1) / SCAN_PIECE;
    if (npieces > 1024) return pcq_fail(PCQ_ERR_ARG, "scan chunk too large (%llu points)", (unsigned long long)cols.n);
    // thin tiles park their matches (k_tile_counts): positions predicate; 16 (with a colour block: 32) bytes x park_max per tile, a sixth of the input
    uint32_t park_max = pred.kind != PCQ_PRED_CLASS && cols.xyz && ctx->emit_park_max > 0 ? (uint32_t)ctx->emit_park_max : 0u;
    const size_t base_words = (size_t)(2 * ntiles + npieces + 2 + ntiles * (EMIT_ITEMS * WAVES));  // counts | offsets (+ total) | piece sums | match bits
    int rc = pcq_ensure_partials(ctx, base_words + 2 + (size_t)ntiles * park_max * (cols.rgb ? 4 : 2));  // | parked matches (16-byte aligned)
    if (rc && park_max) {  // no room for the parked matches (a sixth of the input): the thin tiles are read a second time instead
        (void)hipGetLastError();
        park_max = 0;
        rc = pcq_ensure_partials(ctx, base_words + 2);
    }
    if (rc) return rc;
    uint64_t *counts = ctx->d_partials, *offsets = counts + ntiles, *pieces = offsets + ntiles + 1, *bits = pieces + npieces + 1;
    uint4 *park = park_max ? reinterpret_cast<uint4 *>(((uintptr_t)(ctx->d_partials + base_words) + 15) & ~(uintptr_t)15) : nullptr;
    const uint32_t sparse_max = ctx->emit_sparse_max < 0 ? 0u : (uint32_t)ctx->emit_sparse_max;
    const dim3 g((unsigned)ntiles), b(BLOCK);
    const bool park_rgb = park_max && cols.rgb;
    if (pred.kind == PCQ_PRED_CLASS) hipLaunchKernelGGL((k_tile_counts<PCQ_PRED_CLASS, false>), g, b, 0, s, cols, pred, counts, bits, park, park_max);
    else if (pred.kind == PCQ_PRED_BOUNDS && park_rgb) hipLaunchKernelGGL((k_tile_counts<PCQ_PRED_BOUNDS, true>), g, b, 0, s, cols, pred, counts, bits, park, park_max);
    else if (pred.kind == PCQ_PRED_BOUNDS) hipLaunchKernelGGL((k_tile_counts<PCQ_PRED_BOUNDS, false>), g, b, 0, s, cols, pred, counts, bits, park, park_max);
    else if (park_rgb) hipLaunchKernelGGL((k_tile_counts<PCQ_PRED_BOUNDS_F64, true>), g, b, 0, s, cols, pred, counts, bits, park, park_max);
    else hipLaunchKernelGGL((k_tile_counts<PCQ_PRED_BOUNDS_F64, false>), g, b, 0, s, cols, pred, counts, bits, park, park_max);
    hipLaunchKernelGGL(k_scan_piece_sums, dim3((unsigned)npieces), dim3(1024), 0, s, counts, (uint32_t)ntiles, pieces);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, pieces, (uint32_t)npieces);
    hipLaunchKernelGGL(k_scan_pieces, dim3((unsigned)npieces), dim3(1024), 0, s, counts, (uint32_t)ntiles, pieces, offsets);
    const uint32_t skip_max = park_max > sparse_max ? park_max : sparse_max;  // tiles k_emit_points leaves to the two writers below
#define PCQ_EMIT(KIND)                                                                                                                                   \
    do {                                                                                                                                                 \
        if (cols.rgb) hipLaunchKernelGGL((k_emit_points<KIND, true>), g, b, 0, s, cols, pred, offsets, d_npoints_in, d_npoints_out, d_out31, (uint32_t)ntiles, skip_max); \
        else hipLaunchKernelGGL((k_emit_points<KIND, false>), g, b, 0, s, cols, pred, offsets, d_npoints_in, d_npoints_out, d_out31, (uint32_t)ntiles, skip_max);  \
    } while (0)
    if (pred.kind == PCQ_PRED_BOUNDS) PCQ_EMIT(PCQ_PRED_BOUNDS);
    else if (pred.kind == PCQ_PRED_CLASS) PCQ_EMIT(PCQ_PRED_CLASS);
    else PCQ_EMIT(PCQ_PRED_BOUNDS_F64);
#undef PCQ_EMIT
    if (park_max) {  // tiles with 1 .. park_max matches: from the 16-byte words the count pass left
        if (cols.rgb) hipLaunchKernelGGL(k_emit_parked<true>, g, b, 0, s, cols, offsets, park, d_npoints_in, d_out31, park_max);
        else hipLaunchKernelGGL(k_emit_parked<false>, g, b, 0, s, cols, offsets, park, d_npoints_in, d_out31, park_max);
    }
    if (sparse_max > park_max) {  // tiles with 1 .. sparse_max matches: a wave each, from the match bits (same stream: behind the scan of the offsets)
        const dim3 gs((unsigned)((ntiles + WAVES - 1) / WAVES));
        if (cols.rgb) hipLaunchKernelGGL(k_emit_sparse<true>, gs, b, 0, s, cols, offsets, bits, d_npoints_in, d_out31, (uint32_t)ntiles, park_max, sparse_max);
        else hipLaunchKernelGGL(k_emit_sparse<false>, gs, b, 0, s, cols, offsets, bits, d_npoints_in, d_out31, (uint32_t)ntiles, park_max, sparse_max);
    }
    PCQ_HIP(hipGetLastError());
    return PCQ_OK;
}
