// scan_generic.hip — layout-generic scan kernels (any column strides: LAST column blocks or LAS
// AoS records), used where the count-only fast paths of scan_count.hip do not apply:
//   * strided count            — LAS bounds/class count (las.rs:101-119, :221-231), unaligned LAST
//                                (one record per lane: 4.1-5.9 TB/s of record bytes; an LDS-tiled variant
//                                with 16-byte coalesced loads was measured and was NOT faster,
//                                profiles/r01_las_aos_count_rate.log)
//   * order-preserving emit    — BufferCollector semantics (collect_points.rs:29-31): matches are
//                                appended in file order, as 31-byte Point records built like
//                                last.rs:137-163.
// The emit is a two-pass stable stream compaction: (1) per-tile match counts, (2) exclusive scan of
// the tile counts, (3) re-evaluate the predicate and write each match at
// tile_offset + rank-in-tile, where the rank comes from a wave64 ballot prefix
// (popcount(mask & lanes_below)) plus an LDS prefix over the block's four waves.
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

// Pass 1: block b owns points [b*TILE, (b+1)*TILE); counts[b] = matches in the tile.
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_tile_counts(DevCols c, DevPred pr, uint64_t *__restrict__ counts) {
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    uint32_t cnt = 0;
    bool m[ITEMS];
    // all ITEMS loads of a thread are issued before the first compare (index clamped instead of branching)
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
        m[j] = eval_pred_kind<KIND>(c, pr, i < c.n ? i : c.n - 1) & (i < c.n);
    }
#pragma unroll
    for (int j = 0; j < ITEMS; j++) cnt += (uint32_t)__popcll(__ballot(m[j]));  // wave-uniform
    __shared__ uint32_t s_w[WAVES];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < WAVES; i++) t += s_w[i];
        counts[blockIdx.x] = t;
    }
}

// Pass 2: in-place exclusive scan of counts[0..n) by one block; total -> *total_out.
__global__ __launch_bounds__(1024) void k_exclusive_scan(uint64_t *__restrict__ counts, uint64_t n,
                                                         uint64_t *__restrict__ total_out) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < n ? counts[i] : 0;
        uint64_t incl = v;  // inclusive scan inside the wave
#pragma unroll
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

// Pass 3: write each match of tile b at out[(out_base + offsets[b] + rank) * 31].
// 31-byte records at a 31-byte pitch are hostile to per-lane stores (31 byte-stores per match, 16
// cache lines touched per wave-instruction).  The block instead assembles the records of 1024 input
// points in LDS, laid out congruent (mod 16) to their final global position, and then streams the
// contiguous byte range out with 16-byte stores; only the two ragged ends use byte stores, so
// neighbouring blocks never write the same 16-byte chunk.
constexpr int FLUSH_ITEMS = 4;                                  // input rows of 256 points per LDS flush
constexpr int STAGE_BYTES = FLUSH_ITEMS * BLOCK * 31 + 16;      // 31,760 B: five blocks per CU fit in LDS

template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_emit_points(DevCols c, DevPred pr, const uint64_t *__restrict__ offsets,
                                                       uint8_t *__restrict__ out31, uint64_t out_base) {
    __shared__ uint32_t s_w[WAVES];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[STAGE_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    uint64_t run = out_base + offsets[blockIdx.x];  // record index of the block's next match
#pragma unroll 1
    for (int h = 0; h < ITEMS / FLUSH_ITEMS; h++) {
        const uint64_t gbyte0 = run * 31ull;
        const uint32_t pad = (uint32_t)(gbyte0 & 15);
        uint32_t seg = 0;  // matches staged so far (block-uniform)
        // the predicate inputs of the FLUSH_ITEMS rows are loaded together, then the rows are ranked one by one
        RawPoint rps[FLUSH_ITEMS];
        bool passes[FLUSH_ITEMS];
#pragma unroll
        for (int jj = 0; jj < FLUSH_ITEMS; jj++) {
            const uint64_t i = base + (uint64_t)(h * FLUSH_ITEMS + jj) * BLOCK + threadIdx.x;
            passes[jj] = eval_pred_kind<KIND>(c, pr, i < c.n ? i : c.n - 1, rps[jj]) & (i < c.n);
        }
#pragma unroll
        for (int jj = 0; jj < FLUSH_ITEMS; jj++) {
            const int j = h * FLUSH_ITEMS + jj;
            const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
            RawPoint rp = rps[jj];
            const bool have = KIND != PCQ_PRED_CLASS;
            const bool pass = passes[jj];
            const uint64_t mask = __ballot(pass);
            if (lane == 0) s_w[wave] = (uint32_t)__popcll(mask);
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < WAVES; w++) {
                const uint32_t v = s_w[w];
                before += w < wave ? v : 0;
                all += v;
            }
            if (pass) {
                const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                if (!have) rp = ld_xyz(c, i);
                pcq_point pt;
                make_point(c, i, rp, pt);
                store_point31(s_stage + pad + 31u * (seg + before + rank), pt);
            }
            seg += all;
            __syncthreads();
        }
        const uint32_t total = pad + seg * 31u;  // staged image is [pad, total)
        uint8_t *gdst = out31 + (gbyte0 - pad);   // 16-byte aligned (out31 comes from hipMalloc)
        for (uint32_t b0 = threadIdx.x * 16u; b0 < total; b0 += BLOCK * 16u) {
            const uint32_t b1 = b0 + 16u;
            if (b0 >= pad && b1 <= total) {
                *reinterpret_cast<uint4 *>(gdst + b0) = *reinterpret_cast<const uint4 *>(s_stage + b0);
            } else {
                const uint32_t lo = b0 > pad ? b0 : pad, hi = b1 < total ? b1 : total;
                for (uint32_t k = lo; k < hi; k++) gdst[k] = s_stage[k];
            }
        }
        run += seg;
        __syncthreads();  // the stage is reused by the next half
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

// Runs passes 1+2 and returns the number of matches (synchronises `s`).
int pcq_emit_prepare(pcq_ctx *ctx, const DevCols &cols, const DevPred &pred, uint64_t *matches, hipStream_t s) {
    *matches = 0;
    if (cols.n == 0) return PCQ_OK;
    const uint64_t nblocks = (cols.n + TILE - 1) / TILE;
    if (nblocks > 0x7fffffffull) return pcq_fail(PCQ_ERR_ARG, "scan chunk too large (%llu points)", (unsigned long long)cols.n);
    int rc = pcq_ensure_partials(ctx, (size_t)nblocks);
    if (rc) return rc;
    if (pred.kind == PCQ_PRED_BOUNDS) hipLaunchKernelGGL(k_tile_counts<PCQ_PRED_BOUNDS>, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, cols, pred, ctx->d_partials);
    else if (pred.kind == PCQ_PRED_CLASS) hipLaunchKernelGGL(k_tile_counts<PCQ_PRED_CLASS>, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, cols, pred, ctx->d_partials);
    else hipLaunchKernelGGL(k_tile_counts<PCQ_PRED_BOUNDS_F64>, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, cols, pred, ctx->d_partials);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, ctx->d_partials, nblocks, ctx->d_scalars);
    PCQ_HIP(hipGetLastError());
    PCQ_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    PCQ_HIP(hipStreamSynchronize(s));
    *matches = ctx->h_scalars[0];
    return PCQ_OK;
}

int pcq_launch_emit_points(pcq_ctx *ctx, const DevCols &cols, const DevPred &pred, uint8_t *d_out31,
                           uint64_t out_base, uint64_t expected, hipStream_t s) {
    (void)expected;
    if (cols.n == 0) return PCQ_OK;
    const uint64_t nblocks = (cols.n + TILE - 1) / TILE;
    if (pred.kind == PCQ_PRED_BOUNDS)
        hipLaunchKernelGGL(k_emit_points<PCQ_PRED_BOUNDS>, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, cols, pred, ctx->d_partials, d_out31, out_base);
    else if (pred.kind == PCQ_PRED_CLASS)
        hipLaunchKernelGGL(k_emit_points<PCQ_PRED_CLASS>, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, cols, pred, ctx->d_partials, d_out31, out_base);
    else
        hipLaunchKernelGGL(k_emit_points<PCQ_PRED_BOUNDS_F64>, dim3((unsigned)nblocks), dim3(BLOCK), 0, s, cols, pred, ctx->d_partials, d_out31, out_base);
    PCQ_HIP(hipGetLastError());
    return PCQ_OK;
}
