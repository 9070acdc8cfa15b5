// scan_generic.hip — layout-generic scan kernels (any column strides: LAST column blocks or LAS
// AoS records), used where the count-only fast paths of scan_count.hip do not apply:
//   * strided count            — LAS bounds/class count (las.rs:101-119, :221-231), unaligned LAST
//                                (one record per lane: 4.1-5.9 TB/s of record bytes; an LDS-tiled variant
//                                with 16-byte coalesced loads was measured and was NOT faster,
//                                profiles/r01_las_aos_count_rate.log)
//   * order-preserving emit    — BufferCollector semantics (collect_points.rs:29-31): matches are
//                                appended in file order, as 31-byte Point records built like
//                                last.rs:137-163.
// The emit is a SINGLE-pass stable stream compaction (decoupled look-back): a workgroup takes the next tile of 2048
// points, evaluates the predicate once, publishes the tile's match count, finds its output offset by looking back
// over the tiles in front of it (their counts, or the running prefix the nearest finished one published), and
// writes each match at offset + rank-in-tile — the rank from a wave64 ballot prefix (popcount(mask & lanes_below))
// plus an LDS prefix over the block's four waves.  The collector's point count lives on the device: the scan
// reads it as its base and the last tile moves it on, so nothing comes back to the host between scans.
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

// Tile states of the look-back: the top two bits say what the low 62 hold.
constexpr uint64_t ST_AGG = 1ull << 62;     // the tile's own match count
constexpr uint64_t ST_PREFIX = 2ull << 62;  // matches of this tile and of every tile in front of it
constexpr uint64_t ST_VALUE = (1ull << 62) - 1;

// 31-byte records at a 31-byte pitch are hostile to per-lane stores (31 byte-stores per match, 16
// cache lines touched per wave-instruction).  The block instead assembles the records of 1024 input
// points in LDS, laid out congruent (mod 16) to their final global position, and then streams the
// contiguous byte range out with 16-byte stores; only the two ragged ends use byte stores, so
// neighbouring blocks never write the same 16-byte chunk.
constexpr int FLUSH_ITEMS = 4;                                  // input rows of 256 points per LDS flush
constexpr int STAGE_BYTES = FLUSH_ITEMS * BLOCK * 31 + 16;      // 31,760 B: five blocks per CU fit in LDS

template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_emit_points(DevCols c, DevPred pr, uint64_t *__restrict__ tile_state, uint32_t *__restrict__ ticket,
                                                       uint64_t *__restrict__ d_npoints, uint8_t *__restrict__ out31, uint32_t ntiles) {
    __shared__ uint32_t s_w[WAVES];
    __shared__ uint32_t s_tile;
    __shared__ uint64_t s_before, s_base;
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[STAGE_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // tiles are handed out in launch order, so every tile a workgroup looks back at belongs to a workgroup that is
    // already running (or done): the wait below always ends
    if (threadIdx.x == 0) {
        s_tile = atomicAdd(ticket, 1u);
        s_base = *d_npoints;  // points in the collector before this scan; read before this tile publishes anything
    }
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t base = (uint64_t)tile * TILE;

    // the predicate, once: all ITEMS loads of a thread are issued before the first compare
    RawPoint rps[ITEMS];
    bool passes[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
        passes[j] = eval_pred_kind<KIND>(c, pr, i < c.n ? i : c.n - 1, rps[j]) & (i < c.n);
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; j++) cnt += (uint32_t)__popcll(__ballot(passes[j]));  // wave-uniform
    if (lane == 0) s_w[wave] = cnt;
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) total += s_w[w];

    if (wave == 0) {  // publish, then look back: 64 predecessors at a time, nearest first
        if (lane == 0)
            __hip_atomic_store(&tile_state[tile], (tile == 0 ? ST_PREFIX : ST_AGG) | (uint64_t)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t before = 0;
        int64_t look = (int64_t)tile - 1;
        while (look >= 0) {
            const int64_t t = look - lane;
            uint64_t v = ST_PREFIX;  // lanes in front of tile 0 read as "prefix 0"
            if (t >= 0) {
                do {
                    v = __hip_atomic_load(&tile_state[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (!(v >> 62)) __builtin_amdgcn_s_sleep(1);
                } while (!(v >> 62));
            }
            const uint64_t has_prefix = __ballot((v & ST_PREFIX) != 0);
            const int stop = has_prefix ? __builtin_ctzll(has_prefix) : 63;  // the nearest tile that knows everything in front of it
            uint64_t part = lane <= stop ? (v & ST_VALUE) : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_down((unsigned long long)part, off, 64);
            before += __shfl((unsigned long long)part, 0, 64);
            if (has_prefix) break;
            look -= 64;
        }
        if (lane == 0) {
            if (tile > 0)
                __hip_atomic_store(&tile_state[tile], ST_PREFIX | (before + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_before = before;
            if (tile == ntiles - 1) *d_npoints = s_base + before + total;  // every tile has read the old value: they all published before this one got here
        }
    }
    __syncthreads();
    uint64_t run = s_base + s_before;  // record index of the block's next match
#pragma unroll 1
    for (int h = 0; h < ITEMS / FLUSH_ITEMS; h++) {
        const uint64_t gbyte0 = run * 31ull;
        const uint32_t pad = (uint32_t)(gbyte0 & 15);
        uint32_t seg = 0;  // matches staged so far (block-uniform)
#pragma unroll
        for (int jj = 0; jj < FLUSH_ITEMS; jj++) {
            const int j = h * FLUSH_ITEMS + jj;
            const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
            RawPoint rp = rps[j];
            const bool have = KIND != PCQ_PRED_CLASS;
            const bool pass = passes[j];
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
        const uint32_t total_b = pad + seg * 31u;  // staged image is [pad, total_b)
        uint8_t *gdst = out31 + (gbyte0 - pad);     // 16-byte aligned (out31 comes from hipMalloc)
        for (uint32_t b0 = threadIdx.x * 16u; b0 < total_b; b0 += BLOCK * 16u) {
            const uint32_t b1 = b0 + 16u;
            if (b0 >= pad && b1 <= total_b) {
                *reinterpret_cast<uint4 *>(gdst + b0) = *reinterpret_cast<const uint4 *>(s_stage + b0);
            } else {
                const uint32_t lo = b0 > pad ? b0 : pad, hi = b1 < total_b ? b1 : total_b;
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

// Appends the matches of `cols` to the packed records at d_out31, in file order.  *d_npoints (device) is the number of
// records in front of them and is moved on by the kernel.  Asynchronous.
int pcq_launch_emit_points(pcq_ctx *ctx, const DevCols &cols, const DevPred &pred, uint8_t *d_out31, uint64_t *d_npoints, hipStream_t s) {
    if (cols.n == 0) return PCQ_OK;
    const uint64_t ntiles = (cols.n + TILE - 1) / TILE;
    if (ntiles > 0x7fffffffull) return pcq_fail(PCQ_ERR_ARG, "scan chunk too large (%llu points)", (unsigned long long)cols.n);
    int rc = pcq_ensure_partials(ctx, (size_t)ntiles + 2);  // tile states + the ticket
    if (rc) return rc;
    PCQ_HIP(hipMemsetAsync(ctx->d_partials, 0, ((size_t)ntiles + 2) * sizeof(uint64_t), s));
    uint64_t *state = ctx->d_partials;
    uint32_t *ticket = reinterpret_cast<uint32_t *>(ctx->d_partials + ntiles);
    const dim3 g((unsigned)ntiles), b(BLOCK);
    if (pred.kind == PCQ_PRED_BOUNDS) hipLaunchKernelGGL(k_emit_points<PCQ_PRED_BOUNDS>, g, b, 0, s, cols, pred, state, ticket, d_npoints, d_out31, (uint32_t)ntiles);
    else if (pred.kind == PCQ_PRED_CLASS) hipLaunchKernelGGL(k_emit_points<PCQ_PRED_CLASS>, g, b, 0, s, cols, pred, state, ticket, d_npoints, d_out31, (uint32_t)ntiles);
    else hipLaunchKernelGGL(k_emit_points<PCQ_PRED_BOUNDS_F64>, g, b, 0, s, cols, pred, state, ticket, d_npoints, d_out31, (uint32_t)ntiles);
    PCQ_HIP(hipGetLastError());
    return PCQ_OK;
}
