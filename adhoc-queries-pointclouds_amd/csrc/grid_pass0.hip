#include "grid_common.h"

namespace pcqgrid {

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
// WIDE: the block's tuples are 24 bytes (a colour column, or no axis on which a 16-byte tuple could carry the class byte);
// otherwise 16: {x - lo, y - lo, z - lo, place in the pending stream} with the class in the top byte of one coordinate (pk).
template <int KIND, bool RGB, bool PACKED, bool WIDE>
__global__ __launch_bounds__(P0_NT, 4) void k_p0_part(P0Args A) {
    // (columns, predicate, grid and packing are read out of the argument segment where the tile loop uses them: karg())
    const uint32_t ntiles = A.ntiles, tile0 = A.tile0;
    const int agg_mode = A.agg_mode;
    uint8_t *__restrict__ const out = A.out;
    uint16_t *__restrict__ const dir = A.dir;
    static_assert(WIDE || !RGB, "a colour column needs the 24-byte tuple");
    constexpr int NT = P0_NT, ITEMS = P0_ITEMS;
    constexpr uint32_t TS = WIDE ? 24 : 16;
    constexpr int STAGE_BYTES = P0_TILE * (int)TS;
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
        const uint64_t left = A.c.n - (uint64_t)t * P0_TILE;
        return left < (uint64_t)P0_TILE ? (uint32_t)left : (uint32_t)P0_TILE;
    };
    if (tile < ntiles) {
        const uint32_t nv = tile_points(tile);
#pragma unroll
        for (int j = 0; j < ITEMS; j++) cur[j] = p0_load_tile<KIND, PACKED>(A.c, (uint64_t)tile * P0_TILE, (uint32_t)j * NT + tid, nv);
    }
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {  // (arrived: inside the loop nothing is pending at its head)
        if (KIND == PCQ_PRED_CLASS) asm volatile("" ::"v"(cur[j].cls));
        else asm volatile("" ::"v"(cur[j].rp.x), "v"(cur[j].rp.y), "v"(cur[j].rp.z));
    }
    for (; tile < ntiles; tile += gridDim.x, parity ^= 1) {
        const DevCols c = karg<DevCols>(karg_base(), offsetof(P0Args, c));
        const DevPred pr = karg<DevPred>(karg_base(), offsetof(P0Args, pr));
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
        const DevGrid g = karg<DevGrid>(karg_base(), offsetof(P0Args, g));
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            if (!passes[j]) continue;
            npass++;
            const double px = world(cur[j].rp.x, c.scale[0], c.offset[0]), py = world(cur[j].rp.y, c.scale[1], c.offset[1]),
                         pz = world(cur[j].rp.z, c.scale[2], c.offset[2]);
            if (agg) {
                const TupleEval ev = eval_world(g, px, py, pz);
                const uint64_t h = cell_hash(ev.key, g.keys_wide);
                const uint32_t place = (uint32_t)j * NT + tid;
                s_akey[place] = ev.key;
                const bool through = ev.alias || ev.dbits >= 0x7ff0000000000000ull;
                pk[j] = (through ? 0ull : ((ev.dbits >> AGG_POS_BITS) + 1) << AGG_POS_BITS) | place;
                metas[j] = bin_of(h) | (sel16_of(h) << 16);  // (the bin, and the second level's selector for the spare top bytes of a 16-byte tuple)
                ranks[j] = (uint32_t)(h >> 24) & (AGG_SLOTS - 1);  // the table slot, until the tile's fold is over
            } else {
                const uint64_t h = cell_hash(key_only(g, px, py, pz), g.keys_wide);
                metas[j] = bin_of(h) | (sel16_of(h) << 16);  // (the bin, and the second level's selector for the spare top bytes of a 16-byte tuple)
            }
        }
        if (agg) {
            __syncthreads();  // the table is clear, the keys are in place
            // Where the tile's fold pays, neighbours in the file are neighbours in space: the 64 lanes of a wave aim at a handful of
            // slots, and an LDS atomic takes the lanes of one address one after the other.  So every eighth lane goes first; the
            // others look at the slot afterwards and stay away when they cannot lower it (a minimum only falls: whoever is not
            // below what it reads is not below what will be there at the end — skipping that atomicMin changes nothing in the table).
            const bool scout = (lane & 7u) == 0;
#pragma unroll
            for (int j = 0; j < ITEMS; j++)
                if (passes[j] && scout) atomicMin((unsigned long long *)&s_atab[ranks[j]], (unsigned long long)pk[j]);
#pragma unroll
            for (int j = 0; j < ITEMS; j++)
                if (passes[j] && !scout && (unsigned long long)pk[j] < __hip_atomic_load(&s_atab[ranks[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                    atomicMin((unsigned long long *)&s_atab[ranks[j]], (unsigned long long)pk[j]);
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
        const P0Pack pk16 = karg<P0Pack>(karg_base(), offsetof(P0Args, pk16));
        if (tid < (uint32_t)DIR_WORDS) {  // the directory row, two entries per word
            const uint32_t lo = s_base[2 * tid], hi = 2 * tid + 1 <= (uint32_t)F1 ? s_base[2 * tid + 1] : 0;
            *(PCQ_GLOBAL uint32_t *)(reinterpret_cast<uint32_t *>(dir + (size_t)tile * DIR_STRIDE) + tid) = lo | (hi << 16);
        }
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            if (!passes[j]) continue;
            const uint32_t at = s_base[metas[j] & (F1 - 1)] + ranks[j];
            uint32_t *q = s_img + at * (TS / 4);
            const uint32_t ord = (tile0 + tile) * (uint32_t)P0_TILE + (uint32_t)j * NT + tid;  // the tuple's place in the pending stream
            if (WIDE) {
                uint2 *q2 = reinterpret_cast<uint2 *>(q);  // (a 24-byte stride is 8-byte aligned)
                q2[0] = make_uint2((uint32_t)cur[j].rp.x, (uint32_t)cur[j].rp.y);
                q2[1] = make_uint2((uint32_t)cur[j].rp.z, ord);
                q2[2] = make_uint2(cl[j] | (RGB ? rg[j] << 16 : 0u), RGB ? (rg[j] >> 16) | (bb[j] << 16) : 0u);
            } else {
                // (the class byte is used as it was loaded, HERE: when the shift could move up to the load, the compiler kept one
                // register for the five bytes and waited for each load on the spot — five serial round trips per tile, each of
                // them also waiting for the next tile's positions: 1.1 -> 1.55 ms)
                uint32_t cbyte = cl[j];
                asm volatile("" : "+v"(cbyte));
                const uint32_t V = cbyte | ((metas[j] >> 16) << 8);  // class | sel16 << 8: a byte of it per coordinate top byte that is free
                *reinterpret_cast<uint4 *>(q) = make_uint4(((uint32_t)(cur[j].rp.x - pk16.lo[0]) & pk16.cmask[0]) | (((V >> pk16.top_shift[0]) << 24) & ~pk16.cmask[0]),
                                                           ((uint32_t)(cur[j].rp.y - pk16.lo[1]) & pk16.cmask[1]) | (((V >> pk16.top_shift[1]) << 24) & ~pk16.cmask[1]),
                                                           ((uint32_t)(cur[j].rp.z - pk16.lo[2]) & pk16.cmask[2]) | (((V >> pk16.top_shift[2]) << 24) & ~pk16.cmask[2]), ord);
            }
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
        uint4 *blk = reinterpret_cast<uint4 *>(out + (uint64_t)tile * pk16.block_bytes);
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

// every shape grid_host.hip launches
#define PCQ_P0_INST(KIND)                                                                                                                     \
    template __global__ void k_p0_part<KIND, true, true, true>(P0Args);   \
    template __global__ void k_p0_part<KIND, true, false, true>(P0Args);  \
    template __global__ void k_p0_part<KIND, false, true, true>(P0Args);  \
    template __global__ void k_p0_part<KIND, false, false, true>(P0Args); \
    template __global__ void k_p0_part<KIND, false, true, false>(P0Args); \
    template __global__ void k_p0_part<KIND, false, false, false>(P0Args);
PCQ_P0_INST(PCQ_PRED_BOUNDS)
PCQ_P0_INST(PCQ_PRED_CLASS)
PCQ_P0_INST(PCQ_PRED_BOUNDS_F64)
#undef PCQ_P0_INST

}  // namespace pcqgrid
