#include "grid_common.h"

namespace pcqgrid {

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
    auto visit = [&](const GridTuple &t, const RawTuple &, bool) {
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
        bin_for_each<L2_NT, L2_FB, 1>(P.src, P.entries, p, s_pre, s_addr, visit);
    } else {
        const GridSeg sg = P.seg;
        const bool wide = sg.wide;
        const uint32_t lo = sg.off[p], hi = lo + (sg.cnt ? sg.cnt[p] : sg.off[p + 1] - lo);
        for (uint32_t i = lo + threadIdx.x; i < hi; i += L2_NT) visit(ld_tuple(sg.tuples + (uint64_t)i * tuple_bytes(wide), wide, P.entries), RawTuple{}, wide);
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
    const uint64_t h = cell_hash(key, keys_wide_of(P.g));
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

template __global__ void k_alias_gather<false, true>(FoldParams, AliasItem *, unsigned long long *);
template __global__ void k_alias_gather<false, false>(FoldParams, AliasItem *, unsigned long long *);
template __global__ void k_alias_gather<true, true>(FoldParams, AliasItem *, unsigned long long *);
template __global__ void k_alias_gather<true, false>(FoldParams, AliasItem *, unsigned long long *);

}  // namespace pcqgrid
