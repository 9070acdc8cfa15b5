#include "grid_common.h"

namespace pcqgrid {

// ---------------------------------------------------------------------------------------------------------------
// second partition level: one workgroup per level-1 bin cuts the bin's tuples (and, when the fan-out changes, the
// earlier winners of the bin) into f2 partitions by the next bits of hash(key).
// ---------------------------------------------------------------------------------------------------------------

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
        bin_for_each<L2_NT, L2_FB, L2_UNROLL>(P.src, P.entries, bin, s_pre, s_addr, [&](const GridTuple &, const RawTuple &r, bool rw) {
            atomicAdd(&s_hist[raw_sub(P.g, P.entries, r, rw, f2)], 1u);
        });
    if (P.okeys)
        for (uint32_t q = bin * P.f2old; q < (bin + 1) * P.f2old; q++) {
            const uint64_t base = P.obase[q];
            const uint32_t n = P.ocount[q];
            for (uint32_t i = threadIdx.x; i < n; i += L2_NT) atomicAdd(&s_ohist[sub_of(cell_hash(P.okeys[base + i], P.g.keys_wide), f2)], 1u);
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
        bin_for_each<L2_NT, L2_FB, L2_UNROLL>(P.src, P.entries, bin, s_pre, s_addr, [&](const GridTuple &, const RawTuple &r, bool rw) {
            const uint32_t pos = atomicAdd(&s_cur[raw_sub(P.g, P.entries, r, rw, f2)], 1u);
            st_raw_as(P.out + (uint64_t)pos * tuple_bytes(wide), r, rw, wide, P.entries);
        });
    }
    if (P.okeys)
        for (uint32_t q = bin * P.f2old; q < (bin + 1) * P.f2old; q++) {
            const uint64_t base = P.obase[q];
            const uint32_t n = P.ocount[q];
            for (uint32_t i = threadIdx.x; i < n; i += L2_NT) {
                const uint64_t key = P.okeys[base + i];
                const uint32_t pos = atomicAdd(&s_ocur[sub_of(cell_hash(key, P.g.keys_wide), f2)], 1u);
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
// ANYWIDE = false: every pending run has 16-byte tuples, and so has the output (four registers per tuple in flight, one aligned
// load, one aligned store); MULTI = false: one entry — no load in a branch between the prefetch of the next tile and its use
// (EntryRef::get).  Everything else: <true, true>.
// NT: threads per workgroup — 1024: one workgroup per CU, tiles of 4096 tuples.  (512 — TWO workgroups per CU, tiles of 2048,
// one's barriers and one-wave scan under the other's loads and stores — was measured 0.3 ms SLOWER on ca13 XL at 10 m:
// DESIGN.md section 10.)
template <bool ANYWIDE, bool MULTI, int NT>
__global__ __launch_bounds__(NT, NT == 1024 ? 4 : 4) void k_level2(Level2Params P) {
    constexpr int L2S_NT = NT, L2S_ITEMS = 4, L2S_TILE = NT * 4, L2S_FB = NT * 2;
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
            for (uint32_t i = threadIdx.x; i < n; i += L2S_NT) atomicAdd(&s_ohist[sub_of(cell_hash(P.okeys[base + i], P.g.keys_wide), f2)], 1u);
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
        const bool wide_out = ANYWIDE && P.wide;
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
                // The tuples travel as the words they are in memory (a 16-byte tuple: four registers, one aligned load, one aligned
                // store); only the sub-partition is computed from the decoded form.
                RawTuple t[L2S_ITEMS], tn[L2S_ITEMS];
                bool tw[L2S_ITEMS], tnw[L2S_ITEMS];
#pragma unroll
                for (int j = 0; j < L2S_ITEMS; j++) {
                    const uint32_t i = j0 + j * L2S_NT + threadIdx.x;
                    t[j] = frag_ld_raw<ANYWIDE>(s_pre, s_addr, nfr, i < hi ? i : hi - 1, &tw[j]);
                    tn[j] = t[j], tnw[j] = tw[j];
                }
#pragma unroll
                for (int j = 0; j < L2S_ITEMS; j++)  // (arrived: see k_p0_part on the one counter for loads and stores)
                    if (ANYWIDE) asm volatile("" ::"v"(t[j].a.x), "v"(t[j].a.w), "v"(t[j].b.x), "v"(t[j].b.y));
                    else asm volatile("" ::"v"(t[j].a.x), "v"(t[j].a.w));
                for (uint32_t base = j0; base < hi; base += L2S_TILE) {
                    if (base + L2S_TILE < hi) {  // the next tile is on its way while this one is sorted
#pragma unroll
                        for (int j = 0; j < L2S_ITEMS; j++) {
                            const uint32_t i = base + L2S_TILE + j * L2S_NT + threadIdx.x;
                            tn[j] = frag_ld_raw<ANYWIDE>(s_pre, s_addr, nfr, i < hi ? i : hi - 1, &tnw[j]);
                        }
                    }
                    uint32_t subs[L2S_ITEMS], ranks[L2S_ITEMS];
                    bool valid[L2S_ITEMS];
                    // (grid and entry out of the argument segment, here: see karg())
                    const KArgPtr ka = karg_base();
                    const DevGrid G = karg<DevGrid>(ka, offsetof(Level2Params, g));
                    const EntryRef E = karg<EntryRef>(ka, offsetof(Level2Params, entries));
#pragma unroll
                    for (int j = 0; j < L2S_ITEMS; j++) {
                        valid[j] = base + j * L2S_NT + threadIdx.x < hi;
                        subs[j] = raw_sub<MULTI>(G, E, t[j], tw[j], f2);
                        ranks[j] = 0;
                        if (valid[j]) ranks[j] = atomicAdd(&s_cnt[subs[j]], 1u);
                    }
                    __syncthreads();
                    if (wave == 0) {  // exclusive scan of the tile's counts over the sub-partitions, by ONE wave (lane l = entries 16 l ..): no barrier inside
                        constexpr int BPL = L2_STAGED_F2 / 64;
                        uint32_t v[BPL], mine = 0;
#pragma unroll
                        for (int q = 0; q < BPL; q++) v[q] = (uint32_t)(lane * BPL + q) < f2 ? s_cnt[lane * BPL + q] : 0u, mine += v[q];  // (nothing counts beyond f2)
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
                        if (wide_out && !tw[j]) {  // a 16-byte tuple into a 24-byte output (some other run is wide): decoded
                            const GridTuple d = decode_raw<MULTI>(t[j], false, E);
                            s_xyzi[at] = make_uint4((uint32_t)d.x, (uint32_t)d.y, (uint32_t)d.z, d.idx);
                            s_attr[at] = make_uint2(d.w0 & 0xffff00ffu, 0u);
                        } else {
                            s_xyzi[at] = make_uint4(t[j].a.x, t[j].a.y, t[j].a.z, t[j].a.w);
                            if (wide_out) s_attr[at] = make_uint2(t[j].b.x, t[j].b.y);
                        }
                        const uint32_t within = s_cur[subs[j]] + ranks[j];  // place in the sub-partition's region
                        s_tpos[at] = within < cap ? subs[j] * cap + within : 0xffffffffu;
                    }
#pragma unroll
                    for (int j = 0; j < L2S_ITEMS; j++) {  // the next tile has arrived — before this tile's stores are issued
                        t[j] = tn[j], tw[j] = tnw[j];
                        if (ANYWIDE) asm volatile("" ::"v"(t[j].a.x), "v"(t[j].a.w), "v"(t[j].b.x), "v"(t[j].b.y));
                    else asm volatile("" ::"v"(t[j].a.x), "v"(t[j].a.w));
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
                            uint8_t *q = P.out + (region0 + tp) * ts_out;
                            if (!wide_out) {
                                u32x4_a16 va = {a.x, a.y, a.z, a.w};
                                *(PCQ_GLOBAL u32x4_a16 *)q = va;
                            } else {
                                const uint2 b = s_attr[k];
                                u32x4_a8 va = {a.x, a.y, a.z, a.w};
                                u32x2_a8 vb = {b.x, b.y};
                                *(PCQ_GLOBAL u32x4_a8 *)q = va;
                                *(PCQ_GLOBAL u32x2_a8 *)(q + 16) = vb;
                            }
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
                const uint32_t pos = atomicAdd(&s_ocur[sub_of(cell_hash(key, P.g.keys_wide), f2)], 1u);
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

template __global__ void k_level2<false, false, 1024>(Level2Params);
template __global__ void k_level2<true, true, 1024>(Level2Params);

}  // namespace pcqgrid
