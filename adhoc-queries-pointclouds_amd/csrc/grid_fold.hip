#include "grid_common.h"

namespace pcqgrid {

// ---------------------------------------------------------------------------------------------------------------
// the fold: one workgroup per partition, open-addressing table in LDS
// ---------------------------------------------------------------------------------------------------------------

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
        const int s = lds_find_or_insert<NSLOT, LIMIT>(s_key, ev.key, cell_hash(ev.key, keys_wide_of(P.g)), s_ncell);
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
            const int s = lds_find_or_insert<NSLOT, LIMIT>(s_key, key, cell_hash(key, keys_wide_of(P.g)), &s_ncell);
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
                tu[k] = ld_tuple(sg.tuples + (uint64_t)(cur_lo + (i < cnt ? i : (cnt ? cnt - 1 : 0))) * seg_ts, seg_wide, P.entries);
            }
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) {  // phase 1: cells and their minimum distance
                const uint32_t i = k * NT + threadIdx.x;
                slot[k] = -1;
                dbits[k] = 0;
                if (i >= cnt) continue;
                const TupleEval ev = eval_tuple(P.g, P.entries, tu[k]);
                dbits[k] = ev.dbits;
                const int s = lds_find_or_insert<NSLOT, LIMIT>(s_key, ev.key, cell_hash(ev.key, keys_wide_of(P.g)), &s_ncell);
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
                for (int k = 0; k < FOLD_K; k++) tn[k] = GridTuple{0, 0, 0, 0, 0, 0};
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
                                tn[k] = frag_ld_tuple(s_pre, s_addr, nfr, j0 + (i < cnt_n ? i : cnt_n - 1), P.entries);
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
                        tu[k] = ld_tuple(sg.tuples + (uint64_t)(cur_lo + j0 + (i < cnt ? i : cnt - 1)) * seg_ts, seg_wide, P.entries);
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
// fill up (at most a chunk of 1536 tuples goes into 2048 slots), so the loop ends; the cells are counted per wave.  (The
// loop itself is in the kernel: a thread's tuples probe together.)
static_assert((SMALL_SLOTS & (SMALL_SLOTS - 1)) == 0, "the dense fold's probe step visits every slot of a power-of-two table");

// WIDE = false: the second level's output holds 16-byte tuples (one aligned load each); MULTI = false: one entry — no
// load in a branch anywhere between the tuples' loads and their use (EntryRef::get).  Everything else: <true, true>.
template <int NSLOT, int NT, int FOLD_K, int LIMIT, int MIN_WAVES, bool WIDE, bool MULTI>
__global__ __launch_bounds__(NT, MIN_WAVES) void k_fold_dense(DenseParams P, uint32_t nparts) {
    constexpr int CHUNK = NT * FOLD_K;
    __shared__ uint64_t s_key[NSLOT];
    __shared__ uint64_t s_dist[NSLOT];
    __shared__ uint64_t s_ord[NSLOT];
    __shared__ uint32_t s_aliasbits[(NSLOT + 31) / 32];
    __shared__ uint32_t s_tie, s_wsum[NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *tuples = P.tuples;
    const bool wide = WIDE && P.wide;
    const uint32_t ts = tuple_bytes(wide);
    const uint32_t *off = P.off;
    for (int t = threadIdx.x; t < NSLOT; t += NT) s_key[t] = PCQ_EMPTY_KEY, s_dist[t] = ~0ull, s_ord[t] = ~0ull;
    for (int t = threadIdx.x; t < (NSLOT + 31) / 32; t += NT) s_aliasbits[t] = 0;
    if (threadIdx.x == 0) s_tie = 0;
    unsigned long long winners = 0;  // thread 0: this workgroup's winners

    // Three partitions deep: the current one (range, output base, and — 16-byte tuples — its tuples, asked for a whole
    // partition ago), the next one (range known; its tuples are asked for at the head of this round and have arrived by
    // the time this partition's winners are stored), and the one behind it (its range is on its way).  A partition is a
    // chain of latencies — offsets, tuples, three barriers, the stores — and at two workgroups per CU nothing else hid the
    // tuples' round trip.
    uint32_t cur_lo = 0, cur_cnt = 0, nxt_lo = 0, nxt_cnt = 0;
    uint64_t cur_out = 0, nxt_out = 0;
    uint32_t p = blockIdx.x;
    // (the partition's range and output base are the same for the whole workgroup: scalar registers)
    const uint32_t *cntp = P.cnt;
    auto range_of = [&](uint32_t q, uint32_t *lo, uint32_t *cn, uint64_t *ob) {
        *lo = uni32(off[q]), *cn = cntp ? uni32(cntp[q]) : uni32(off[q + 1]) - *lo, *ob = uni64(P.wbase[q]);
    };
    // The same in two halves: asked for at the head of a round (plain loads into vector registers: nothing waits), made
    // scalar at its end.  As one step the round stood still for the loads' round trip — readfirstlane needs the value — with
    // the next partition's tuples queued behind them: 13 % of the kernel (profiles/r04_grid_progress.txt, stamps).
    uint32_t v_lo = 0, v_cn = 0;
    uint64_t v_ob = 0;
    auto range_ask = [&](uint32_t q) {
        v_lo = ldg(off + q), v_cn = cntp ? ldg(cntp + q) : ldg(off + q + 1), v_ob = ldg(P.wbase + q);
    };
    auto range_take = [&](uint32_t *lo, uint32_t *cn, uint64_t *ob) {
        *lo = uni32(v_lo), *cn = cntp ? uni32(v_cn) : uni32(v_cn) - *lo, *ob = uni64(v_ob);
    };
    u32x4_a16 rcur[FOLD_K], rnxt[FOLD_K];
    auto ask16 = [&](u32x4_a16 (&r)[FOLD_K], uint32_t lo, uint32_t cnt) {
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) {
            const uint32_t i = k * NT + threadIdx.x;
            r[k] = *(const PCQ_GLOBAL u32x4_a16 *)(tuples + (uint64_t)(lo + (i < cnt ? i : (cnt ? cnt - 1 : 0))) * 16u);
        }
    };
#pragma unroll
    for (int k = 0; k < FOLD_K; k++) rcur[k] = rnxt[k] = (u32x4_a16){0u, 0u, 0u, 0u};
    if (p < nparts) {
        range_of(p, &cur_lo, &cur_cnt, &cur_out);
        if (!WIDE && cur_cnt <= (uint32_t)CHUNK) ask16(rcur, cur_lo, cur_cnt);
        if (p + gridDim.x < nparts) range_of(p + gridDim.x, &nxt_lo, &nxt_cnt, &nxt_out);
    }
    ST_DECL;
    for (; p < nparts; p += gridDim.x) {
        ST(9);  // (the tail of the round: rotating the registers)
        const uint32_t pn = p + gridDim.x, pnn = pn + gridDim.x;
        if (pnn < nparts) range_ask(pnn);
        const bool ask_next = !WIDE && pn < nparts && nxt_cnt <= (uint32_t)CHUNK;
        if (ask_next) ask16(rnxt, nxt_lo, nxt_cnt);
        const uint32_t cnt = cur_cnt;
        GridTuple tu[FOLD_K];
        if (cnt > (uint32_t)CHUNK) {  // (the same for every thread of the workgroup)
            if (threadIdx.x == 0) P.defer_list[atomicAdd(&P.stats[3], 1ull)] = p;
            cur_lo = nxt_lo, cur_cnt = nxt_cnt, cur_out = nxt_out;
            if (pnn < nparts) range_take(&nxt_lo, &nxt_cnt, &nxt_out);
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) rcur[k] = rnxt[k];
            continue;
        }
        uint64_t key[FOLD_K], dbits[FOLD_K];
        bool alias[FOLD_K];
        ST(0);  // ranges of the partition after next, asking for the next one's tuples
        uint32_t inexact = 0;  // bit k: tuple k is next to a cell boundary (or outside the short computation's range)
        {
            // (grid and entry out of the argument segment, here: see karg())
            const KArgPtr ka = karg_base();
            const DevGridFast GF = karg<DevGridFast>(ka, offsetof(DenseParams, g) + offsetof(GridRef, f));
            const EntryRef E = karg<EntryRef>(ka, offsetof(DenseParams, entries));
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) {
                if (WIDE) {
                    const uint32_t i = k * NT + threadIdx.x;
                    const uint8_t *src = tuples + (uint64_t)(cur_lo + (i < cnt ? i : (cnt ? cnt - 1 : 0))) * ts;
                    tu[k] = ld_tuple<MULTI>(src, wide, E);
                } else {
                    tu[k] = decode16<MULTI>(rcur[k], E);
                }
            }
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) {
                const GridEntryDev e = E.get<MULTI>((tu[k].w0 >> 8) & 0xff);
                const double px = world(tu[k].x, e.scale[0], e.offset[0]), py = world(tu[k].y, e.scale[1], e.offset[1]),
                             pz = world(tu[k].z, e.scale[2], e.offset[2]);
                const CellFast cf = cell_fast(GF, px, py, pz);
                key[k] = key_fast(GF, cf, &alias[k]);
                dbits[k] = (uint64_t)__double_as_longlong(centre_dist_fast(GF, cf, px, py, pz));
                inexact |= cf.ok ? 0u : 1u << k;
            }
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
                const GridEntryDev e = P.entries.get<MULTI>((w0 >> 8) & 0xff);
                const TupleEval ev = eval_exact(*P.g.full, world(x, e.scale[0], e.offset[0]), world(y, e.scale[1], e.offset[1]), world(z, e.scale[2], e.offset[2]));
#pragma unroll
                for (int j = 0; j < FOLD_K; j++)
                    if (j == kk) key[j] = ev.key, dbits[j] = ev.dbits, alias[j] = ev.alias;
            }
        }
        ST(1);  // decode, cell, key, distance
        __syncthreads();  // the table is clean: the previous partition's winners have reset their slots
        ST(2);  // barrier: table clean
        // phase 1: cells and their minimum distance.  The thread's tuples probe TOGETHER: a round issues the compare-and-swap of
        // every tuple that has no slot yet and only then looks at the answers — one LDS round trip per round instead of one
        // per tuple and round (inserting them one after the other was a fifth of the kernel: three dependent chains).
        int slot[FOLD_K];
        {
            uint32_t ps[FOLD_K], pstep[FOLD_K];
            bool todo[FOLD_K];
            bool any = false;
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) {
                const uint64_t h = cell_hash(key[k], keys_wide_of(P.g));
                ps[k] = slot_of<NSLOT>(h);
                pstep[k] = ((uint32_t)(h >> 15) & (NSLOT - 1)) | 1u;  // (lds_insert_dense's sequence)
                todo[k] = (uint32_t)(k * NT) + threadIdx.x < cnt;
                slot[k] = -1;
                any |= todo[k];
            }
            while (any) {
                uint64_t prev[FOLD_K];
#pragma unroll
                for (int k = 0; k < FOLD_K; k++)
                    if (todo[k]) prev[k] = atomicCAS((unsigned long long *)&s_key[ps[k]], (unsigned long long)PCQ_EMPTY_KEY, (unsigned long long)key[k]);
                any = false;
#pragma unroll
                for (int k = 0; k < FOLD_K; k++) {
                    if (!todo[k]) continue;
                    if (prev[k] == PCQ_EMPTY_KEY || prev[k] == key[k]) {
                        slot[k] = (int)ps[k];
                        todo[k] = false;
                    } else {
                        ps[k] = (ps[k] + pstep[k]) & (NSLOT - 1);
                        any = true;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) {
                if (slot[k] < 0) continue;
                if (alias[k]) atomicOr(&s_aliasbits[slot[k] >> 5], 1u << (slot[k] & 31));
                atomicMin((unsigned long long *)&s_dist[slot[k]], (unsigned long long)dbits[k]);
            }
        }
        ST(3);  // phase 1: insert, minimum distance
        // (No count of the cells: a partition this kernel takes has at most CHUNK tuples, hence at most CHUNK <= LIMIT cells — it
        // cannot outgrow the table; longer partitions went to the defer list above.)
        static_assert(NT * FOLD_K <= LIMIT, "a chunk's tuples fit the table as cells");
        __syncthreads();  // every minimum is final
        ST(4);  // barrier: every minimum is final
        // phase 2: among the tuples at the minimum, the earliest in file order.  Two tuples of one cell at exactly the same
        // distance are rare (a point stored twice), so every tuple at its cell's minimum is taken for the winner and ranked at
        // once — ONE barrier for the order and the ranks —; a tuple that finds another one's order in its slot raises s_tie,
        // and only then the winners are told apart and ranked again.
        bool cand[FOLD_K];
        bool tie = false;
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) {
            cand[k] = slot[k] >= 0 && dbits[k] == s_dist[slot[k]];
            if (cand[k]) tie |= atomicMin((unsigned long long *)&s_ord[slot[k]], (unsigned long long)ord_of(tu[k])) != ~0ull;
        }
        if (tie) s_tie = 1;
        // Every occupied slot has exactly one tuple at (minimum distance, earliest order): its thread writes the cell.  The
        // winners leave wave by wave and, inside a wave, tuple slot by tuple slot (k), in lane order: the lanes of ONE store
        // instruction then write one contiguous run of keys (8 bytes each) and of records — with a per-thread order the same
        // instruction wrote every second or third record of a 4 KiB span, and the 8-byte key stores reached the memory side as
        // partial writes (counted: 6.7 GB written and 1.9 GB fetched beyond the tuples for 4.9 GB of winners).
        uint32_t cnt_k[FOLD_K], rank_k[FOLD_K], wave_total = 0;
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) {
            const unsigned long long m = __ballot(cand[k]);
            cnt_k[k] = (uint32_t)__popcll(m);
            rank_k[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            wave_total += cnt_k[k];
        }
        if (lane == 0) s_wsum[wave] = wave_total;
        __syncthreads();
        ST(5);  // phase 2, ranks + barrier
        if (s_tie) {  // (the same for the whole workgroup) equal distances in some cell: the earliest in file order only
            wave_total = 0;
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) {
                cand[k] = cand[k] && s_ord[slot[k]] == ord_of(tu[k]);
                const unsigned long long m = __ballot(cand[k]);
                cnt_k[k] = (uint32_t)__popcll(m);
                rank_k[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                wave_total += cnt_k[k];
            }
            __syncthreads();  // everybody has read s_tie and the first sums
            if (lane == 0) s_wsum[wave] = wave_total;
            if (threadIdx.x == 0) s_tie = 0;
            __syncthreads();
        }
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) {
            before += w < wave ? s_wsum[w] : 0;
            total += s_wsum[w];
        }
        ST(6);  // ranks + barrier
        if (ask_next) {  // the next partition's tuples have arrived (asked for three barriers ago) — BEFORE this one's stores are issued:
                         // loads and stores share one in-order counter, a wait behind the stores would wait for them too
#pragma unroll
            for (int k = 0; k < FOLD_K; k++) asm volatile("" ::"v"(rnxt[k].x), "v"(rnxt[k].y), "v"(rnxt[k].z), "v"(rnxt[k].w));
        }
        ST(7);  // the next partition's tuples have arrived
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
                st_record(P.wrecs, o, P.entries.get<MULTI>((w0 >> 8) & 0xff), x, y, z, w0, tu[k].w1, R_HAS);
            }
            s_key[sl] = PCQ_EMPTY_KEY, s_dist[sl] = ~0ull, s_ord[sl] = ~0ull;
        }
        if (threadIdx.x == 0) {
            P.wcount[p] = total;
            winners += total;
        }
        ST(8);  // the winners' stores, slots reset
        cur_lo = nxt_lo, cur_cnt = nxt_cnt, cur_out = nxt_out;
        if (pnn < nparts) range_take(&nxt_lo, &nxt_cnt, &nxt_out);
#pragma unroll
        for (int k = 0; k < FOLD_K; k++) rcur[k] = rnxt[k];
    }
    if (threadIdx.x == 0 && winners) atomicAdd(&P.stats[0], winners);
    ST_FLUSH(P.stats);
}

// the shapes grid_host.hip launches
template __global__ void k_fold<BIG_SLOTS, BIG_NT, BIG_K, BIG_LIMIT, true, false, 4>(FoldParams, uint32_t);
template __global__ void k_fold<SMALL_SLOTS, SMALL_NT, SMALL_K, SMALL_LIMIT, false, true, 3>(FoldParams, uint32_t);
template __global__ void k_fold_dense<SMALL_SLOTS, DENSE_NT, DENSE_K, SMALL_LIMIT, 4, false, false>(DenseParams, uint32_t);
template __global__ void k_fold_dense<SMALL_SLOTS, DENSE_NT, DENSE_K, SMALL_LIMIT, 4, true, true>(DenseParams, uint32_t);

}  // namespace pcqgrid
