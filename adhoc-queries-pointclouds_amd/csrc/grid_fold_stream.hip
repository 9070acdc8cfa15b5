#include "grid_common.h"

namespace pcqgrid {

// ---------------------------------------------------------------------------------------------------------------
// the fold of a coarse grid's level-1 bins, as a stream (round 4)
// ---------------------------------------------------------------------------------------------------------------
// A coarse grid has few cells and many tuples per cell: of the ~320 k tuples of a bin of ca13 XL at 100 m all but a few
// per cent lose against a tuple the table has already seen.  k_fold<BIG> pays three workgroup barriers per chunk of 4096
// tuples for the exact (distance, file order) minimum and parks every provisional winner's payload in HBM; its waves wait
// 60 % of their cycles (profiles/r03_grid_sq_counters.txt).  This kernel splits the work the other way round:
//   1. STREAM, no barrier: every wave walks its own share of the bin's tuples.  A tuple finds its cell's slot, reads the
//      best distance seen so far and is OUT when its own is larger (the minimum only falls, a stale value is only too
//      large); otherwise it lowers the minimum (atomicMin on the f64 bits) and is appended to the workgroup's SURVIVOR list
//      in HBM scratch.  The tuple that ends up at the cell's minimum — and every tuple that ties with it — survives: when it
//      was tested the minimum was not below it.
//   2. EXACT, one pass over the survivors: those at their cell's final minimum atomicMin a 64-bit word
//      (order + 1) << 32 | list index — the earliest in file order wins and says where its record lies; nothing was parked.
//   3. The cells leave in slot order, a wave's 64 consecutive slots per round (one store instruction = one contiguous run),
//      from the survivor records, which are gathered before the first store is issued.
// A file in random order leaves a few survivors per cell (the running minima of a random sequence: ln n, a little more
// for what 1024 lanes see at the same time); a file sorted TOWARDS the cell centres makes every tuple a running minimum:
// a bin whose survivors outgrow the list goes on the defer list, and k_fold<BIG> folds it the old way.
// A wave takes the bin's fragments 64 at a time (batches handed out through a counter in LDS) — lane L holds fragment L's
// prefix and address — and turns "tuple g of the bin" into an address without a search: the fragments write their numbers
// at their first tuple's place in a per-wave map, and a prefix maximum over the lanes (six DPP steps) gives every tuple its
// fragment.  No window in LDS, nothing shared with the other waves.  The kernel is bound by its vector instructions
// (profiles/r04_grid_sq_counters.txt: 0.70 of the issue slots), so what it does per tuple is kept short: grid and entry
// come out of the argument segment where they are used (karg), the hash is two 32-bit multiplies, the slot one 24-bit one.
// ANYWIDE = false: every pending run has 16-byte tuples (one aligned load per tuple, four registers in flight); MULTI = false:
// one entry (EntryRef::get) — the common fold of one file; anything else takes the <true, true> form.
// A survivor is the tuple AS IT CAME (16 bytes: nothing is decoded for the few lanes of a wave that append one — the winner is,
// when its record is written) and {distance bits, slot, place}: the distance travels with it (recomputed in both exact
// passes, cell and distance of 8 M survivors were a seventh of the kernel).  ANYWIDE: a third word, the 24-byte tuple's
// last eight bytes and whether it is one.  (The list has room for three words per survivor either way.)
constexpr int SURV_WORDS = 3;

// The table of the streaming fold: key and best distance side by side (one LDS access brings both: a probe that finds its
// key has the cell's minimum with it), probed with DOUBLE hashing.  With linear probing the 64 lanes of a wave leave the
// loop together, after the longest cluster any of them ran into — at 3657 cells in 6400 slots that were 8.6 probe rounds
// per 64 tuples (counted: SQ_INSTS_LDS), a third of the kernel's vector instructions; a second hash gives every key its
// own sequence (the step is odd and no multiple of 5: coprime with 6400 = 2^8 x 5^2, so a sequence visits every slot).
struct KeyDist {
    uint64_t key, dist;
};
// the probe sequence's step: coprime with 6400 = 2^8 5^2, i.e. ending in 1, 3, 7 or 9 — ten times eight hash bits plus one of the four
__device__ __forceinline__ uint32_t stream_probe_step(uint64_t h) {
    const uint32_t b = (uint32_t)(h >> 38);
    return __umul24(b & 255u, 10u) + ((0x9731u >> ((b >> 6) & 12u)) & 15u);
}
template <int NSLOT, int LIMIT>
__device__ __forceinline__ int stream_find_or_insert(KeyDist *s_kd, uint64_t key, uint64_t h, uint32_t *s_ncell, uint64_t *seen) {
    static_assert(NSLOT == 6400, "the probe step below is chosen coprime with 6400");
    uint32_t s = slot_of<NSLOT>(h);
    const uint32_t step = stream_probe_step(h);
    for (int probes = 0; probes < NSLOT; probes++) {
        const uint64_t k = __hip_atomic_load(&s_kd[s].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t d = __hip_atomic_load(&s_kd[s].dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (a stale value is only too large)
        if (k == key) {
            *seen = d;
            return (int)s;
        }
        if (k == PCQ_EMPTY_KEY) {
            const uint64_t prev = atomicCAS((unsigned long long *)&s_kd[s].key, (unsigned long long)PCQ_EMPTY_KEY, (unsigned long long)key);
            if (prev == PCQ_EMPTY_KEY || prev == key) {
                *seen = ~0ull;  // (somebody may have been faster: too large is safe)
                if (prev == PCQ_EMPTY_KEY && atomicAdd(s_ncell, 1u) >= (uint32_t)LIMIT) return -1;
                return (int)s;
            }
        }
        s += step;
        if (s >= (uint32_t)NSLOT) s -= NSLOT;
    }
    return -1;
}

template <int NSLOT, int LIMIT>
__device__ __forceinline__ int stream_resolve(KeyDist *s_kd, uint64_t key, uint32_t s, uint32_t step, uint64_t k, uint64_t d, uint32_t *s_ncell, uint64_t *seen) {
    for (int probes = 0; probes < NSLOT; probes++) {
        if (k == key) {
            *seen = d;
            return (int)s;
        }
        if (k == PCQ_EMPTY_KEY) {
            const uint64_t prev = atomicCAS((unsigned long long *)&s_kd[s].key, (unsigned long long)PCQ_EMPTY_KEY, (unsigned long long)key);
            if (prev == PCQ_EMPTY_KEY || prev == key) {
                *seen = ~0ull;  // (somebody may have been faster: too large is safe)
                if (prev == PCQ_EMPTY_KEY && atomicAdd(s_ncell, 1u) >= (uint32_t)LIMIT) return -1;
                return (int)s;
            }
        }
        s += step;
        if (s >= (uint32_t)NSLOT) s -= NSLOT;
        k = __hip_atomic_load(&s_kd[s].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        d = __hip_atomic_load(&s_kd[s].dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return -1;
}

template <int NSLOT, int NT, int LIMIT, int U, bool ANYWIDE, bool MULTI>
__global__ __launch_bounds__(NT, 4) void k_fold_stream(FoldParams P, uint32_t nparts, uint32_t surv_cap, uint4 *__restrict__ surv_scratch) {
    constexpr int SPT = (NSLOT + NT - 1) / NT;  // slots per thread in the compaction
    constexpr int NW = NT / 64;
    __shared__ __attribute__((aligned(16))) KeyDist s_kd[NSLOT];  // key; f64 bits of the best squared distance (monotone for d >= 0)
    // the winner: file order (+ 1; 0 = an earlier fold's winner) << 32 | its place in the survivor list (an earlier fold's
    // winner: its index among the partition's old winners); ~0 = none yet.  ONE atomicMin finds the earliest survivor at
    // the minimum and says where its record lies.
    __shared__ uint64_t s_oi[NSLOT];
    __shared__ uint32_t s_aliasbits[(NSLOT + 31) / 32];
    __shared__ uint32_t s_ncell, s_over, s_nsurv, s_next_batch, s_anyalias, s_cnt2[SPT * NW];
    __shared__ uint32_t s_map[NW][64 * U];  // per wave: tag << 6 | fragment lane, at the fragment's first tuple's place in the chunk
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const BinSrc &S = P.src;
    constexpr int SW = ANYWIDE ? 3 : 2;
    uint4 *surv = surv_scratch + (size_t)blockIdx.x * surv_cap * SURV_WORDS;
    ST_DECL;
#pragma unroll
    for (int u = 0; u < U; u++) s_map[wave][u * 64 + lane] = 0;  // (tag 0: never)
    uint32_t tag = 0;                                              // chunks this wave has asked for (26 bits: 4 G tuples per wave)
    for (uint32_t it = blockIdx.x; it < nparts; it += gridDim.x) {
        const uint32_t p = xcd_order(it, nparts);
        // (what only the bin's end needs is read there, again: a value that lives across the stream is a register the stream
        // does not have — it goes to scratch and comes back behind the first stores, where waiting for it waits for them)
        const uint32_t n_old = P.okeys ? P.ocount[p] : 0;
        for (int t = threadIdx.x; t < NSLOT; t += NT) s_kd[t].key = PCQ_EMPTY_KEY, s_kd[t].dist = ~0ull, s_oi[t] = ~0ull;
        for (int t = threadIdx.x; t < (NSLOT + 31) / 32; t += NT) s_aliasbits[t] = 0;
        if (threadIdx.x == 0) s_ncell = 0, s_over = 0, s_nsurv = 0, s_next_batch = 0, s_anyalias = 0;
        __syncthreads();

        // earlier winners first: their distance is recomputed from the record (same f64 expressions, same bits)
        for (uint32_t i = threadIdx.x; i < n_old; i += NT) {
            const uint64_t old_base = P.obase[p];
            const uint64_t key = P.okeys[old_base + i];
            uint64_t unused;
            const int s = stream_find_or_insert<NSLOT, LIMIT>(s_kd, key, cell_hash(key, keys_wide_of(P.g)), &s_ncell, &unused);
            if (s < 0) {
                s_over = 1;
                continue;
            }
            const uint4 ra = *P.orecs.a(old_base + i), rb = *P.orecs.b(old_base + i);
            s_oi[s] = i;  // (order 0)
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
                s_kd[s].dist = (uint64_t)__double_as_longlong(centre_dist(gf, cell, ox, oy, oz));
            }
        }
        if (n_old) __syncthreads();
        ST(0);  // clear + earlier winners

        // ---- 1. the stream: this wave's tuples g_lo .. g_hi of the bin ----
        // Two stages, one chunk (U x 64 tuples) apart: ISSUE turns the chunk's tuple numbers into addresses and asks for the
        // tuples; PROCESS folds the chunk asked for a round earlier — so a wave always has a chunk in flight while it computes
        // (as one stage, its waves waited 58 % of their cycles: search chain, memory round trip and arithmetic one after the
        // other; profiles/r04_grid_progress.txt).  The issue side carries its own batch — fragments f0 .. f0 + 63, lane L
        // holds fragment f0 + L: the bin's tuples in front of it, and the address its tuple 0 WOULD have (fragment address -
        // prefix x tuple size, | 1 for 24-byte tuples: the address of tuple q is that + q x size) — and the next batch's,
        // asked for when the current one is entered.
        {
            const uint32_t *pre = S.preT + (size_t)p * S.Tp1;
            const uint32_t nbatches = (S.T + 63) / 64;
            {
                struct Batch {
                    uint32_t myp, vb_lo, vb_hi;  // per lane (until settled vb_lo / vb_hi hold the tile's address as loaded, st its start in the block)
                    uint32_t st;
                    uint32_t b_end_v;            // tuples of the bin in front of the next batch (the same in every lane)
                    uint32_t f0;                 // its first fragment (scalar); >= S.T: there is no such batch
                    uint32_t first, end;         // scalar copies of lane 0's prefix and of b_end_v — valid once SETTLED
                    bool settled;
                };
                // A batch's loads are asked for when the batch before it is entered; its two scalars are read out of the loaded
                // registers where waiting costs nothing — behind the hand-over of the chunk, which waits for the memory anyway.
                // (Read in the loop's header they made every round wait for EVERYTHING in flight: the compiler cannot count
                // loads across the loop's back edge.)
                auto settle = [&](Batch &B) {
                    if (!B.settled) {
                        B.first = uni32(B.myp), B.end = uni32(B.b_end_v);
                        const uint64_t ta = (uint64_t)B.vb_lo | ((uint64_t)B.vb_hi << 32);
                        const bool fv = B.f0 + lane < S.T;
                        const uint64_t fa = fv ? (ta & ~1ull) + (uint64_t)B.st * tuple_bytes(ta & 1) : 0ull;  // (frag_addr)
                        const uint64_t vb = (fa - (uint64_t)B.myp * tuple_bytes(ta & 1)) | (fv ? ta & 1 : 0ull);
                        B.vb_lo = (uint32_t)vb, B.vb_hi = (uint32_t)(vb >> 32);
                        B.settled = true;
                    }
                };
                // The bin's fragments in batches of 64, handed out through a counter in LDS: a wave takes the next batch when it
                // has used one up.  (Equal shares fixed in advance left the workgroup waiting a sixth of its time for its slowest
                // wave at the end of every bin: survivors, probe lengths and memory luck differ from wave to wave.)
                auto take_batch = [&]() {
                    uint32_t id = 0;
                    if (lane == 0) id = atomicAdd(&s_next_batch, 1u);
                    id = uni32(id);
                    Batch B;
                    B.f0 = id < nbatches ? id * 64u : S.T;
                    const uint32_t f = B.f0 + lane;
                    const bool fv = f < S.T;
                    B.myp = ldg(pre + (fv ? f : S.T));
                    B.b_end_v = ldg(pre + (B.f0 + 64 < S.T ? B.f0 + 64 : S.T));
                    B.first = B.end = 0, B.settled = false;
                    // (no load in a branch, and nothing computed on a loaded value here: this only ASKS — a lane behind the last
                    // fragment for the last one's entries —, settle() computes)
                    const uint32_t fc = fv ? f : S.T - 1;
                    const uint64_t ta = ldg(S.tile_addr + fc);
                    B.vb_lo = (uint32_t)ta, B.vb_hi = (uint32_t)(ta >> 32);
                    B.st = ldg(S.startT + (size_t)p * S.Tp + fc);
                    return B;
                };
                Batch cb = take_batch(), nb = take_batch();
                settle(cb), settle(nb);
                uint32_t g = cb.first;  // the next tuple to ask for
                uint32_t carry = 0;     // the lane (fragment of the batch) the last tuple asked for lies in
                RawTuple cur[U], nxt[U];
                uint32_t cur_wide = 0, nxt_wide = 0;  // bit u: tuple u of the chunk is 24 bytes
                uint32_t cur_n = 0, nxt_n = 0;        // tuples in the chunk
                ST(1);  // search of the first fragment, first batches
                for (;;) {  // (every condition below is the same for the whole wave)
                    nxt_n = 0;
                    while (cb.f0 < S.T && g >= cb.end) {  // the batch is used up (or empty): the next one — asked for when this one was entered — takes over
                        cb = nb;
                        settle(cb);  // (settled already, unless the batch before it was empty)
                        g = cb.first;
                        carry = 0;
                        nb = take_batch();
                    }
                    if (cb.f0 < S.T) {
                        const uint32_t stop = cb.end;
                        nxt_n = stop - g < (uint32_t)(64 * U) ? stop - g : (uint32_t)(64 * U);
                        // tuple number -> lane of the batch.  The fragments SAY where they start: lane f puts its number at its first
                        // tuple's place in the wave's map (a fragment with no tuple says nothing; a start in front of the chunk is the
                        // fragment the chunk before ended in: `carry`), and a tuple's fragment is the largest number at or in front
                        // of its place — one prefix maximum, six DPP steps in the vector pipe.  (A binary search per tuple over the
                        // 64 prefixes — six dependent ds_bpermute round trips and 36 instructions — was 45 of the 259 vector
                        // instructions the kernel spent per 64 tuples, and the kernel is bound by them: profiles/r04_grid_progress.txt.)
                        // The map is never cleared: an entry counts when it carries this chunk's tag.
                        tag++;
                        {
                            const uint32_t rel = cb.myp - g;  // (a start in front of g wraps around: out of range)
                            const uint32_t nextp = wave_next_lane(cb.myp, stop);
                            if (nextp != cb.myp && rel < (uint32_t)(64 * U)) s_map[wave][rel] = (tag << 6) | lane;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        uint32_t qq[U], lo[U];
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const uint32_t q = g + (uint32_t)u * 64 + lane;
                            qq[u] = q < stop ? q : stop - 1;  // (a place behind the batch's end holds no start: its lane re-reads the last tuple)
                            const uint32_t m = s_map[wave][u * 64 + lane];
                            lo[u] = wave_max_scan((m >> 6) == tag ? m & 63u : 0u);
                            lo[u] = lo[u] > carry ? lo[u] : carry;
                            carry = uni32((uint32_t)__builtin_amdgcn_readlane((int)lo[u], 63));
                        }
                        uint32_t al[U], ah[U];
#pragma unroll
                        for (int u = 0; u < U; u++) al[u] = (uint32_t)__shfl((int)cb.vb_lo, (int)lo[u], 64), ah[u] = (uint32_t)__shfl((int)cb.vb_hi, (int)lo[u], 64);
                        nxt_wide = 0;
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const uint64_t vb = (uint64_t)al[u] | ((uint64_t)ah[u] << 32);
                            const bool w = ANYWIDE && (vb & 1);
                            const uint8_t *src = reinterpret_cast<const uint8_t *>(vb & ~1ull) + (uint64_t)qq[u] * tuple_bytes(w);
                            if (ANYWIDE) {
                                nxt[u] = ld_raw(src, w);
                                nxt_wide |= w ? 1u << u : 0u;
                            } else {
                                nxt[u].a = *(const PCQ_GLOBAL u32x4_a16 *)src;
                            }
                        }
                        g += nxt_n;
                    }
                    ST(2);  // issue stage
                    if (cur_n) {
                        // The chunk's U tuples go through the stages TOGETHER — all evaluated, then all first probes asked for,
                        // then resolved, compared, and ONE append for the survivors of the whole chunk: the LDS round trips of
                        // the U tuples overlap instead of following one another (probe and append were a quarter of the
                        // kernel's cycles as four dependent chains per chunk: profiles/r04_grid_progress.txt, stamps).
                        uint64_t key[U], dbits[U];
                        bool alias[U], act[U];
                        // (grid and entry out of the argument segment, here: see karg())
                        const KArgPtr ka = karg_base();
                        const GridRef G = {karg<DevGridFast>(ka, offsetof(FoldParams, g) + offsetof(GridRef, f)), P.g.full};
                        const EntryRef E = karg<EntryRef>(ka, offsetof(FoldParams, entries));
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            act[u] = (uint32_t)u * 64 + lane < cur_n;
                            const GridTuple t = ANYWIDE ? decode_raw<MULTI>(cur[u], (cur_wide >> u) & 1, E) : decode16<MULTI>(cur[u].a, E);
                            const TupleEval ev = eval_tuple<MULTI>(G, E, t);
                            key[u] = ev.key, dbits[u] = ev.dbits, alias[u] = ev.alias;
                        }
                        ST(4);  // decode (+ whatever the chunk's tuples still took to arrive), cell, key, distance
                        uint32_t ps[U], pstep[U];
                        uint64_t k0[U], d0[U];
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const uint64_t h = cell_hash(key[u], keys_wide_of(G));
                            ps[u] = slot_of<NSLOT>(h);
                            pstep[u] = stream_probe_step(h);
                            k0[u] = __hip_atomic_load(&s_kd[ps[u]].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            d0[u] = __hip_atomic_load(&s_kd[ps[u]].dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (a stale value is only too large)
                        }
                        int sl[U];
                        bool surv_me[U];
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            sl[u] = -1;
                            surv_me[u] = false;
                            if (!act[u]) continue;
                            uint64_t seen;
                            sl[u] = stream_resolve<NSLOT, LIMIT>(s_kd, key[u], ps[u], pstep[u], k0[u], d0[u], &s_ncell, &seen);
                            if (sl[u] < 0) {
                                s_over = 1;
                                continue;
                            }
                            if (alias[u]) atomicOr(&s_aliasbits[sl[u] >> 5], 1u << (sl[u] & 31));
                            if (dbits[u] <= seen) {
                                if (dbits[u] < seen) {
                                    const uint64_t old = atomicMin((unsigned long long *)&s_kd[sl[u]].dist, (unsigned long long)dbits[u]);
                                    if (dbits[u] < old) s_oi[sl[u]] = ~0ull;  // a new minimum: an earlier fold's winner is out (racing writers store the same value)
                                }
                                surv_me[u] = true;
                            }
                        }
                        ST(5);  // hash, probe, compare, lower the minimum
                        unsigned long long m[U];
                        uint32_t nsv = 0;
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            m[u] = __ballot(surv_me[u]);
                            nsv += (uint32_t)__popcll(m[u]);
                        }
                        if (nsv) {  // (the same for the whole wave)
                            uint32_t base = 0;
                            if (lane == 0) base = atomicAdd(&s_nsurv, nsv);
                            base = uni32(base);
#pragma unroll
                            for (int u = 0; u < U; u++) {
                                const uint32_t pos = base + (uint32_t)__popcll(m[u] & ((1ull << lane) - 1ull));
                                base += (uint32_t)__popcll(m[u]);
                                if (surv_me[u] && pos < surv_cap) {
                                    surv[SW * (size_t)pos] = make_uint4(cur[u].a.x, cur[u].a.y, cur[u].a.z, cur[u].a.w);
                                    surv[SW * (size_t)pos + 1] = make_uint4((uint32_t)dbits[u], (uint32_t)(dbits[u] >> 32), (uint32_t)sl[u], cur[u].a.w);  // (all the exact pass reads)
                                    if (ANYWIDE) surv[SW * (size_t)pos + 2] = make_uint4(cur[u].b.x, cur[u].b.y, (cur_wide >> u) & 1, 0u);
                                }
                            }
                        }
                        ST(7);  // survivor append
                    }
                    if (!nxt_n) break;
#pragma unroll
                    for (int u = 0; u < U; u++) cur[u] = nxt[u];
                    cur_wide = nxt_wide, cur_n = nxt_n;
#ifdef PCQ_STAMPS
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    settle(nb);
                    ST(8);  // hand-over: the next chunk's tuples have arrived
                }
            }
        }
        // The survivor records are read back by other threads of THIS workgroup: a workgroup-scope fence (the stores have
        // left the wave; all waves of a workgroup share one L1), as k_fold does for its parked payloads.
        __threadfence_block();
        __syncthreads();
        ST(9);  // waiting for the workgroup's other waves
        const uint32_t nsurv = s_nsurv;
        uint32_t report = ~0u;  // the bin's cells, for thread 0 to write down behind the last barrier
        if (s_over) {  // more cells than the table holds: the host repeats the fold with more partitions
            if (threadIdx.x == 0) {
                P.wcount[p] = 0;
                atomicAdd(&P.stats[1], 1ull);
            }
        } else if (nsurv > surv_cap) {  // more running minima than the list holds (a file sorted towards the cell centres): k_fold<BIG> takes the bin
            if (threadIdx.x == 0) {
                P.wcount[p] = 0;
                P.defer_list[atomicAdd(&P.stats[3], 1ull)] = p;
            }
        } else {
            // ---- 2. exact: among the survivors at their cell's minimum distance, the earliest in file order ----
            for (uint32_t i0 = threadIdx.x; i0 < nsurv; i0 += NT * 4) {  // (four records asked for together: the list comes from the L2)
                uint4 rc[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t i = i0 + q * NT;
                    rc[q] = surv[SW * (size_t)(i < nsurv ? i : nsurv - 1) + 1];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t i = i0 + q * NT;
                    if (i < nsurv && ((uint64_t)rc[q].x | ((uint64_t)rc[q].y << 32)) == s_kd[rc[q].z].dist)
                        atomicMin((unsigned long long *)&s_oi[rc[q].z], (unsigned long long)(((uint64_t)rc[q].w + 1) << 32 | i));  // (ord_of: the place + 1)
                }
            }
            __syncthreads();
            ST(3);  // exact pass over the survivors
            // ---- 3. the cells leave in slot order, a wave's 64 CONSECUTIVE slots per round ----
            // Round j: thread t looks at slot j x NT + t; the cells of a wave's 64 slots get consecutive places, so ONE store
            // instruction writes one contiguous run of keys and of records.  (With seven consecutive slots per thread the lanes
            // of a store were four records apart: 64 partial writes per instruction, and the memory side took 26 us per bin to
            // accept them — stamps in profiles/r04_grid_progress.txt; the table reads are free of bank conflicts this way, too.)
            // Places: the cells of (round, wave) counted by a ballot, the SPT x NW counts summed in (round, wave) order by
            // every wave for itself (two per lane, a scan across the lanes).
            // (opaque: what depends on the thread's number is computed HERE.  The addresses of the thread's seven slots in the three
            // tables do not change from bin to bin; computed once in front of the kernel's loop they were twenty registers that
            // lived through the stream, i.e. in scratch, and came back between the stores below, each time waiting for them)
            int tid = (int)threadIdx.x;
            asm volatile("" : "+v"(tid));
            const uint32_t n_old = P.okeys ? P.ocount[p] : 0;
            const uint64_t old_base = P.okeys ? P.obase[p] : 0;
            const uint64_t out_base = P.wbase[p];
            static_assert(SPT * NW <= 128, "two counts per lane");
            unsigned long long occ[SPT];
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                const int s = j * NT + tid;
                occ[j] = __ballot(s < NSLOT && s_kd[s < NSLOT ? s : 0].key != PCQ_EMPTY_KEY);
                if (lane == 0) s_cnt2[j * NW + wave] = (uint32_t)__popcll(occ[j]);
            }
            __syncthreads();
            uint32_t base_j[SPT], total;
            {
                const uint32_t c0 = lane < (uint32_t)(SPT * NW) ? s_cnt2[lane] : 0u, c1 = 64 + lane < (uint32_t)(SPT * NW) ? s_cnt2[64 + lane] : 0u;
                uint32_t i0 = c0, i1 = c1;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t u0 = __shfl_up(i0, off, 64), u1 = __shfl_up(i1, off, 64);
                    if (lane >= (uint32_t)off) i0 += u0, i1 += u1;
                }
                const uint32_t t0 = __shfl(i0, 63, 64);
                total = t0 + __shfl(i1, 63, 64);
                const uint32_t e0 = i0 - c0, e1 = t0 + i1 - c1;  // the cells in front of entry `lane` and of entry 64 + `lane`
#pragma unroll
                for (int j = 0; j < SPT; j++) {
                    const int e = j * NW + (int)wave;
                    base_j[j] = uni32(e < 64 ? __shfl(e0, e, 64) : __shfl(e1, e - 64, 64));  // (the same in every lane: a scalar register)
                }
            }
            ST(6);  // places of the cells
            // The winners' records are asked for TOGETHER, before anything waits for one of them (the list lies in HBM, a round
            // trip takes ~10 us while the rest of the chip streams: one after the other they were a sixth of the kernel), and
            // written as stores only: a load behind a store waits for the store as well (loads and stores share one in-order
            // counter, and the compiler cannot count stores that sit in branches) — so no load follows the first store.  Key and
            // winner word of a slot are read from the table again when the slot is written: next to seven records they did
            // not fit the registers, and what goes to scratch comes back with a memory round trip of its own.  Cells with
            // aliased keys and cells whose earlier winner stays — both read memory — are left to a second pass that only runs
            // where there is such a cell.  (A slot without a survivor record reads record 0; nothing is done with it.)
            bool any_alias = false, later = false;
            uint4 wra[SPT], wrb[SPT];
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                const bool in = j * NT + tid < NSLOT;
                const int s = in ? j * NT + tid : 0;
                const uint64_t oi = s_oi[s];
                const bool simple = !((s_aliasbits[s >> 5] >> (s & 31)) & 1) && (oi >> 32) != 0;
                const size_t wi = in && simple && oi != ~0ull ? (uint32_t)oi : 0u;
                wra[j] = surv[SW * wi];
                wrb[j] = ANYWIDE ? surv[SW * wi + 2] : make_uint4(0u, 0u, 0u, 0u);
            }
#ifdef PCQ_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ST(12);  // the winners' records have arrived
#endif
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                const int s = j * NT + tid;
                const uint64_t key = s < NSLOT ? s_kd[s].key : PCQ_EMPTY_KEY;
                if (key == PCQ_EMPTY_KEY) continue;
                const uint64_t o = out_base + base_j[j] + (uint32_t)__popcll(occ[j] & ((1ull << lane) - 1ull));
                P.wkeys[o] = key;
                if (((s_aliasbits[s >> 5] >> (s & 31)) & 1) || (s_oi[s] >> 32) == 0) {
                    later = true;
                } else {
                    RawTuple r;
                    r.a = (u32x4_a16){wra[j].x, wra[j].y, wra[j].z, wra[j].w};
                    r.b = (u32x2_a8){wrb[j].x, wrb[j].y};
                    const GridTuple t = ANYWIDE ? decode_raw<MULTI>(r, wrb[j].z != 0, P.entries) : decode16<MULTI>(r.a, P.entries);
                    st_record(P.wrecs, o, P.entries.get<MULTI>((t.w0 >> 8) & 0xff), t.x, t.y, t.z, t.w0, t.w1, R_HAS);
                }
            }
            if (later) {
#pragma unroll 1
                for (int j = 0; j < SPT; j++) {
                    const int s = j * NT + tid;
                    if (s >= NSLOT) continue;
                    const uint64_t key = s_kd[s].key;
                    if (key == PCQ_EMPTY_KEY) continue;
                    uint32_t bj = base_j[0];  // (base_j[j] without a register array indexed at run time)
#pragma unroll
                    for (int jj = 1; jj < SPT; jj++) bj = jj == j ? base_j[jj] : bj;
                    unsigned long long oc = occ[0];
#pragma unroll
                    for (int jj = 1; jj < SPT; jj++) oc = jj == j ? occ[jj] : oc;
                    const uint64_t o = out_base + bj + (uint32_t)__popcll(oc & ((1ull << lane) - 1ull));
                    if ((s_aliasbits[s >> 5] >> (s & 31)) & 1) {  // left to the exact replay: the state before this fold (the earlier winner, if there is one) + the flag
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
                    } else if ((s_oi[s] >> 32) == 0) {  // the earlier winner stays
                        const uint64_t at = old_base + (uint32_t)s_oi[s];
                        *P.wrecs.a(o) = *P.orecs.a(at);
                        *P.wrecs.b(o) = *P.orecs.b(at);
                    }
                }
            }
            ST(11);  // the cells' keys and records written (issued)
#ifdef PCQ_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ST(13);  // ... and taken by the memory side
#endif
            if (any_alias) s_anyalias = 1;  // (read by thread 0 behind the barrier below, cleared by it at the next bin's start)
            report = total;
        }
        __syncthreads();  // the table is cleared for the next partition
        if (threadIdx.x == 0 && report != ~0u) {
            if (s_anyalias) {
                P.palias[p] = 1;
                atomicAdd(&P.stats[2], 1ull);
            }
            P.wcount[p] = report;
            if (report) atomicAdd(&P.stats[0], (unsigned long long)report);
        }
        ST(10);  // exact phase + output
    }
    ST_FLUSH(P.stats);
}

template __global__ void k_fold_stream<BIG_SLOTS, BIG_NT, BIG_LIMIT, 2, false, false>(FoldParams, uint32_t, uint32_t, uint4 *);
template __global__ void k_fold_stream<BIG_SLOTS, BIG_NT, BIG_LIMIT, 2, true, true>(FoldParams, uint32_t, uint32_t, uint4 *);

}  // namespace pcqgrid
