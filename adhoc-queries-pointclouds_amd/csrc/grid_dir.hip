#include "grid_common.h"

namespace pcqgrid {

// ---------------------------------------------------------------------------------------------------------------
// fold preparation: the directory rows, transposed into per-bin fragment lists
// ---------------------------------------------------------------------------------------------------------------

// 64 tiles per workgroup: their directory rows through LDS; lane = tile, so that what leaves are whole lines of
// startT[b] / preT[b] (preT gets the fragment's COUNT here; k_bin_prefix turns the counts into the prefix).
__global__ __launch_bounds__(BLOCK) void k_dir_transpose(const DevRun *__restrict__ runs, int nruns, uint32_t T, uint32_t Tp, uint32_t Tp1,
                                                         uint16_t *__restrict__ startT, uint32_t *__restrict__ preT, uint64_t *__restrict__ tile_addr,
                                                         uint8_t *__restrict__ tile_entry) {
    __shared__ uint32_t s_rows[64 * DIR_WORDS];  // 64 rows of 257 words: lane r reads word r * 257 + k — no two lanes in one bank
    __shared__ uint64_t s_rowptr[64];
    const uint32_t t0 = blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const uint32_t t = t0 + threadIdx.x;
        uint64_t rowptr = 0;
        if (t < T) {
            int lo = 0, hi = nruns;  // the last run with tile0 <= t
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (runs[mid].tile0 <= t) lo = mid;
                else hi = mid;
            }
            const DevRun r = runs[lo];
            const uint32_t local = t - r.tile0;
            rowptr = reinterpret_cast<uint64_t>(r.dir + (size_t)local * DIR_STRIDE);
            tile_addr[t] = reinterpret_cast<uint64_t>(r.tuples + (uint64_t)local * r.block_bytes) | (r.wide ? 1u : 0u);
            tile_entry[t] = (uint8_t)r.entry;
        }
        s_rowptr[threadIdx.x] = rowptr;
    }
    __syncthreads();
    for (int r = 0; r < 64; r++) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(s_rowptr[r]);
        for (int w = threadIdx.x; w < DIR_WORDS; w += BLOCK) s_rows[r * DIR_WORDS + w] = row ? ldg(row + w) : 0u;
    }
    __syncthreads();
    const uint32_t r = threadIdx.x & 63;
    if (t0 + r >= T) return;
    for (uint32_t b = threadIdx.x >> 6; b < (uint32_t)F1; b += WAVES) {
        const uint32_t w0 = s_rows[r * DIR_WORDS + (b >> 1)], w1 = s_rows[r * DIR_WORDS + ((b + 1) >> 1)];
        const uint32_t v0 = (b & 1) ? w0 >> 16 : w0 & 0xffffu, v1 = ((b + 1) & 1) ? w1 >> 16 : w1 & 0xffffu;
        startT[(size_t)b * Tp + t0 + r] = (uint16_t)v0;
        preT[(size_t)b * Tp1 + t0 + r] = v1 - v0;
    }
}

// preT[b][0 .. T): counts -> exclusive prefix, preT[b][T] = bintot[b] = the bin's tuples.  One workgroup per bin.
__global__ __launch_bounds__(1024) void k_bin_prefix(uint32_t *__restrict__ preT, uint32_t T, uint32_t Tp1, uint32_t *__restrict__ bintot) {
    __shared__ uint32_t s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *row = preT + (size_t)blockIdx.x * Tp1;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < T; c0 += 4096) {
        const uint32_t i0 = c0 + threadIdx.x * 4;
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = i0 + k < T ? row[i0 + k] : 0;
            sum += v[k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t run = carry + incl - sum, total = 0;
        for (int w = 0; w < 16; w++) {
            run += w < wave ? s_wave[w] : 0;
            total += s_wave[w];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < T) row[i0 + k] = run;
            run += v[k];
        }
        carry += total;
        __syncthreads();  // s_wave is rewritten
    }
    if (threadIdx.x == 0) {
        row[T] = carry;
        bintot[blockIdx.x] = carry;
    }
}

// Short fragments (a scan whose tiles shed most of their tuples, a box that few points match): the reader's window
// would hold a handful of tuples per round.  Then the bins are first copied together — one thread per fragment, comp =
// bin 0's tuples, bin 1's, ... — and described to the fold as pass-0 output of F1 "tiles", tile t = bin t's piece:
// fragment (b, t) is empty unless t == b.  Every consumer reads that through the same window code.
__global__ __launch_bounds__(BLOCK) void k_bin_compact(BinSrc S, EntryRef entries, const uint32_t *__restrict__ binbase, uint8_t *__restrict__ comp,
                                                       uint32_t wide_out) {
    const uint64_t q = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= (uint64_t)S.T * F1) return;
    const uint32_t bin = (uint32_t)(q / S.T), f = (uint32_t)(q % S.T);
    const uint32_t lo = ldg(S.preT + (size_t)bin * S.Tp1 + f), hi = ldg(S.preT + (size_t)bin * S.Tp1 + f + 1);
    if (lo == hi) return;
    const uint64_t a = frag_addr(S, bin, f);
    const bool wide = a & 1;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(a & ~1ull);
    uint8_t *dst = comp + (uint64_t)(ldg(binbase + bin) + lo) * tuple_bytes(wide_out);
    for (uint32_t i = 0; i < hi - lo; i++) st_raw_as(dst + (uint64_t)i * tuple_bytes(wide_out), ld_raw(src + (uint64_t)i * tuple_bytes(wide), wide), wide, wide_out, entries);
}
// the directory of the compacted bins: preT[b][t] = (t <= b ? 0 : the bin's tuples), startT = 0, tile_addr[t] = bin t's piece
__global__ __launch_bounds__(BLOCK) void k_compact_dir(const uint32_t *__restrict__ binbase, const uint8_t *__restrict__ comp, uint32_t wide, uint32_t Tp1,
                                                       uint32_t Tp, uint32_t *__restrict__ preT, uint16_t *__restrict__ startT, uint64_t *__restrict__ tile_addr) {
    const uint32_t b = blockIdx.x, cnt = binbase[b + 1] - binbase[b];
    for (uint32_t t = threadIdx.x; t <= (uint32_t)F1; t += BLOCK) {
        preT[(size_t)b * Tp1 + t] = t <= b ? 0u : cnt;
        if (t < (uint32_t)F1) startT[(size_t)b * Tp + t] = 0;
        if (b == 0 && t < (uint32_t)F1) tile_addr[t] = reinterpret_cast<uint64_t>(comp + (uint64_t)binbase[t] * tuple_bytes(wide)) | (wide ? 1u : 0u);
    }
}

// tot[p] = tuples of partition p of the second level's output.
__global__ __launch_bounds__(BLOCK) void k_part_totals(GridSeg sg, uint32_t nparts, uint32_t *__restrict__ tot) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= nparts) return;
    tot[p] = sg.cnt ? sg.cnt[p] : sg.off[p + 1] - sg.off[p];
}

// out[0..n] = exclusive prefix of in[0..n) (out[n] = the sum); one workgroup, any n.
__global__ __launch_bounds__(1024) void k_excl_scan_u32(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = (n + 1023) / 1024;  // a contiguous piece per thread
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += in[i];
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += s_wave[w];
    uint32_t run = wave_off + incl - sum;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = in[i];
        out[i] = run;
        run += v;
    }
    if (threadIdx.x == 1023) s_carry = wave_off + incl;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = s_carry;
}

__global__ __launch_bounds__(1024) void k_excl_scan_u64(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t n) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += in[i];
    uint64_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t up = __shfl_up((unsigned long long)incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += s_wave[w];
    uint64_t run = wave_off + incl - sum;
    for (uint32_t i = lo; i < hi; i++) {
        const uint64_t v = in[i];
        out[i] = run;
        run += v;
    }
    if (threadIdx.x == 1023) s_carry = wave_off + incl;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = s_carry;
}

// Room for the winners of partition p: never more than its inputs, never more than the LDS table holds.
__global__ __launch_bounds__(BLOCK) void k_winner_room(const uint32_t *__restrict__ tot, const uint32_t *__restrict__ ocount, uint32_t nparts,
                                                       uint32_t limit, uint64_t *__restrict__ room) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= nparts) return;
    const uint64_t in = (uint64_t)tot[p] + (ocount ? ocount[p] : 0);
    room[p] = in < limit ? in : limit;
}

// Exclusive prefix of up to a few hundred thousand u64 in three small launches (a single workgroup walking 200 k
// partitions took 0.2 ms): sums of 4096-element pieces, their prefix, the pieces again.
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
// out[i] = piece_prefix[piece] + exclusive prefix inside the piece; thread t owns 4 consecutive elements; out[n] = total
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

// Distinct cells among the tuples of the first PROBE_BINS level-1 bins (a global hash set; one thread per fragment).
__global__ __launch_bounds__(BLOCK) void k_probe_distinct(BinSrc S, EntryRef entries, DevGrid g, uint64_t *__restrict__ set, uint64_t mask,
                                                          unsigned long long *__restrict__ distinct) {
    uint32_t mine = 0;
    const uint32_t nfrag = S.T * PROBE_BINS;
    for (uint32_t q = blockIdx.x * BLOCK + threadIdx.x; q < nfrag; q += gridDim.x * BLOCK) {
        const uint32_t bin = q / S.T, f = q % S.T;
        const uint32_t lo = ldg(S.preT + (size_t)bin * S.Tp1 + f), hi = ldg(S.preT + (size_t)bin * S.Tp1 + f + 1);
        if (lo == hi) continue;
        const uint64_t a = frag_addr(S, bin, f);
        const bool wide = a & 1;
        const uint8_t *p = reinterpret_cast<const uint8_t *>(a & ~1ull);
        for (uint32_t i = 0; i < hi - lo; i++) {
            const GridTuple t = ld_tuple(p + (uint64_t)i * tuple_bytes(wide), wide, entries);
            const uint64_t key = eval_tuple(g, entries, t).key;
            uint64_t h = hash64(key) & mask;
            bool is_new = true;  // (the set is sized from an estimate: after 64 probes in a crowded one a key counts as new — an estimate either way)
            for (int probes = 0; probes < 64; probes++) {
                const uint64_t k = __hip_atomic_load(&set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k == key) {
                    is_new = false;
                    break;
                }
                if (k == PCQ_EMPTY_KEY) {
                    const uint64_t prev = atomicCAS((unsigned long long *)&set[h], (unsigned long long)PCQ_EMPTY_KEY, (unsigned long long)key);
                    if (prev == PCQ_EMPTY_KEY) break;
                    if (prev == key) {
                        is_new = false;
                        break;
                    }
                }
                h = (h + 1) & mask;
            }
            mine += is_new ? 1 : 0;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(distinct, (unsigned long long)mine);
}

// What the host decides on, into pinned host memory: the pending tuples (binbase[F1]) and the probe's distinct cells (stats[0]).
__global__ void k_fold_numbers(const uint32_t *__restrict__ tuples, const unsigned long long *__restrict__ stats, uint64_t *__restrict__ h_out) {
    if (threadIdx.x == 0) h_out[0] = *tuples;
    if (threadIdx.x == 1) h_out[1] = stats[0];
}
// n 64-bit words into pinned host memory
__global__ void k_words_out(const uint64_t *__restrict__ src, uint64_t *__restrict__ h_out, uint32_t n) {
    if (threadIdx.x < n) h_out[threadIdx.x] = src[threadIdx.x];
}

}  // namespace pcqgrid
