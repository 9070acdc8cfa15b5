#include "grid_common.h"

using namespace pcqgrid;

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct GridRun {       // the output of one pass-0 launch
    uint8_t *tuples;   // ntiles blocks of P0_TILE tuples
    uint16_t *dir;     // ntiles directory rows
    uint32_t ntiles;
    uint32_t wide;     // 24-byte tuples (a colour column, or a scan whose 16-byte tuples had no room for the class byte)
    uint32_t entry;    // the entry (scale / offset / packing) the run was scanned with
    uint32_t block_bytes;  // from one tile's block to the next
};

struct GridState {
    // pending: pass-0 runs not folded yet
    std::vector<GridRun> runs;
    std::vector<GridEntryDev> entries;
    std::vector<void *> slabs;         // pool blocks holding the blocks and directory rows
    uint8_t *slab_cur = nullptr;
    size_t slab_left = 0;
    uint64_t pending_cap = 0;          // points scanned into the pending runs (= the most tuples they can hold)
    uint64_t pending_tiles = 0;
    bool any_wide = false;
    // folded winners, grouped by partition
    uint64_t *wkeys = nullptr;
    uint8_t *wrecs = nullptr;    // two arrays of 16-byte halves (RecArr), wrec_cap records each
    uint64_t wrec_cap = 0;
    uint64_t *wbase = nullptr;   // [P + 1]
    uint32_t *wcount = nullptr;  // [P]
    uint32_t f2 = 1;
    uint64_t wtotal = 0;
};

static void grid_free_pending(pcq_ctx *ctx, GridState *gs) {
    for (void *p : gs->slabs) pcq_pool_free(ctx, p);
    gs->slabs.clear();
    gs->slab_cur = nullptr;
    gs->slab_left = 0;
    gs->pending_cap = 0;
    gs->pending_tiles = 0;
    gs->any_wide = false;
    gs->runs.clear();
    gs->entries.clear();
}

static void grid_free_winners(pcq_ctx *ctx, GridState *gs) {
    pcq_pool_free(ctx, gs->wkeys);
    pcq_pool_free(ctx, gs->wrecs);
    pcq_pool_free(ctx, gs->wbase);
    pcq_pool_free(ctx, gs->wcount);
    gs->wkeys = nullptr, gs->wrecs = nullptr, gs->wbase = nullptr, gs->wcount = nullptr;
    gs->wrec_cap = 0;
    gs->wtotal = 0;
    gs->f2 = 1;
}

void pcq_grid_release(pcq_collector *c) {
    if (!c->gs) return;
    grid_free_pending(c->ctx, c->gs);
    grid_free_winners(c->ctx, c->gs);
    delete c->gs;
    c->gs = nullptr;
}

// a scratch list of pool blocks released together
struct Scratch {
    pcq_ctx *ctx;
    std::vector<void *> blocks;
    explicit Scratch(pcq_ctx *c) : ctx(c) {}
    ~Scratch() {
        for (void *p : blocks) pcq_pool_free(ctx, p);
    }
    template <typename T>
    int get(size_t count, T **out) {
        void *p = nullptr;
        const int rc = pcq_pool_alloc(ctx, count * sizeof(T), &p);
        if (rc) return rc;
        blocks.push_back(p);
        *out = (T *)p;
        return PCQ_OK;
    }
    void keep(void *p) { blocks.erase(std::remove(blocks.begin(), blocks.end(), p), blocks.end()); }
};
// Declared BEHIND a Scratch: whichever way the scope is left, the stream has drained before the Scratch hands its blocks
// back to the pool ("a block may be freed only when the work that used it has completed").
struct StreamDrainOnExit {
    hipStream_t s;
    explicit StreamDrainOnExit(hipStream_t stream) : s(stream) {}
    ~StreamDrainOnExit() { (void)hipStreamSynchronize(s); }
};

static int grid_fold(pcq_ctx *ctx, pcq_collector *c);

// `bytes` of the pending slabs, 256-byte aligned
static int grid_room(pcq_ctx *ctx, GridState *gs, size_t bytes, void **out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > gs->slab_left) {
        size_t slab = 256ull << 20;
        if (slab < bytes) slab = bytes;
        void *p = nullptr;
        const int rc = pcq_pool_alloc(ctx, slab, &p);
        if (rc) return rc;
        gs->slabs.push_back(p);
        gs->slab_cur = (uint8_t *)p;
        gs->slab_left = slab;
    }
    *out = gs->slab_cur;
    gs->slab_cur += bytes;
    gs->slab_left -= bytes;
    return PCQ_OK;
}

int pcq_grid_scan(pcq_ctx *ctx, pcq_collector *c, const DevCols &cols_in, const DevPred &pred, hipStream_t s) {
    if (cols_in.n == 0) return PCQ_OK;
    if (!c->gs) c->gs = new GridState();
    GridState *gs = c->gs;
    uint64_t budget = ctx->grid_pending_budget > 0 ? (uint64_t)ctx->grid_pending_budget : (384ull << 20);  // points: 7.7 - 9.2 GB of tuples
    if (budget > PENDING_MAX) budget = PENDING_MAX;  // (a fold's tuple counts and offsets are 32-bit)
    for (uint64_t first = 0; first < cols_in.n; first += RUN_POINTS) {
        DevCols cols = cols_in;
        cols.n = cols_in.n - first < RUN_POINTS ? cols_in.n - first : RUN_POINTS;
        cols.first_index = cols_in.first_index + first;
        cols.xyz = cols_in.xyz ? cols_in.xyz + first * cols_in.xyz_stride : nullptr;
        cols.cls = cols_in.cls ? cols_in.cls + first * cols_in.cls_stride : nullptr;
        cols.rgb = cols_in.rgb ? cols_in.rgb + first * cols_in.rgb_stride : nullptr;
        const uint32_t ntiles = (uint32_t)((cols.n + P0_TILE - 1) / P0_TILE);
        const bool wide = cols.rgb != nullptr;
        const size_t dir_room = (size_t)ntiles * DIR_STRIDE * sizeof(uint16_t);
        // How this scan's tuples are packed.  16 bytes need an axis on which x - lo < 2^24 for every match (the class byte rides
        // in that coordinate's top byte): the narrowest side of the integer query box; a class query stores no class at all.
        GridEntryDev e{};
        for (int a = 0; a < 3; a++) e.scale[a] = cols.scale[a], e.offset[a] = cols.offset[a], e.lo[a] = 0, e.cmask[a] = 0xffffffffu;
        e.fmt = FMT_NONE | FMT_NONE << 8 | FMT_NONE << 16;
        bool narrow = !wide && ctx->grid_tuple16 != 0;
        if (narrow && pred.kind == PCQ_PRED_CLASS) {
            e.cls_const = pred.cls & 0xffu;
        } else if (narrow && pred.kind == PCQ_PRED_BOUNDS) {
            int axis = 0;
            for (int a = 1; a < 3; a++)
                if (pred.width[a] < pred.width[axis]) axis = a;
            if (pred.empty) {
                // (nothing matches: any packing will do)
            } else if (pred.width[axis] < (1u << 24)) {
                for (int a = 0; a < 3; a++) e.lo[a] = pred.lo[a];
                e.cmask[axis] = 0x00ffffffu;
                uint32_t fmt = (uint32_t)(8 * axis) | FMT_NONE << 8 | FMT_NONE << 16;
                // all three sides below 2^24: the other two top bytes carry the second level's selector (pass 0 has the hash at hand)
                if (pred.width[0] < (1u << 24) && pred.width[1] < (1u << 24) && pred.width[2] < (1u << 24) && ctx->grid_tuple16 != 2) {
                    const int l = axis == 0 ? 1 : 0, h = axis == 2 ? 1 : 2;
                    e.cmask[l] = e.cmask[h] = 0x00ffffffu;
                    fmt = (uint32_t)(8 * axis) | (uint32_t)(8 * l) << 8 | (uint32_t)(8 * h) << 16;
                }
                e.fmt = fmt;
            } else {
                narrow = false;
            }
        } else {
            narrow = false;  // a world-space predicate (LAZER): the integer range of the matches is not known
        }
        const bool wide_t = !narrow;
        // (a block's room: its 5120 tuples, plus the padding option — 16-byte units — that moves the blocks' phase in memory)
        const uint32_t block_bytes = P0_TILE * tuple_bytes(wide_t) + 16u * (uint32_t)ctx->grid_block_pad;
        const size_t tuple_room = (size_t)ntiles * block_bytes + 64;
        // the entry: consecutive scans that agree in scale, offset and packing share it
        auto needs_entry = [&]() {
            if (gs->entries.empty()) return true;
            return memcmp(&gs->entries.back(), &e, sizeof e) != 0;
        };
        if ((needs_entry() && gs->entries.size() == 255) || gs->runs.size() == (size_t)MAX_RUNS || (gs->pending_cap && gs->pending_cap + cols.n > budget)) {
            c->last_stream = s;
            const int frc = grid_fold(ctx, c);
            if (frc) return frc;
        }
        GridRun run{};
        run.ntiles = ntiles, run.wide = wide_t, run.block_bytes = block_bytes;
        auto alloc_run = [&]() {
            void *pt = nullptr, *pd = nullptr;
            int arc = grid_room(ctx, gs, tuple_room, &pt);
            if (!arc) arc = grid_room(ctx, gs, dir_room, &pd);
            run.tuples = (uint8_t *)pt, run.dir = (uint16_t *)pd;
            return arc;
        };
        int rc = alloc_run();
        if (rc == PCQ_ERR_NOMEM && !gs->runs.empty()) {  // no room next to what is pending: fold that first (its slabs go back to the pool)
            c->last_stream = s;
            const int frc = grid_fold(ctx, c);
            if (frc) return frc;
            rc = alloc_run();
        }
        if (rc) return rc;
        if (needs_entry()) gs->entries.push_back(e);
        const uint32_t entry = (uint32_t)gs->entries.size() - 1;
        run.entry = entry;
        const uint32_t tile0 = (uint32_t)gs->pending_tiles;  // the run's first tile among all pending tiles: a tuple's place in the
                                                              // pending stream, (tile0 + tile) * 5120 + place in the tile, is its file order
        gs->pending_cap += cols.n;
        gs->pending_tiles += ntiles;
        gs->any_wide |= wide_t;
        P0Pack pk16{};
        for (int a = 0; a < 3; a++) pk16.lo[a] = e.lo[a], pk16.cmask[a] = e.cmask[a], pk16.top_shift[a] = fmt_top_shift(e.fmt, a);
        pk16.block_bytes = block_bytes;

        const DevGrid &g = c->grid;
        const unsigned nblocks = ntiles < (uint32_t)ctx->num_cus ? ntiles : (unsigned)ctx->num_cus;  // one workgroup per CU is resident (LDS)
        const int agg = ctx->grid_agg;
        // LAST blocks (12-byte positions at an aligned address, a class byte per point, 6-byte colours): the short index arithmetic
        const bool packed = cols.xyz_stride == 12 && ((uintptr_t)cols.xyz & 3) == 0 && (!cols.cls || cols.cls_stride == 1) && (!cols.rgb || cols.rgb_stride == 6);
#define PCQ_P0_LAUNCH(KIND, RGB, PACKED, WIDE) \
    hipLaunchKernelGGL((k_p0_part<KIND, RGB, PACKED, WIDE>), dim3(nblocks), dim3(P0_NT), 0, s, (P0Args{cols, pred, g, pk16, run.tuples, run.dir, ntiles, tile0, agg}))
#define PCQ_P0(KIND)                                                          \
    do {                                                                      \
        if (wide && packed) PCQ_P0_LAUNCH(KIND, true, true, true);            \
        else if (wide) PCQ_P0_LAUNCH(KIND, true, false, true);                \
        else if (wide_t && packed) PCQ_P0_LAUNCH(KIND, false, true, true);    \
        else if (wide_t) PCQ_P0_LAUNCH(KIND, false, false, true);             \
        else if (packed) PCQ_P0_LAUNCH(KIND, false, true, false);             \
        else PCQ_P0_LAUNCH(KIND, false, false, false);                        \
    } while (0)
        if (pred.kind == PCQ_PRED_BOUNDS) PCQ_P0(PCQ_PRED_BOUNDS);
        else if (pred.kind == PCQ_PRED_CLASS) PCQ_P0(PCQ_PRED_CLASS);
        else PCQ_P0(PCQ_PRED_BOUNDS_F64);
#undef PCQ_P0
#undef PCQ_P0_LAUNCH
        PCQ_HIP(hipGetLastError());
        gs->runs.push_back(run);
    }
    return PCQ_OK;
}

// Folds the pending runs (and the earlier winners) into a new set of winners.  Synchronises.
static int grid_fold(pcq_ctx *ctx, pcq_collector *c) {
    GridState *gs = c->gs;
    if (!gs || gs->runs.empty()) return PCQ_OK;
    hipStream_t s = ctx->stream;
    if (c->last_stream && c->last_stream != s) PCQ_HIP(hipStreamSynchronize(c->last_stream));
    Scratch tmp(ctx);
    StreamDrainOnExit drain_before_tmp(s);
    const int nruns = (int)gs->runs.size();
    const DevGrid &g = c->grid;
    if (gs->pending_cap >= (1ull << 32)) return pcq_fail(PCQ_ERR_UNSUPPORTED, "grid collector: more than 2^32 pending tuples in one fold");

    // run directory, entries, the per-bin fragment lists
    const uint32_t T = (uint32_t)gs->pending_tiles, Tp = (T + 63) & ~63u, Tp1 = (T + 1 + 63) & ~63u;
    std::vector<DevRun> hruns(nruns);
    {
        uint32_t tile0 = 0;
        for (int r = 0; r < nruns; r++) {
            hruns[r] = DevRun{gs->runs[r].tuples, gs->runs[r].dir, tile0, gs->runs[r].ntiles, gs->runs[r].wide, gs->runs[r].entry, gs->runs[r].block_bytes, 0};
            tile0 += gs->runs[r].ntiles;
        }
    }
    const bool any_wide = gs->any_wide;
    DevRun *d_runs = nullptr;
    GridEntryDev *d_entries = nullptr;
    uint32_t *d_bintot = nullptr, *d_binbase = nullptr, *d_preT = nullptr;
    uint16_t *d_startT = nullptr;
    uint64_t *d_tile_addr = nullptr;
    uint8_t *d_tile_entry = nullptr;
    unsigned long long *d_stats = nullptr;
    DevGrid *d_grid = nullptr;
    int rc = tmp.get(nruns, &d_runs);
    if (!rc) rc = tmp.get(256, &d_entries);
    if (!rc) rc = tmp.get(F1, &d_bintot);
    if (!rc) rc = tmp.get(F1 + 1, &d_binbase);
    if (!rc) rc = tmp.get(32, &d_stats);
    if (!rc) rc = tmp.get(1, &d_grid);
    if (!rc) rc = tmp.get((size_t)F1 * Tp, &d_startT);
    if (!rc) rc = tmp.get((size_t)F1 * Tp1, &d_preT);
    if (!rc) rc = tmp.get(T, &d_tile_addr);
    if (!rc) rc = tmp.get((size_t)T + 64, &d_tile_entry);
    if (rc) return rc;
    PCQ_HIP(hipMemcpyAsync(d_grid, &g, sizeof g, hipMemcpyHostToDevice, s));
    GridRef gref{};
    gref.full = d_grid;
    for (int a = 0; a < 3; a++) {
        gref.f.bmin[a] = g.bmin[a], gref.f.qk[a] = g.qk[a], gref.f.qmax[a] = g.qmax[a], gref.f.guard[a] = g.guard[a];
        gref.f.mask[a] = (uint32_t)g.mask[a], gref.f.shift[a] = g.shift[a];
    }
    gref.f.cell_size = g.cell_size;
    gref.f.keys_wide = g.keys_wide;
    PCQ_HIP(hipMemcpyAsync(d_runs, hruns.data(), nruns * sizeof(DevRun), hipMemcpyHostToDevice, s));
    PCQ_HIP(hipMemcpyAsync(d_entries, gs->entries.data(), gs->entries.size() * sizeof(GridEntryDev), hipMemcpyHostToDevice, s));
    EntryRef eref{};
    eref.table = d_entries;
    eref.tile_entry = d_tile_entry;
    eref.multi = gs->entries.size() > 1 ? 1u : 0u;
    eref.e0 = gs->entries[0];
    hipLaunchKernelGGL(k_dir_transpose, dim3((T + 63) / 64), dim3(BLOCK), 0, s, d_runs, nruns, T, Tp, Tp1, d_startT, d_preT, d_tile_addr, d_tile_entry);
    hipLaunchKernelGGL(k_bin_prefix, dim3(F1), dim3(1024), 0, s, d_preT, T, Tp1, d_bintot);
    hipLaunchKernelGGL(k_excl_scan_u32, dim3(1), dim3(1024), 0, s, d_bintot, d_binbase, (uint32_t)F1);
    PCQ_HIP(hipGetLastError());
    BinSrc src{d_preT, d_startT, d_tile_addr, T, Tp1, Tp};  // (replaced by the compacted bins below when the fragments are short)
    // How dense is the grid?  The distinct cells of two bins are counted into a global hash set — asked for here, before the
    // host knows how many tuples there are, so that ONE synchronisation brings back the tuple count and the estimate (the
    // set is sized for eight times the mean bin; its probing is bounded).  Not when the bins cannot be large anyway.
    const double old_per_bin = (double)gs->wtotal / F1;
    const bool probed = ctx->grid_f2 <= 0 && (double)gs->pending_cap / F1 + old_per_bin > BIG_DIRECT;
    PCQ_HIP(hipMemsetAsync(d_stats, 0, 256, s));
    if (probed) {
        uint64_t cap = 1024;
        while (cap < 16ull * (gs->pending_cap / F1 + 1) * PROBE_BINS) cap <<= 1;
        if (cap > (1ull << 26)) cap = 1ull << 26;
        uint64_t *d_set = nullptr;
        rc = tmp.get(cap, &d_set);
        if (rc) return rc;
        PCQ_HIP(hipMemsetAsync(d_set, 0xff, cap * 8, s));
        unsigned probe_blocks = (unsigned)(((uint64_t)T * PROBE_BINS + BLOCK - 1) / BLOCK);
        if (probe_blocks > 4096) probe_blocks = 4096;
        hipLaunchKernelGGL(k_probe_distinct, dim3(probe_blocks), dim3(BLOCK), 0, s, src, eref, g, d_set, cap - 1, d_stats);
        PCQ_HIP(hipGetLastError());
    }
    // the two numbers the host decides on come back through the context's pinned words, stored there by a kernel: a
    // device-to-host hipMemcpy into pageable memory is staged by the runtime (tens of microseconds each, two of them here)
    hipLaunchKernelGGL(k_fold_numbers, dim3(1), dim3(64), 0, s, d_binbase + F1, d_stats, ctx->h_scalars + 32);
    PCQ_HIP(hipGetLastError());
    PCQ_HIP(hipStreamSynchronize(s));  // also: the pageable sources above have been read
    const uint64_t m = ctx->h_scalars[32], w_old = gs->wtotal;
    const unsigned long long distinct = ctx->h_scalars[33];
    ctx->grid_last_tuples = (int64_t)m;
    if (m == 0) {
        grid_free_pending(ctx, gs);
        return PCQ_OK;
    }
    ctx->grid_folds++;
    // estimated cells per level-1 bin -> fold the bins directly, or cut them again first
    uint32_t f2 = 1;
    if (ctx->grid_f2 > 0) {
        f2 = (uint32_t)ctx->grid_f2;
    } else if (probed && (double)m / F1 + old_per_bin > BIG_DIRECT) {
        const double est = (double)distinct / PROBE_BINS + old_per_bin;
        if (est > BIG_DIRECT) {
            f2 = (uint32_t)std::ceil(est / SMALL_TARGET);
            if (f2 > F2_MAX) f2 = F2_MAX;
        }
    }

    // Fragments of less than two tuples on average (a scan whose tiles shed most of their tuples, a box that few points match):
    // the window readers would hold a handful of tuples per round, so the bins are copied together first — for the readers
    // that have a window.  The streaming fold of a coarse grid takes the fragments 64 at a time whatever they hold: it reads
    // the sparse run as it is (the copy was 0.25 ms of a 0.9 ms fold on the scan-ordered file).
    const bool stream_reads_bins = f2 == 1 && ctx->grid_stream != 0 && !(w_old && gs->f2 != 1);
    if (T > 2u * BIG_FB && m < 2ull * T * F1 && !stream_reads_bins) {
        const uint32_t Tc = F1, Tcp = F1, Tcp1 = (F1 + 1 + 63) & ~63u;
        uint8_t *d_comp = nullptr;
        uint32_t *d_cpre = nullptr;
        uint16_t *d_cstart = nullptr;
        uint64_t *d_caddr = nullptr;
        rc = tmp.get((size_t)m * tuple_bytes(any_wide) + 64, &d_comp);
        if (!rc) rc = tmp.get((size_t)F1 * Tcp1, &d_cpre);
        if (!rc) rc = tmp.get((size_t)F1 * Tcp, &d_cstart);
        if (!rc) rc = tmp.get(Tc, &d_caddr);
        if (rc) return rc;
        const uint64_t nfrag = (uint64_t)T * F1;
        hipLaunchKernelGGL(k_bin_compact, dim3((unsigned)((nfrag + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, src, eref, d_binbase, d_comp, any_wide ? 1u : 0u);
        hipLaunchKernelGGL(k_compact_dir, dim3(F1), dim3(BLOCK), 0, s, d_binbase, d_comp, any_wide ? 1u : 0u, Tcp1, Tcp, d_cpre, d_cstart, d_caddr);
        PCQ_HIP(hipGetLastError());
        src = BinSrc{d_cpre, d_cstart, d_caddr, Tc, Tcp1, Tcp};
        ctx->grid_compactions++;
    }

    bool level2_exact = false;
    for (int attempt = 0;; attempt++) {
        const uint32_t nparts = (uint32_t)F1 * f2;
        Scratch att(ctx);
        StreamDrainOnExit drain_before_att(s);
        bool staged_level2 = false;
        GridSeg seg2{};  // the second level's output (f2 > 1)
        const uint32_t *d_tot = d_bintot;
        const uint64_t *obase = gs->wbase;
        const uint32_t *ocount = gs->wcount;
        const uint64_t *okeys = gs->wkeys;
        RecArr orecs{gs->wrecs, gs->wrec_cap};
        const bool recut_old = w_old && gs->f2 != f2;
        if (f2 > 1 || recut_old) {
            Level2Params L{};
            L.src = src, L.entries = eref, L.g = g, L.f2 = f2, L.stats = d_stats, L.wide = any_wide;
            uint8_t *d_t2 = nullptr;
            uint32_t *d_off2 = nullptr, *d_cnt2 = nullptr;
            // one pass into regions with slack (k_level2), unless that failed for this fold or does not apply
            const uint64_t cap = (uint64_t)std::ceil((double)m / nparts * 1.3) + 64;
            const bool staged = f2 <= (uint32_t)L2_STAGED_F2 && !level2_exact && cap * nparts < (1ull << 32);
            if (f2 > 1) {
                rc = att.get((staged ? (size_t)(cap * nparts) : (size_t)m) * tuple_bytes(any_wide) + 64, &d_t2);
                if (!rc) rc = att.get((size_t)nparts + 1, &d_off2);
                if (!rc && staged) rc = att.get((size_t)nparts, &d_cnt2);
                if (rc) return rc;
                L.binbase = d_binbase, L.out = d_t2, L.off2 = d_off2, L.cnt2 = d_cnt2, L.cap = (uint32_t)cap;
            }
            uint64_t *d_okeys2 = nullptr, *d_obase2 = nullptr;
            uint8_t *d_orecs2 = nullptr;
            uint32_t *d_ooff2 = nullptr, *d_ocount2 = nullptr, *d_obin = nullptr, *d_obinbase = nullptr;
            if (recut_old) {
                rc = att.get(w_old, &d_okeys2);
                if (!rc) rc = att.get(w_old * 32, &d_orecs2);
                if (!rc) rc = att.get((size_t)nparts + 1, &d_ooff2);
                if (!rc) rc = att.get((size_t)nparts + 1, &d_obase2);
                if (!rc) rc = att.get((size_t)nparts + 1, &d_ocount2);
                if (!rc) rc = att.get(F1, &d_obin);
                if (!rc) rc = att.get(F1 + 1, &d_obinbase);
                if (rc) return rc;
                hipLaunchKernelGGL(k_old_per_bin, dim3(F1 / BLOCK), dim3(BLOCK), 0, s, gs->wcount, gs->f2, d_obin);
                hipLaunchKernelGGL(k_excl_scan_u32, dim3(1), dim3(1024), 0, s, d_obin, d_obinbase, (uint32_t)F1);
                L.okeys = gs->wkeys, L.orecs = RecArr{gs->wrecs, gs->wrec_cap}, L.obase = gs->wbase, L.ocount = gs->wcount, L.f2old = gs->f2;
                L.obinbase = d_obinbase, L.okeys2 = d_okeys2, L.orecs2 = RecArr{d_orecs2, w_old}, L.ooff2 = d_ooff2;
            }
            PCQ_HIP(hipMemsetAsync(d_stats, 0, 64, s));
            if (staged && (any_wide || eref.multi)) hipLaunchKernelGGL((k_level2<true, true, 1024>), dim3(F1), dim3(1024), 0, s, L);
            else if (staged) hipLaunchKernelGGL((k_level2<false, false, 1024>), dim3(F1), dim3(1024), 0, s, L);
            else hipLaunchKernelGGL(k_level2_direct, dim3(F1), dim3(L2_NT), 0, s, L);
            PCQ_HIP(hipGetLastError());
            if (f2 > 1) {
                seg2 = GridSeg{d_t2, d_off2, staged ? d_cnt2 : nullptr, any_wide ? 1u : 0u};
                uint32_t *d_tot2 = nullptr;
                rc = att.get(nparts, &d_tot2);
                if (rc) return rc;
                staged_level2 = staged;  // (whether a region was outgrown is read back with the fold's counters: the fold of truncated
                                         // partitions is wasted then, but the common case saves a synchronisation)
                hipLaunchKernelGGL(k_part_totals, dim3((nparts + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, seg2, nparts, d_tot2);
                d_tot = d_tot2;
            }
            if (recut_old) {
                hipLaunchKernelGGL(k_unpack_old_dir, dim3((nparts + 1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, d_ooff2, nparts, d_obase2, d_ocount2);
                okeys = d_okeys2, orecs = RecArr{d_orecs2, w_old}, obase = d_obase2, ocount = d_ocount2;
            }
        }
        // room for the winners
        const bool big = f2 == 1;
        const uint32_t limit = big ? BIG_LIMIT : SMALL_LIMIT;
        uint64_t wcap = m + w_old;
        if (wcap > (uint64_t)nparts * limit) wcap = (uint64_t)nparts * limit;
        if (wcap >= (1ull << 32)) return pcq_fail(PCQ_ERR_UNSUPPORTED, "grid collector: more than 2^32 cells in one fold");
        uint64_t *n_wkeys = nullptr, *n_wbase = nullptr, *d_room = nullptr, *d_pieces = nullptr, *d_piece_pre = nullptr;
        uint8_t *n_wrecs = nullptr;
        uint32_t *n_wcount = nullptr, *d_palias = nullptr, *d_pay = nullptr, *d_defer = nullptr;
        uint4 *d_surv = nullptr;
        // The streaming fold's survivor list per resident workgroup: twice the mean bin, at most 65536 records (3 MB) — for the
        // large bins of a whole file that is a fifth of the bin (beyond that the filter does not pay: the bin goes to
        // k_fold<BIG>), while a sparse run — tiles that folded their own duplicates hand on what IS, mostly, a running minimum —
        // keeps all of its tuples if it must (a quarter of the mean bin sent every bin of the scan-ordered file to the fallback).
        uint32_t surv_cap = (uint32_t)std::min<uint64_t>(65536, std::max<uint64_t>(4096, 2 * (m / F1)));
        const bool stream = ctx->grid_stream != 0;
        const bool dense = !big && !w_old;  // k_fold_dense first, k_fold for what it leaves
        const uint32_t npieces = (nparts + SCAN_PIECE - 1) / SCAN_PIECE;
        rc = att.get(wcap, &n_wkeys);
        if (!rc) rc = att.get(wcap * 32, &n_wrecs);
        if (!rc) rc = att.get((size_t)nparts + 1, &n_wbase);
        if (!rc) rc = att.get(nparts, &n_wcount);
        if (!rc) rc = att.get(nparts, &d_room);
        if (!rc) rc = att.get(nparts, &d_palias);
        if (!rc) rc = att.get(npieces, &d_pieces);
        if (!rc) rc = att.get((size_t)npieces + 1, &d_piece_pre);
        const uint32_t resident_wgs = (uint32_t)ctx->num_cus * (big ? 1u : 3u);  // what fits the LDS: the rest of the partitions is looped over
        if (!rc) rc = att.get((size_t)resident_wgs * (big ? BIG_SLOTS : SMALL_SLOTS) * 5, &d_pay);  // parked payloads, per resident workgroup
        if (!rc && (dense || (big && stream))) rc = att.get(nparts, &d_defer);
        if (!rc && big && stream) rc = att.get((size_t)resident_wgs * surv_cap * 3, &d_surv);  // (three 16-byte words per survivor: grid_fold_stream.hip)
        if (rc) return rc;
        hipLaunchKernelGGL(k_winner_room, dim3((nparts + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, d_tot, w_old ? ocount : nullptr, nparts, limit, d_room);
        hipLaunchKernelGGL(k_scan_piece_sums, dim3(npieces), dim3(1024), 0, s, d_room, nparts, d_pieces);
        hipLaunchKernelGGL(k_excl_scan_u64, dim3(1), dim3(1024), 0, s, d_pieces, d_piece_pre, npieces);
        hipLaunchKernelGGL(k_scan_pieces, dim3(npieces), dim3(1024), 0, s, d_room, nparts, d_piece_pre, n_wbase);
        PCQ_HIP(hipMemsetAsync(d_palias, 0, (size_t)nparts * 4, s));
        PCQ_HIP(hipMemsetAsync(d_stats, 0, 40, s));  // [0 .. 5): the fold's counters ([5]: the second level's, still to be read)
        if (!(f2 > 1 || recut_old)) PCQ_HIP(hipMemsetAsync(d_stats + 5, 0, 24, s));
        FoldParams F{};
        F.src = src, F.seg = seg2, F.entries = eref, F.g = gref;
        if (w_old) F.okeys = okeys, F.orecs = orecs, F.obase = obase, F.ocount = ocount;
        F.wkeys = n_wkeys, F.wrecs = RecArr{n_wrecs, wcap}, F.wbase = n_wbase, F.wcount = n_wcount, F.palias = d_palias, F.pay_scratch = d_pay, F.stats = d_stats;
        {
            uint32_t resident = resident_wgs;
            if (resident > nparts) resident = nparts;
            if (dense) {
                F.defer_list = d_defer;
                DenseParams D{};
                D.tuples = seg2.tuples, D.wide = seg2.wide, D.off = seg2.off, D.cnt = seg2.cnt, D.entries = eref, D.g = gref;
                D.wkeys = n_wkeys, D.wrecs = RecArr{n_wrecs, wcap}, D.wbase = n_wbase, D.wcount = n_wcount, D.palias = d_palias, D.stats = d_stats, D.defer_list = d_defer;
                // two workgroups per CU (4 waves per SIMD, 103 registers, nothing spilled) — three (6 waves per SIMD, 80 registers)
                // spilled 92 bytes per lane and partition, 4.8 GB of scratch each way per file: 2.58 against 2.20 ms in one process
                uint32_t dense_wgs = (uint32_t)ctx->num_cus * 2u;
                if (dense_wgs > nparts) dense_wgs = nparts;
                if (any_wide || eref.multi) hipLaunchKernelGGL((k_fold_dense<SMALL_SLOTS, DENSE_NT, DENSE_K, SMALL_LIMIT, 4, true, true>), dim3(dense_wgs), dim3(DENSE_NT), 0, s, D, nparts);
                else hipLaunchKernelGGL((k_fold_dense<SMALL_SLOTS, DENSE_NT, DENSE_K, SMALL_LIMIT, 4, false, false>), dim3(dense_wgs), dim3(DENSE_NT), 0, s, D, nparts);
            }
            if (big && stream) {  // the bins as streams; a bin with more running minima than its survivor list holds is left to k_fold<BIG>
                F.defer_list = d_defer;
                if (any_wide || eref.multi) hipLaunchKernelGGL((k_fold_stream<BIG_SLOTS, BIG_NT, BIG_LIMIT, 2, true, true>), dim3(resident), dim3(BIG_NT), 0, s, F, nparts, surv_cap, d_surv);
                else hipLaunchKernelGGL((k_fold_stream<BIG_SLOTS, BIG_NT, BIG_LIMIT, 2, false, false>), dim3(resident), dim3(BIG_NT), 0, s, F, nparts, surv_cap, d_surv);
            }
            if (big) hipLaunchKernelGGL((k_fold<BIG_SLOTS, BIG_NT, BIG_K, BIG_LIMIT, true, false, 4>), dim3(resident), dim3(BIG_NT), 0, s, F, nparts);
            else hipLaunchKernelGGL((k_fold<SMALL_SLOTS, SMALL_NT, SMALL_K, SMALL_LIMIT, false, true, 3>), dim3(resident), dim3(SMALL_NT), 0, s, F, nparts);
        }
        PCQ_HIP(hipGetLastError());
        unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        hipLaunchKernelGGL(k_words_out, dim3(1), dim3(64), 0, s, (const uint64_t *)d_stats, ctx->h_scalars + 40, 8u);
        PCQ_HIP(hipGetLastError());
        PCQ_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < 8; i++) st[i] = ctx->h_scalars[40 + i];
        if (staged_level2 && st[5]) {  // a sub-partition outgrew its region (cells with very many points): count first, then cut
            level2_exact = true;
            ctx->grid_level2_exact++;
            attempt--;
            continue;
        }
#ifdef PCQ_STAMPS
        {
            unsigned long long sx[16] = {0};
            PCQ_HIP(hipMemcpy(sx, d_stats + 16, sizeof sx, hipMemcpyDeviceToHost));
            if (sx[15]) {
                fprintf(stderr, "[pcq] stamps (cycles per wave, %llu waves):", sx[15]);
                for (int i = 0; i < 15; i++) fprintf(stderr, " [%d] %.0f", i, (double)sx[i] / (double)sx[15]);
                fprintf(stderr, "\n");
            }
            PCQ_HIP(hipMemsetAsync(d_stats + 16, 0, 128, s));
        }
#endif
        if (f2 > 1) ctx->grid_level2++;
        if (big && stream) ctx->grid_deferred += (int64_t)st[3];
        if (st[1]) {  // a partition held more cells than the LDS table: more partitions
            if (f2 >= F2_MAX || attempt > 8) return pcq_fail(PCQ_ERR_UNSUPPORTED, "grid collector: a partition does not fit the LDS table at the largest fan-out");
            ctx->grid_refolds++;
            f2 = big ? (uint32_t)((BIG_LIMIT * 3 / 2 + SMALL_TARGET - 1) / SMALL_TARGET) : (f2 * 2 > F2_MAX ? F2_MAX : f2 * 2);
            continue;
        }
        if (st[2]) {  // aliased keys: gather their tuples, sort by (key, file order), replay
            PCQ_HIP(hipMemsetAsync(d_stats + 4, 0, 8, s));
            if (big) hipLaunchKernelGGL((k_alias_gather<false, true>), dim3(nparts), dim3(L2_NT), 0, s, F, (AliasItem *)nullptr, d_stats + 4);
            else hipLaunchKernelGGL((k_alias_gather<false, false>), dim3(nparts), dim3(L2_NT), 0, s, F, (AliasItem *)nullptr, d_stats + 4);
            unsigned long long na = 0;
            PCQ_HIP(hipMemcpyAsync(&na, d_stats + 4, 8, hipMemcpyDeviceToHost, s));
            PCQ_HIP(hipStreamSynchronize(s));
            if (na) {
                AliasItem *d_list = nullptr, *d_sorted = nullptr;
                rc = att.get(na, &d_list);
                if (!rc) rc = att.get(na, &d_sorted);
                if (rc) return rc;
                PCQ_HIP(hipMemsetAsync(d_stats + 4, 0, 8, s));
                if (big) hipLaunchKernelGGL((k_alias_gather<true, true>), dim3(nparts), dim3(L2_NT), 0, s, F, d_list, d_stats + 4);
                else hipLaunchKernelGGL((k_alias_gather<true, false>), dim3(nparts), dim3(L2_NT), 0, s, F, d_list, d_stats + 4);
                if (na <= ALIAS_QUADRATIC) {
                    hipLaunchKernelGGL(k_alias_rank, dim3((unsigned)((na + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, d_list, (uint64_t)na, d_sorted);
                } else {  // massive aliasing: a real sort (alias_sort.hip)
                    rc = pcq_sort_by_key_then_order(ctx, d_list, sizeof(AliasItem), na, d_sorted, s);
                    if (rc) return rc;
                }
                hipLaunchKernelGGL(k_alias_replay, dim3((unsigned)((na + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, d_sorted, (uint64_t)na, F, f2);
                PCQ_HIP(hipGetLastError());
                PCQ_HIP(hipStreamSynchronize(s));
            }
        }
        // install
        att.keep(n_wkeys), att.keep(n_wrecs), att.keep(n_wbase), att.keep(n_wcount);
        grid_free_winners(ctx, gs);
        gs->wkeys = n_wkeys, gs->wrecs = n_wrecs, gs->wbase = n_wbase, gs->wcount = n_wcount;
        gs->wrec_cap = wcap;
        gs->f2 = f2;
        gs->wtotal = st[0];
        ctx->grid_last_f2 = f2;
        break;
    }
    grid_free_pending(ctx, gs);
    return PCQ_OK;
}

// Folds what is pending now (the host layer calls it when a file is done, so that a collector kept for later holds its
// winners — a few bytes per cell — instead of a tuple per scanned point).
int pcq_grid_flush(pcq_collector *c) { return c->gs ? grid_fold(c->ctx, c) : PCQ_OK; }

int pcq_grid_drain(pcq_collector *c, pcq_point *out, uint64_t *keys_out, uint64_t cap, uint64_t *out_n) {
    pcq_ctx *ctx = c->ctx;
    hipStream_t s = ctx->stream;
    *out_n = 0;
    GridState *gs = c->gs;
    if (!gs) return PCQ_OK;
    int rc = grid_fold(ctx, c);
    if (rc) return rc;
    const uint64_t n = gs->wtotal;
    *out_n = n;
    if ((!out && !keys_out) || n == 0) return PCQ_OK;
    if (cap < n) return pcq_fail(PCQ_ERR_CAPACITY, "grid collector holds %llu points, capacity %llu", (unsigned long long)n,
                                 (unsigned long long)cap);
    Scratch tmp(ctx);
    const uint32_t nparts = (uint32_t)F1 * gs->f2;
    uint32_t *d_pre = nullptr;
    uint8_t *d_out = nullptr;
    uint64_t *d_keys = nullptr;
    rc = tmp.get((size_t)nparts + 1, &d_pre);
    if (!rc && out) rc = tmp.get(n * 31 + 16, &d_out);
    if (!rc && keys_out) rc = tmp.get(n, &d_keys);
    if (rc) return rc;
    hipLaunchKernelGGL(k_excl_scan_u32, dim3(1), dim3(1024), 0, s, gs->wcount, d_pre, nparts);
    hipLaunchKernelGGL(k_drain, dim3(nparts), dim3(BLOCK), 0, s, gs->wkeys, RecArr{gs->wrecs, gs->wrec_cap}, gs->wbase, gs->wcount, d_pre, d_out, d_keys);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && out) e = hipMemcpyAsync(out, d_out, n * 31, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && keys_out) e = hipMemcpyAsync(keys_out, d_keys, n * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return pcq_fail(PCQ_ERR_HIP, "grid drain failed: %s", hipGetErrorString(e));
    return PCQ_OK;
}
