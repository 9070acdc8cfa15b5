// lazer.cpp — the LAZER file searches: query/src/search/lazer.rs over
// readers/src/lazer_reader.rs.
//
// LAZER = LAS header | u64 block_size | u64 block_offsets[num_blocks] | blocks.  A block holds
// `number_of_attributes` u64 absolute file offsets followed by one LZ4 *frame* per attribute column
// (0 positions i32 x,y,z | 1 intensity | 2 return byte | 3 classification | ... | 8 colour u16 r,g,b),
// block_size points per block (lazer_reader.rs:58-127, 136-265).
//
// The split follows the rest of the host layer: locating and inflating the column blobs is host work
// (LZ4 is a serial byte format; the reference does it on the CPU too); the per-point part — world
// position rebuild, `bounds.contains`, the class compare, building the 31-byte records and feeding
// the collector (lazer.rs:60-76, 100-113) — runs on the GPU through libpcq.so on the inflated columns,
// which have exactly the LAST layout.  No per-point work happens here.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <sys/mman.h>
#include <cstring>
#include <memory>
#include <thread>

#include "lz4_frame.hpp"
#include "pcq_host.hpp"

namespace pcq {
namespace {

uint64_t rd64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
Status eof() { return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer"); }

// las::point::Format flags by format number
bool fmt_has_color(uint8_t f) { return f == 2 || f == 3 || f == 5 || f == 7 || f == 8 || f == 10; }
bool fmt_has_gps_time(uint8_t f) { return f == 1 || (f >= 3 && f <= 10); }
bool fmt_has_waveform(uint8_t f) { return f == 4 || f == 5 || f == 9 || f == 10; }
bool fmt_has_nir(uint8_t f) { return f == 8 || f == 10; }

struct Blob {
    const uint8_t *p = nullptr;
    size_t n = 0;
};
struct BlockBlobs {
    Blob positions, classifications, colors;
};

}  // namespace

// LAZERSource after LAZERSource::from (lazer_reader.rs:58-127).
struct LazerFile {
    MappedFile file;
    LasHeader header;
    uint64_t block_size = 0, num_blocks = 0;
    std::vector<uint64_t> block_offsets, block_byte_sizes;
    size_t number_of_attributes = 8;
    bool has_colors = false;

    // move_decoders_to_point_in_block(block, 0) (lazer_reader.rs:136-265): where the blobs of one block are.
    Status locate(size_t b, BlockBlobs *out) const {
        const uint64_t at = block_offsets[b], bytes = block_byte_sizes[b], fsz = file.size();
        const uint64_t table = (uint64_t)number_of_attributes * 8;
        if (at > fsz || fsz - at < table) return eof();  // :148-150 read_u64 per attribute
        std::vector<uint64_t> off(number_of_attributes);
        for (size_t k = 0; k < number_of_attributes; k++) off[k] = rd64(file.data() + at + 8 * k);
        if (bytes < table)  // :161 `block_size - number_of_attributes * 8` underflows
            return Status::Panic("attempt to subtract with overflow (LAZER block " + std::to_string(b) + " is smaller than its offset table)");
        const uint64_t blob_bytes = bytes - table;
        if (fsz - at - table < blob_bytes) return eof();  // :169-170 read_exact of the compressed attributes
        const uint8_t *cache = file.data() + at + table;  // current_block_cache
        // The reference builds the decoder slices with unchecked pointer arithmetic (:181-189 ...); offsets
        // that leave the block are undefined behaviour there and an error here.
        auto slice = [&](uint64_t from, uint64_t to, Blob *o) -> bool {
            if (from < off[0] || to < from) return false;
            const uint64_t a = from - off[0], e = std::min(to - off[0], blob_bytes);
            if (a > blob_bytes) return false;
            o->p = cache + a;
            o->n = (size_t)(e - a);
            return true;
        };
        bool ok = slice(off[0], off[1], &out->positions) && slice(off[3], off[4], &out->classifications);  // :176-177, :217-218
        if (ok && has_colors)  // :235-241 — without a tenth attribute the slice runs to the end of the block (and beyond)
            ok = slice(off[8], number_of_attributes > 9 ? off[9] : at + bytes, &out->colors);
        if (!ok) return Status::Err(PCQ_ERR_HEADER, "LAZER block " + std::to_string(b) + ": attribute offsets outside the block");
        return Status::Ok();
    }

    uint64_t points_in_block(size_t b) const { return std::min<uint64_t>(block_size, header.number_of_points - (uint64_t)b * block_size); }

    // read_into for one block's worth of points (lazer_reader.rs:590-716) minus the per-point arithmetic:
    // inflates the position / class / colour columns of block b into the given column slices.
    Status inflate(size_t b, uint64_t count, uint8_t *xyz, uint8_t *cls, uint8_t *rgb) const {
        BlockBlobs bl;
        Status st = locate(b, &bl);
        if (!st.ok()) return st;
        std::vector<uint8_t> scratch;  // only for dry runs (null destinations)
        auto blob = [&](const Blob &b, size_t bytes, size_t unit, uint8_t *dst) {
            return dst ? lz4_frame_decode_into(b.p, b.n, bytes, unit, dst) : lz4_frame_decode(b.p, b.n, bytes, unit, &scratch);
        };
        st = blob(bl.positions, (size_t)count * 12, 4, xyz);  // :598-600 read_i32 x3
        if (!st.ok()) return st;
        st = blob(bl.classifications, (size_t)count, 1, cls);  // :665 read_u8
        if (!st.ok()) return st;
        if (has_colors) st = blob(bl.colors, (size_t)count * 6, 2, rgb);  // :693-695 read_u16 x3
        if (!st.ok()) return st;
        return Status::Ok();
    }
};

// LAZERSource::from (lazer_reader.rs:58-127)
static Status lazer_open(const std::string &path, LazerFile *lz) {
    Status st = lz->file.open(path);
    if (!st.ok()) return st;
    st = parse_las_header(lz->file.data(), lz->file.size(), /*mask_format=*/false, &lz->header);  // :59-60
    if (!st.ok()) return st;
    const uint64_t fsz = lz->file.size(), otp = lz->header.offset_to_point_data, n = lz->header.number_of_points;
    if (otp > fsz || fsz - otp < 8) return eof();  // :66
    lz->block_size = rd64(lz->file.data() + otp);
    if (lz->block_size == 0) return Status::Panic("attempt to divide by zero (LAZER block size is 0)");  // :67
    lz->num_blocks = n / lz->block_size + (n % lz->block_size ? 1 : 0);
    if ((fsz - otp - 8) / 8 < lz->num_blocks) return eof();  // :70-72
    lz->block_offsets.resize(lz->num_blocks);
    for (uint64_t b = 0; b < lz->num_blocks; b++) lz->block_offsets[b] = rd64(lz->file.data() + otp + 8 + 8 * b);
    lz->block_byte_sizes.resize(lz->num_blocks);
    for (uint64_t b = 0; b < lz->num_blocks; b++) {  // :79-87
        const uint64_t end = b + 1 == lz->num_blocks ? fsz : lz->block_offsets[b + 1];
        if (end < lz->block_offsets[b]) return Status::Panic("attempt to subtract with overflow (LAZER block offsets are not ascending)");
        lz->block_byte_sizes[b] = end - lz->block_offsets[b];
    }
    const uint8_t f = lz->header.point_data_record_format;
    lz->has_colors = fmt_has_color(f);  // :93-105
    lz->number_of_attributes = 8 + (fmt_has_color(f) ? 1 : 0) + (fmt_has_gps_time(f) ? 1 : 0) + (fmt_has_waveform(f) ? 1 : 0) + (fmt_has_nir(f) ? 1 : 0);
    if (lz->num_blocks == 0)  // :123 -> :143 `self.block_offsets[0]` on an empty Vec
        return Status::Panic("index out of bounds: the len is 0 but the index is 0 (LAZER file without points)");
    BlockBlobs first;
    return lz->locate(0, &first);  // :123
}

Status lazer_file_bounds(const std::string &path, AABB *out) {  // main.rs:102-107 + :111
    LazerFile lz;
    Status st = lazer_open(path, &lz);
    if (!st.ok()) return st;
    *out = lz.header.bounds;
    return Status::Ok();
}

namespace {

// Host copies of inflated columns, in the LAST layout.
// Anonymous memory asking for transparent huge pages: a few hundred MB of inflate buffer are first touched by the
// inflating threads, and 4 KiB faults (plus the munmap at the end) were measured to cost more than the inflate.
struct HugeBuffer {
    uint8_t *p = nullptr;
    size_t bytes = 0;
    ~HugeBuffer() { reset(); }
    void reset(size_t n = 0) {
        if (p) {
            const auto t0 = std::chrono::steady_clock::now();
            munmap(p, bytes);
            if (getenv("PCQ_TIMING"))
                fprintf(stderr, "[pcq] released %.0f MB of inflate buffer in %.1f ms\n", (double)bytes / 1e6,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        p = nullptr;
        bytes = 0;
        if (!n) return;
        const size_t len = (n + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        (void)madvise(m, len, MADV_HUGEPAGE);
        p = (uint8_t *)m;
        bytes = len;
    }
    uint8_t *get() const { return p; }
};

struct Columns {
    HugeBuffer xyz, cls, rgb;
    uint64_t cap = 0;
    bool cap_colors = false;
    // grow-only: a searcher thread keeps its buffers from file to file, so only the first (largest) file pays
    // for the page faults of a few hundred MB and nothing is handed back to the kernel between files
    void alloc(uint64_t n, bool colors) {
        if (n <= cap && (!colors || cap_colors)) return;
        xyz.reset();
        cls.reset();
        rgb.reset();
        const uint64_t want = n > cap ? n : cap;
        xyz.reset(want * 12);
        cls.reset(want);
        cap_colors = colors || cap_colors;
        if (cap_colors) rgb.reset(want * 6);
        cap = want;
    }
};
Columns &thread_columns() {
    static thread_local Columns c;
    return c;
}

// Blocks are inflated by worker threads into a ring of block-sized slots (one contiguous
// arena, so that consecutive finished blocks form one run) and handed to `consume` IN BLOCK ORDER on the calling
// thread — the thread that owns the GPU context — while later blocks are still being inflated.  The ring bounds
// the memory (and with it the page faults and the munmap) to a few blocks instead of the whole file.
// consume(first_block, xyz, cls, rgb, points): the columns of a run of whole blocks starting at first_block.
using ConsumeFn = std::function<Status(size_t, const uint8_t *, const uint8_t *, const uint8_t *, uint64_t)>;

Status inflate_stream(const LazerFile &lz, pcq_ctx *ctx, const ConsumeFn &consume) {
    const size_t nb = (size_t)lz.num_blocks;
    const uint64_t n = lz.header.number_of_points;
    const uint64_t slot_points = std::min<uint64_t>(lz.block_size, n);
    const size_t nthreads = std::min<size_t>({nb, 16, std::max(1u, std::thread::hardware_concurrency())});
    size_t slots = std::min<size_t>(nb, nthreads + 4);
    const uint64_t budget = 1536ull << 20;  // bytes of ring: enough for 16 threads on 1 M-point blocks with room to spare
    if ((uint64_t)slots * slot_points * 19 > budget) slots = (size_t)std::max<uint64_t>(1, budget / (slot_points * 19));
    Columns &c = thread_columns();
    c.alloc((uint64_t)slots * slot_points, lz.has_colors);

    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::vector<char> done(nb, 0);
    std::vector<Status> results(nb);
    size_t next = 0, consumed = 0;
    bool abort = false;
    auto worker = [&]() {
        (void)pcq_bind_thread_near_device(ctx);  // the ring is first touched here and read by the staging copy next to the GPU
        for (;;) {
            size_t b;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return abort || next >= nb || next < consumed + slots; });
                if (abort || next >= nb) return;
                b = next++;
            }
            const uint64_t at = (uint64_t)(b % slots) * slot_points;
            Status st = lz.inflate(b, lz.points_in_block(b), c.xyz.get() + at * 12, c.cls.get() + at, lz.has_colors ? c.rgb.get() + at * 6 : nullptr);
            {
                std::lock_guard<std::mutex> lk(m);
                results[b] = std::move(st);
                done[b] = 1;
            }
            cv_done.notify_one();
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nthreads; t++) pool.emplace_back(worker);
    auto stop = [&]() {
        {
            std::lock_guard<std::mutex> lk(m);
            abort = true;
        }
        cv_work.notify_all();
        for (auto &t : pool) t.join();
    };
    Status out = Status::Ok();
    for (size_t b = 0; b < nb && out.ok();) {
        size_t e;
        {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return done[b] != 0; });
            if (!results[b].ok()) {
                out = results[b];
                break;
            }
            // the run of finished, intact blocks behind b that is contiguous in the ring
            e = b + 1;
            while (e < nb && e % slots != 0 && done[e] && results[e].ok()) e++;
        }
        uint64_t points = 0;
        for (size_t k = b; k < e; k++) points += lz.points_in_block(k);
        const uint64_t at = (uint64_t)(b % slots) * slot_points;
        out = consume(b, c.xyz.get() + at * 12, c.cls.get() + at, lz.has_colors ? c.rgb.get() + at * 6 : nullptr, points);
        {
            std::lock_guard<std::mutex> lk(m);
            consumed = e;
        }
        cv_work.notify_all();
        b = e;
    }
    stop();
    return out;
}

// A header may claim far more points than the file can hold (LZ4 expands at most 255x).  The reference
// would reserve the memory, start inflating and fail inside the first short blob; find that same error
// without the allocation.
Status impossible_point_count(const LazerFile &lz) {
    if (lz.header.number_of_points / 256 < lz.file.size() / 12 + 1) return Status::Ok();
    for (size_t b = 0; b < lz.num_blocks; b++) {
        Status st = lz.inflate(b, lz.points_in_block(b), nullptr, nullptr, nullptr);
        if (!st.ok()) return st;
    }
    return Status::Panic("capacity overflow");
}

}  // namespace

// ---- columns in HBM -----------------------------------------------------------------------------------------
// (A device-side LZ4 inflater — one wave per frame — was built and measured in round 1: 25 x slower than host
// threads on real columns, because the lz4 crate writes linked blocks and a frame is one sequential job
// (profiles/r01_lz4_device_rate.log, DESIGN_HISTORY.md D).  It is not part of the product.)
struct DeviceColumns {
    pcq_ctx *ctx = nullptr;
    void *xyz = nullptr, *cls = nullptr, *rgb = nullptr;
    ~DeviceColumns() {
        for (void *p : {xyz, cls, rgb})
            if (p) pcq_device_free(ctx, p);
    }
};

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void device_columns(const LazerFile &lz, const DeviceColumns &dc, pcq_columns *cols) {
    *cols = pcq_columns{};
    cols->xyz = dc.xyz;
    cols->xyz_stride = 12;
    cols->cls = dc.cls;
    cols->cls_stride = 1;
    cols->rgb = lz.has_colors ? dc.rgb : nullptr;  // no colour decoder: the record's colour stays 0 (:683-685)
    cols->rgb_stride = 6;
    for (int a = 0; a < 3; a++) cols->scale[a] = lz.header.scale[a], cols->offset[a] = lz.header.offset[a];  // :602-609
}

// ---- lazer.rs:34-78 -----------------------------------------------------------------------------------------
Status search_lazer_file_by_bounds(const std::string &path, const AABB &bounds, ResultCollector &rc) {
    LazerFile lz;
    Status st = lazer_open(path, &lz);  // :39-41
    if (!st.ok()) return st;
    if (!lz.header.bounds.intersects(bounds)) return Status::Ok();  // :47-53

    // :56-76 — chunk == block: every block is inflated, filtered with bounds.contains(world position)
    // and its matches collected in file order.  All blocks go through one scan.
    const uint64_t n = lz.header.number_of_points;
    st = impossible_point_count(lz);
    if (!st.ok()) return st;
    pcq_predicate pred{};
    pred.kind = PCQ_PRED_BOUNDS_F64;  // :69
    for (int a = 0; a < 3; a++) pred.wmin[a] = bounds.min[a], pred.wmax[a] = bounds.max[a];
    int r;
    {
        const double t0 = now_ms();
        double scan_ms = 0;
        const uint64_t first_index = rc.next_index;
        st = inflate_stream(lz, rc.context(), [&](size_t b, const uint8_t *xyz, const uint8_t *cls, const uint8_t *rgb, uint64_t points) -> Status {
            const double ts = now_ms();
            pcq_columns run{};
            run.xyz = xyz;
            run.xyz_stride = 12;
            run.cls = cls;
            run.cls_stride = 1;
            run.rgb = rgb;  // null without a colour decoder: the record's colour stays 0 (:683-685)
            run.rgb_stride = 6;
            run.n = points;
            run.first_index = first_index + (uint64_t)b * lz.block_size;
            for (int a = 0; a < 3; a++) run.scale[a] = lz.header.scale[a], run.offset[a] = lz.header.offset[a];  // :602-609
            // the slot is free again as soon as its bytes sit in the staging buffers; transfer and kernels of this run
            // overlap the staging copy of the next one
            const int rr = pcq_scan_host_nowait(rc.context(), &run, &pred, rc.handle());
            scan_ms += now_ms() - ts;
            return Status::FromLib(rr);
        });
        const int sr = pcq_ctx_synchronize(rc.context());
        if (!st.ok()) return st;
        r = sr;
        if (getenv("PCQ_TIMING"))
            fprintf(stderr, "[pcq] lazer: %zu blocks inflated and scanned in %.1f ms (scans from host columns: %.1f ms of it)\n",
                    (size_t)lz.num_blocks, now_ms() - t0, scan_ms);
    }
    rc.next_index += n;
    return Status::FromLib(r);
}

// ---- lazer.rs:80-116 ----------------------------------------------------------------------------------------
// The reference never clears its point buffer on this path (no `point_buffer.clear()` as at :75), so
// `get_attribute_range_ref(0..points_in_chunk)` and `get_point(idx)` always address the FIRST block's
// points: chunk k re-filters points [0, points_in_chunk(k)) of block 0.  Every block is still inflated
// (and can fail).  Reproduced as is: block 0's columns are scanned once per chunk.
Status search_lazer_file_by_classification(const std::string &path, uint8_t cls, ResultCollector &rc) {
    LazerFile lz;
    Status st = lazer_open(path, &lz);  // :85-87
    if (!st.ok()) return st;
    const uint64_t n0 = lz.points_in_block(0);
    st = impossible_point_count(lz);
    if (!st.ok()) return st;

    pcq_ctx *ctx = rc.context();
    DeviceColumns dc;
    {
        // every block is inflated by read_into (:101); only block 0 is ever looked at: it goes to the device as soon
        // as it is there, the rest is inflated for the errors it may raise
        dc.ctx = ctx;
        st = inflate_stream(lz, ctx, [&](size_t b, const uint8_t *xyz, const uint8_t *cls_col, const uint8_t *rgb, uint64_t) -> Status {
            if (b != 0) return Status::Ok();
            int r = pcq_device_alloc(ctx, n0 * 12, &dc.xyz);
            if (!r) r = pcq_device_alloc(ctx, n0, &dc.cls);
            if (!r && lz.has_colors) r = pcq_device_alloc(ctx, n0 * 6, &dc.rgb);
            if (!r) r = pcq_copy_to_device(ctx, dc.xyz, xyz, n0 * 12);
            if (!r) r = pcq_copy_to_device(ctx, dc.cls, cls_col, n0);
            if (!r && lz.has_colors) r = pcq_copy_to_device(ctx, dc.rgb, rgb, n0 * 6);
            return Status::FromLib(r);
        });
        if (!st.ok()) return st;
    }
    pcq_predicate pred{};
    pred.kind = PCQ_PRED_CLASS;
    pred.cls = cls;  // :107
    pcq_columns cols;
    device_columns(lz, dc, &cols);
    int r = 0;
    for (uint64_t k = 0; k < lz.num_blocks && !r; k++) {  // :98-113
        cols.n = lz.points_in_block((size_t)k);  // points_in_chunk
        cols.first_index = rc.next_index;
        r = pcq_scan_dev(ctx, &cols, &pred, rc.handle(), nullptr);
        rc.next_index += cols.n;
    }
    if (!r) r = pcq_ctx_synchronize(ctx);
    return Status::FromLib(r);
}

}  // namespace pcq
