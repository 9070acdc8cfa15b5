// lz4_inflate.hip — LZ4 block inflate on the device for LAZER column blobs.
//
// A LAZER file holds thousands of independent LZ4 frames (one per attribute column per block,
// readers/src/lazer_reader.rs:136-265).  Inside a frame the blocks are usually *linked* (the lz4 crate's
// default): a match may reach back into the previous block, so a frame is a sequential job — but the
// frames are independent.  One wave inflates one frame:
//
//   * the 64 KiB LZ4 window lives in LDS as a ring (a wave's LDS accesses are ordered, so match copies
//     read bytes written a few instructions earlier without any global-memory coherence question), and
//     the compressed bytes the sequence parser looks at are staged 4 KiB at a time in LDS as well — a
//     token, its length bytes and its offset cost LDS reads, not dependent global loads;
//   * long literal runs and stored blocks (incompressible columns: noisy i32 positions) bypass the staging:
//     16 bytes per lane, four loads in flight, aligned 16-byte stores to the destination and the ring;
//   * short literals and matches are wave-wide byte copies (64 bytes per step); overlapping matches use the
//     periodic form src = out - offset + (i mod offset), in chunks that cannot wrap the ring onto their own source;
//   * everything the wave reads or writes is bounds-checked, and every loop consumes input, so a damaged
//     frame ends the wave instead of faulting or spinning.
//
// The kernel accepts only what liblz4 accepts (the same end-of-block rules as host/lz4_frame.cpp) and
// reports anything else — damage, block checksums, a frame that ends early or at an odd place — as
// "not handled": the host layer then runs its own reader on that frame, which reproduces the reference's
// exact error behaviour.  The device path is a fast path for well-formed frames, never a second opinion.
#include "pcq_internal.h"

namespace {

constexpr uint32_t RING = 65536;  // the LZ4 window
constexpr uint32_t WIN = 4096;    // staged compressed input

struct DevLz4Job {
    const uint8_t *src;  // first byte of the block sequence (behind the frame header)
    uint64_t n;          // bytes from src to the end of the blob
    uint8_t *dst;
    uint64_t need;
    uint32_t max_block;
    uint32_t independent;
    uint32_t has_size;
    uint32_t _pad;
    uint64_t content_size;
    int32_t status;      // out: 0 handled, 1 not handled
    uint32_t _pad2;
};

// 16 bytes at any address: the compiler picks the widest access the target's unaligned mode allows
struct __attribute__((packed, aligned(1))) U16B {
    uint32_t w[4];
};
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// One wave's view of a job.  `ring` holds the last 64 KiB of output at index (position + phase) mod 64 Ki,
// where phase = dst & 15, so that a 16-byte-aligned destination address is a 16-byte-aligned ring slot;
// `win` holds compressed bytes [win_base, win_base + WIN) for the sequence parser.
struct Wave {
    const uint8_t *__restrict__ src;
    uint8_t *__restrict__ dst;
    uint8_t *ring, *win;
    uint64_t n, need, win_base;
    uint32_t phase;
    int lane;

    __device__ __forceinline__ uint32_t slot(uint64_t pos) const { return (uint32_t)(pos + phase) & (RING - 1); }

    __device__ void refill(uint64_t q) {  // compressed bytes [q, q + WIN), zeros behind the end of the blob
        win_base = q;
#pragma unroll
        for (int k = 0; k < (int)(WIN / 1024); k++) {
            const uint32_t off = (uint32_t)(k * 64 + lane) * 16;
            const uint64_t a = q + off;
            u4 v = {0, 0, 0, 0};
            if (a + 16 <= n) {
                const U16B t = *reinterpret_cast<const U16B *>(src + a);
                v = u4{t.w[0], t.w[1], t.w[2], t.w[3]};
            } else if (a < n) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint64_t j = 0; j < n - a; j++) w[j >> 2] |= (uint32_t)src[a + j] << (8 * (j & 3));
                v = u4{w[0], w[1], w[2], w[3]};
            }
            *reinterpret_cast<u4 *>(win + off) = v;
        }
        __syncthreads();
    }
    // makes [q, q + span) readable through in(); span <= 512
    __device__ __forceinline__ void ensure(uint64_t q, uint32_t span) {
        if (q < win_base || q + span > win_base + WIN) refill(q);
    }
    __device__ __forceinline__ uint32_t in(uint64_t q) const { return win[(uint32_t)(q - win_base)]; }

    // short copy out of the staged input: src[q, q + len) -> output position `out`; len <= 256
    __device__ __forceinline__ void copy_short(uint64_t q, uint32_t len, uint64_t out) {
        for (uint32_t i = lane; i < len; i += 64) {
            const uint8_t b = win[(uint32_t)(q + i - win_base)];
            ring[slot(out + i)] = b;
            if (out + i < need) dst[out + i] = b;
        }
        __syncthreads();
    }

    // bulk copy straight from the compressed stream (long literal runs, stored blocks): 16 bytes per lane,
    // four loads in flight; destination and ring slots are 16-byte aligned by construction
    __device__ void copy_bulk(uint64_t q, uint64_t len, uint64_t out) {
        uint64_t head = (16 - ((uint64_t)(uintptr_t)(dst + out) & 15)) & 15;
        if (head > len) head = len;
        if ((uint64_t)lane < head) {
            const uint8_t b = src[q + lane];
            ring[slot(out + lane)] = b;
            if (out + lane < need) dst[out + lane] = b;
        }
        const uint64_t nvec = (len - head) / 16, body = head + nvec * 16;
        const uint8_t *s0 = src + q + head;
        const uint64_t o0 = out + head;
        uint64_t v = lane;
        for (; v + 192 < nvec; v += 256) {
            U16B t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] = *reinterpret_cast<const U16B *>(s0 + (v + 64 * u) * 16);
#pragma unroll
            for (int u = 0; u < 4; u++) put16(o0 + (v + 64 * u) * 16, t[u]);
        }
        for (; v < nvec; v += 64) put16(o0 + v * 16, *reinterpret_cast<const U16B *>(s0 + v * 16));
        const uint64_t tail = len - body;
        if ((uint64_t)lane < tail) {
            const uint8_t b = src[q + body + lane];
            ring[slot(out + body + lane)] = b;
            if (out + body + lane < need) dst[out + body + lane] = b;
        }
        __syncthreads();
    }
    __device__ __forceinline__ void put16(uint64_t pos, const U16B &t) {
        const u4 v = {t.w[0], t.w[1], t.w[2], t.w[3]};
        *reinterpret_cast<u4 *>(ring + slot(pos)) = v;
        if (pos + 16 <= need) {
            *reinterpret_cast<u4 *>(dst + pos) = v;
        } else if (pos < need) {
            for (uint64_t j = 0; j < need - pos; j++) dst[pos + j] = (uint8_t)(t.w[j >> 2] >> (8 * (j & 3)));
        }
    }

    // match: `len` bytes from `offset` back, through the ring, in chunks that cannot wrap onto their own source
    __device__ void copy_match(uint32_t offset, uint64_t len, uint64_t out) {
        while (len > 0) {
            const uint64_t chunk = len < (uint64_t)(RING - offset) ? len : (uint64_t)(RING - offset);
            const uint64_t from = out - offset;
            for (uint64_t i = lane; i < chunk; i += 64) {
                const uint32_t k = offset >= chunk ? (uint32_t)i : (uint32_t)(i % offset);
                const uint8_t b = ring[slot(from + k)];
                ring[slot(out + i)] = b;
                if (out + i < need) dst[out + i] = b;
            }
            __syncthreads();
            out += chunk;
            len -= chunk;
        }
    }
};

__global__ __launch_bounds__(64) void k_lz4_inflate(DevLz4Job *jobs, int njobs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if ((int)blockIdx.x >= njobs) return;
    DevLz4Job *job = jobs + blockIdx.x;
    Wave w;
    w.src = job->src;
    w.dst = job->dst;
    w.ring = lds;
    w.win = lds + RING;
    w.n = job->n;
    w.need = job->need;
    w.phase = (uint32_t)((uintptr_t)job->dst & 15);
    w.lane = threadIdx.x;
    w.win_base = ~0ull >> 1;  // nothing staged yet
    const uint64_t n = w.n, need = w.need;
    const int64_t cap = (int64_t)job->max_block;
    const bool independent = job->independent != 0;
    uint64_t p = 0, out = 0;
    bool ok = true, last_stored = false;

    while (ok && out < need) {
        if (n - p < 4) { ok = false; break; }
        w.ensure(p, 4);
        const uint32_t bs = w.in(p) | (w.in(p + 1) << 8) | (w.in(p + 2) << 16) | (w.in(p + 3) << 24);
        p += 4;
        if (bs == 0) { ok = false; break; }  // EndMark before `need` bytes: the host reader sorts out which error
        const uint64_t sz = bs & 0x7FFFFFFFu;
        if (sz > (uint64_t)cap) { ok = false; break; }
        if (bs >> 31) {  // stored block: passed through, possibly cut short (host/lz4_frame.cpp)
            const uint64_t avail = sz < n - p ? sz : n - p;
            w.copy_bulk(p, avail, out);
            out += avail;
            last_stored = true;
            if (out >= need) break;
            if (avail < sz) { ok = false; break; }
            p += sz;
            continue;
        }
        if (n - p < sz) { ok = false; break; }
        // ---- one compressed block: LZ4_decompress_safe's rules, positions relative to the block ----
        const int64_t in = (int64_t)sz;
        const uint64_t floor_out = independent ? out : 0;
        int64_t ip = 0, op = 0;
        if (in == 0) { ok = false; break; }
        for (;;) {
            w.ensure(p + (uint64_t)ip, 1);
            const uint32_t token = w.in(p + (uint64_t)ip);
            ip++;
            int64_t lit = token >> 4;
            if (lit == 15) {
                if (ip >= in - 15) { ok = false; break; }
                uint32_t x;
                do {
                    w.ensure(p + (uint64_t)ip, 1);
                    x = w.in(p + (uint64_t)ip);
                    ip++;
                    lit += x;
                } while (x == 255 && ip < in - 15);
            }
            const bool last_seq = op + lit > cap - 12 || ip + lit > in - 8;
            if (last_seq && (ip + lit != in || op + lit > cap)) { ok = false; break; }
            if (lit > 0) {
                if (lit <= 256) {
                    w.ensure(p + (uint64_t)ip, (uint32_t)lit + (last_seq ? 0 : 2));
                    w.copy_short(p + (uint64_t)ip, (uint32_t)lit, out);
                } else {
                    w.copy_bulk(p + (uint64_t)ip, (uint64_t)lit, out);
                }
            }
            out += (uint64_t)lit;
            ip += lit;
            op += lit;
            if (last_seq) break;
            w.ensure(p + (uint64_t)ip, 2);
            const uint32_t offset = w.in(p + (uint64_t)ip) | (w.in(p + (uint64_t)ip + 1) << 8);
            ip += 2;
            int64_t mlen = token & 15;
            if (mlen == 15) {
                uint32_t x;
                bool bad = false;
                do {
                    w.ensure(p + (uint64_t)ip, 1);
                    x = w.in(p + (uint64_t)ip);
                    ip++;
                    mlen += x;
                    if (ip >= in - 4) { bad = true; break; }
                } while (x == 255);
                if (bad) { ok = false; break; }
            }
            mlen += 4;
            if (offset == 0 || (uint64_t)offset > out - floor_out || op + mlen > cap - 5) { ok = false; break; }
            w.copy_match(offset, (uint64_t)mlen, out);
            out += (uint64_t)mlen;
            op += mlen;
        }
        if (!ok) break;
        p += sz;
        last_stored = false;
        if (p == n) { ok = false; break; }  // input ends right behind a compressed block: the host reader's "dry" rule decides
    }
    // what the read of the last needed byte still looks at (host/lz4_frame.cpp, end of frame_core)
    if (ok && out == need && !last_stored && n - p >= 4) {
        w.ensure(p, 4);
        const uint32_t x = w.in(p) | (w.in(p + 1) << 8) | (w.in(p + 2) << 16) | (w.in(p + 3) << 24);
        if (x == 0) {
            if (job->has_size && job->content_size != out) ok = false;
        } else if ((x & 0x7FFFFFFFu) > (uint32_t)cap) {
            ok = false;
        }
    }
    if (w.lane == 0) job->status = ok ? 0 : 1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
extern "C" int pcq_lz4_inflate_dev(pcq_ctx *ctx, pcq_lz4_job *jobs, size_t njobs, void *stream) {
    if (!ctx || (!jobs && njobs)) return pcq_fail(PCQ_ERR_ARG, "pcq_lz4_inflate_dev: null argument");
    if (njobs == 0) return PCQ_OK;
    if (njobs > (size_t)1 << 30) return pcq_fail(PCQ_ERR_ARG, "pcq_lz4_inflate_dev: too many jobs");
    PCQ_HIP(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    std::vector<DevLz4Job> table(njobs);
    size_t live = 0;
    std::vector<size_t> index(njobs);
    for (size_t i = 0; i < njobs; i++) {
        pcq_lz4_job &j = jobs[i];
        j.status = 1;
        const unsigned bsid = j.block_size_id;
        // frames this kernel does not take: block checksums, bad descriptors, nothing to do, too short to hold a block
        if (j.need == 0 || !j.src || !j.dst || bsid < 4 || bsid > 7 || j.block_checksum || j.src_len < 4) continue;
        DevLz4Job d;
        memset(&d, 0, sizeof d);
        d.src = (const uint8_t *)j.src;
        d.n = j.src_len;
        d.dst = (uint8_t *)j.dst;
        d.need = j.need;
        d.max_block = 1u << (8 + 2 * bsid);
        d.independent = j.independent_blocks ? 1 : 0;
        d.has_size = j.has_content_size ? 1 : 0;
        d.content_size = j.content_size;
        d.status = 1;
        index[live] = i;
        table[live++] = d;
    }
    if (live == 0) return PCQ_OK;
    DevLz4Job *d_jobs = nullptr;
    PCQ_HIP(hipMalloc((void **)&d_jobs, live * sizeof(DevLz4Job)));
    hipError_t e = hipMemcpyAsync(d_jobs, table.data(), live * sizeof(DevLz4Job), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        constexpr size_t kLds = RING + WIN;  // above the 64 KiB a kernel gets by default
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_lz4_inflate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_lz4_inflate, dim3((unsigned)live), dim3(64), kLds, s, d_jobs, (int)live);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(table.data(), d_jobs, live * sizeof(DevLz4Job), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_jobs);
    if (e != hipSuccess) return pcq_fail(PCQ_ERR_HIP, "pcq_lz4_inflate_dev: %s", hipGetErrorString(e));
    for (size_t k = 0; k < live; k++) jobs[index[k]].status = table[k].status;
    return PCQ_OK;
}

// Streams [file_offset, file_offset + bytes) of an open file into device memory through the context's
// pinned staging buffers (pread split over the copy pool, H2D overlapped with the next pread).
extern "C" int pcq_read_fd_to_device(pcq_ctx *ctx, int fd, uint64_t file_offset, uint64_t bytes, void *d_dst) {
    if (!ctx || fd < 0 || (!d_dst && bytes)) return pcq_fail(PCQ_ERR_ARG, "pcq_read_fd_to_device: bad argument");
    if (bytes == 0) return PCQ_OK;
    PCQ_HIP(hipSetDevice(ctx->device));
    return pcq_stream_fd_to_device(ctx, fd, file_offset, bytes, (uint8_t *)d_dst);
}
