// lz4_inflate.hip — LZ4 block inflate on the device for LAZER column blobs.
//
// A LAZER file holds thousands of independent LZ4 frames (one per attribute column per block,
// readers/src/lazer_reader.rs:136-265).  Inside a frame the blocks are usually *linked* (the lz4 crate's
// default): a match may reach back into the previous block, so a frame is a sequential job — but the
// frames are independent.  One wave inflates one frame:
//
//   * the 64 KiB LZ4 window lives in LDS as a ring (a wave's LDS accesses are ordered, so match copies
//     read bytes written a few instructions earlier without any global-memory coherence question);
//   * literal and match copies are wave-wide (64 bytes per step); overlapping matches use the periodic
//     form src = out - offset + (i mod offset), in chunks that cannot wrap the ring onto their own source;
//   * everything the wave reads or writes is bounds-checked, and every loop consumes input, so a damaged
//     frame ends the wave instead of faulting or spinning.
//
// The kernel accepts only what liblz4 accepts (the same end-of-block rules as host/lz4_frame.cpp) and
// reports anything else — damage, block checksums, a frame that ends early or at an odd place — as
// "not handled": the host layer then runs its own reader on that frame, which reproduces the reference's
// exact error behaviour.  The device path is a fast path for well-formed frames, never a second opinion.
#include "pcq_internal.h"

namespace {

constexpr uint32_t RING = 65536;

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

__device__ __forceinline__ uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// src[from, from+len) -> ring + dst (dst writes clipped to `need`); all lanes take part
__device__ __forceinline__ void copy_in(const uint8_t *__restrict__ src, uint64_t from, uint64_t len, uint8_t *__restrict__ dst,
                                        uint64_t out, uint64_t need, uint8_t *ring, int lane) {
    for (uint64_t i = lane; i < len; i += 64) {
        const uint8_t b = src[from + i];
        ring[(uint32_t)(out + i) & (RING - 1)] = b;
        if (out + i < need) dst[out + i] = b;
    }
}

__global__ __launch_bounds__(64) void k_lz4_inflate(DevLz4Job *jobs, int njobs) {
    __shared__ uint8_t ring[RING];
    if ((int)blockIdx.x >= njobs) return;
    DevLz4Job *job = jobs + blockIdx.x;
    const uint8_t *__restrict__ src = job->src;
    uint8_t *__restrict__ dst = job->dst;
    const uint64_t n = job->n, need = job->need;
    const int64_t cap = (int64_t)job->max_block;
    const bool independent = job->independent != 0;
    const int lane = threadIdx.x;
    uint64_t p = 0, out = 0;
    bool ok = true, last_stored = false;

    while (ok && out < need) {
        if (n - p < 4) { ok = false; break; }
        const uint32_t bs = rd32(src + p);
        p += 4;
        if (bs == 0) { ok = false; break; }  // EndMark before `need` bytes: the host reader sorts out which error
        const uint64_t sz = bs & 0x7FFFFFFFu;
        if (sz > (uint64_t)cap) { ok = false; break; }
        if (bs >> 31) {  // stored block: passed through, possibly cut short (host/lz4_frame.cpp)
            const uint64_t avail = sz < n - p ? sz : n - p;
            copy_in(src, p, avail, dst, out, need, ring, lane);
            __syncthreads();
            out += avail;
            last_stored = true;
            if (out >= need) break;
            if (avail < sz) { ok = false; break; }
            p += sz;
            continue;
        }
        if (n - p < sz) { ok = false; break; }
        // ---- one compressed block: LZ4_decompress_safe's rules, positions relative to the block ----
        const uint8_t *__restrict__ b = src + p;
        const int64_t in = (int64_t)sz;
        const uint64_t floor_out = independent ? out : 0;
        int64_t ip = 0, op = 0;
        if (in == 0) { ok = false; break; }
        for (;;) {
            const uint32_t token = b[ip++];
            int64_t lit = token >> 4;
            if (lit == 15) {
                if (ip >= in - 15) { ok = false; break; }
                uint32_t x;
                do {
                    x = b[ip++];
                    lit += x;
                } while (x == 255 && ip < in - 15);
            }
            const bool last_seq = op + lit > cap - 12 || ip + lit > in - 8;
            if (last_seq && (ip + lit != in || op + lit > cap)) { ok = false; break; }
            copy_in(b, (uint64_t)ip, (uint64_t)lit, dst, out, need, ring, lane);
            __syncthreads();
            out += (uint64_t)lit;
            ip += lit;
            op += lit;
            if (last_seq) break;
            const uint32_t offset = (uint32_t)b[ip] | ((uint32_t)b[ip + 1] << 8);
            ip += 2;
            int64_t mlen = token & 15;
            if (mlen == 15) {
                uint32_t x;
                bool bad = false;
                do {
                    x = b[ip++];
                    mlen += x;
                    if (ip >= in - 4) { bad = true; break; }
                } while (x == 255);
                if (bad) { ok = false; break; }
            }
            mlen += 4;
            if (offset == 0 || (uint64_t)offset > out - floor_out || op + mlen > cap - 5) { ok = false; break; }
            // match copy through the ring, in chunks that cannot wrap onto their own source region
            int64_t left = mlen;
            while (left > 0) {
                const int64_t chunk = left < (int64_t)(RING - offset) ? left : (int64_t)(RING - offset);
                const uint64_t from = out - offset;
                for (int64_t i = lane; i < chunk; i += 64) {
                    const uint32_t k = offset >= (uint32_t)chunk ? (uint32_t)i : (uint32_t)i % offset;
                    const uint8_t v = ring[(uint32_t)(from + k) & (RING - 1)];
                    ring[(uint32_t)(out + (uint64_t)i) & (RING - 1)] = v;
                    if (out + (uint64_t)i < need) dst[out + (uint64_t)i] = v;
                }
                __syncthreads();
                out += (uint64_t)chunk;
                left -= chunk;
            }
            op += mlen;
        }
        if (!ok) break;
        p += sz;
        last_stored = false;
        if (p == n) { ok = false; break; }  // input ends right behind a compressed block: the host reader's "dry" rule decides
    }
    // what the read of the last needed byte still looks at (host/lz4_frame.cpp, end of frame_core)
    if (ok && out == need && !last_stored && n - p >= 4) {
        const uint32_t w = rd32(src + p);
        if (w == 0) {
            if (job->has_size && job->content_size != out) ok = false;
        } else if ((w & 0x7FFFFFFFu) > (uint32_t)cap) {
            ok = false;
        }
    }
    if (lane == 0) job->status = ok ? 0 : 1;
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
        hipLaunchKernelGGL(k_lz4_inflate, dim3((unsigned)live), dim3(64), 0, s, d_jobs, (int)live);
        e = hipGetLastError();
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
