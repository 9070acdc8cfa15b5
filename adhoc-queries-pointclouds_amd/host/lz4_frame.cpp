// lz4_frame.cpp — LZ4 Frame decoder (host side) for LAZER column blobs.
//
// The reference decodes LAZER attribute blobs with the `lz4` crate's `Decoder` (lz4 1.23.2 ->
// lz4-sys -> liblz4's LZ4F_decompress; Cargo.lock:462-474, readers/src/lazer_reader.rs:176-265).
// That library is not in the container; this is an implementation of the published LZ4 Frame and
// LZ4 Block formats (lz4_Frame_format.md v1.6.x, lz4_Block_format.md):
//   frame  = magic 0x184D2204 | FLG | BD | [content size u64] | [dict id u32] | HC
//            | { block size u32 (bit 31: stored) | data | [block checksum u32] }* | EndMark 0
//            | [content checksum u32]
//   block  = sequences of { token | [literal length bytes] | literals | offset u16 | [match length bytes] }
// Not modelled: the read-ahead the crate falls into after a stored (uncompressed) block, which lets the
// real reader notice damage behind the last byte it hands out.
// Header and block checksums (xxHash32) are verified where liblz4 verifies them; see lz4_frame_decode
// for what a streaming reader that stops after `need` bytes does and does not get to check.
#include "lz4_frame.hpp"

#include <algorithm>
#include <cstring>

namespace pcq {
namespace {

uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

}  // namespace

// xxHash32 (seed 0) — the checksum of the LZ4 frame format.
uint32_t xxh32(const uint8_t *p, size_t len) {
    constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const uint8_t *end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0u - P1;
        const uint8_t *limit = end - 16;
        do {
            v1 = rotl(v1 + rd32(p) * P2, 13) * P1;
            v2 = rotl(v2 + rd32(p + 4) * P2, 13) * P1;
            v3 = rotl(v3 + rd32(p + 8) * P2, 13) * P1;
            v4 = rotl(v4 + rd32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (p <= limit);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else {
        h = P5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) {
        h = rotl(h + rd32(p) * P3, 17) * P4;
        p += 4;
    }
    while (p < end) {
        h = rotl(h + (*p) * P5, 11) * P1;
        p++;
    }
    h ^= h >> 15;
    h *= P2;
    h ^= h >> 13;
    h *= P3;
    h ^= h >> 16;
    return h;
}

// Where inflated bytes go: either a caller-owned buffer of fixed capacity (the column slice of a LAZER
// block: no copy, no zero-fill) or a growing vector.
struct Out {
    uint8_t *p = nullptr;
    size_t size = 0, cap = 0;
    std::vector<uint8_t> *grow = nullptr;
    bool overflow = false;
    bool room(size_t extra) {
        if (size + extra <= cap) return true;
        if (!grow) {
            overflow = true;
            return false;
        }
        size_t nc = std::max<size_t>(cap * 2, size + extra + 64);
        grow->resize(nc);
        p = grow->data();
        cap = nc;
        return true;
    }
};
static Status overflowed() { return Status::Err(PCQ_ERR_ARG, "LZ4: destination too small"); }

// One LZ4 block appended to `out`; matches may reach back to `window_start` (linked blocks: the whole
// frame so far — offsets are 16-bit, liblz4 keeps 64 KiB of history; independent blocks: this block's own
// output).  Accepts and rejects exactly what liblz4's LZ4_decompress_safe (lz4.c, LZ4_decompress_generic
// with endOnInput, full decoding) does when LZ4F inflates a block into its `max_block`-sized buffer — a
// damaged file must fail here whenever it fails in the reference:
//   * a literal run that comes within 8 bytes of the end of the input, or within 12 bytes of the end of
//     the output buffer, has to be the last sequence: it must end exactly at the end of the input;
//   * an extended literal length may not start in the last 15 input bytes; after every byte of an extended
//     match length at least 5 input bytes must remain;
//   * a match may not end in the last 5 bytes of the output buffer, nor start before the window.
// (Offset 0 is rejected here; liblz4 1.9.3 copies stale buffer bytes for it, which has no defined result.)
static Status lz4_block(const uint8_t *src, size_t n, Out *out, size_t window_start, size_t max_block) {
    const Status bad = Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_decompressionFailed");
    const int64_t in = (int64_t)n, cap = (int64_t)max_block;
    const size_t base = out->size;
    int64_t ip = 0, op = 0;
    if (n == 0) return bad;
    for (;;) {
        const uint8_t token = src[ip++];
        int64_t lit = token >> 4;
        if (lit == 15) {
            if (ip >= in - 15) return bad;
            uint8_t b;
            do {
                b = src[ip++];
                lit += b;
            } while (b == 255 && ip < in - 15);  // running into the last 15 bytes ends the length early (lz4.c read_variable_length)
        }
        if (op + lit > cap - 12 || ip + lit > in - 8) {  // must be the last sequence
            if (ip + lit != in || op + lit > cap) return bad;
            if (!out->room((size_t)lit)) return overflowed();
            memcpy(out->p + out->size, src + ip, (size_t)lit);
            out->size += (size_t)lit;
            return Status::Ok();
        }
        if (lit) {
            if (!out->room((size_t)lit)) return overflowed();
            memcpy(out->p + out->size, src + ip, (size_t)lit);
            out->size += (size_t)lit;
            ip += lit;
            op += lit;
        }
        const size_t offset = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
        ip += 2;
        int64_t mlen = token & 15;
        if (mlen == 15) {
            uint8_t b;
            do {
                b = src[ip++];
                mlen += b;
                if (ip >= in - 4) return bad;
            } while (b == 255);
        }
        mlen += 4;
        if (offset == 0 || offset > out->size - window_start) return bad;
        if (op + mlen > cap - 5) return bad;
        if (!out->room((size_t)mlen)) return overflowed();
        uint8_t *dst = out->p + out->size;
        const uint8_t *from = dst - offset;
        if (offset >= (size_t)mlen) memcpy(dst, from, (size_t)mlen);
        else
            for (int64_t k = 0; k < mlen; k++) dst[k] = from[k];  // overlapping match: the pattern repeats
        out->size += (size_t)mlen;
        op += mlen;
    }
    (void)base;
}

// Decodes the frame at `src` until at least `need` bytes of content exist (whole blocks), the way the
// reference consumes a blob in reads of `unit` bytes (4 = read_i32, 1 = read_u8, 2 = read_u16): lz4::Decoder is a streaming reader and the LAZER reader pulls exactly
// count * size bytes out of it (lazer_reader.rs:590-716), so what lies behind the block holding the
// last needed byte — later blocks, the content checksum — is never looked at (see the end of this
// function for the one exception).  Running out of input, or out of frame, before `need` bytes is read_exact's
// UnexpectedEof ("failed to fill whole buffer"); a frame that ends early is checked (content checksum,
// content size) first, as LZ4F_decompress does at the EndMark.
// The frame descriptor: magic | FLG | BD | [content size] | [dictionary id] | header checksum.
Status lz4_frame_descriptor(const uint8_t *src, size_t n, Lz4FrameInfo *fi) {
    *fi = Lz4FrameInfo{};
    if (n < 4) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
    // A skippable frame is a complete frame to LZ4F_decompress: it returns 0 after it, lz4::Decoder then
    // treats the stream as finished (decoder.rs `self.next = 0`) and read_exact fails — the data frame
    // behind it is never reached (checked against the real library in tests/test_lz4_lazer.py).
    if ((rd32(src) & 0xFFFFFFF0u) == 0x184D2A50u) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
    if (rd32(src) != 0x184D2204u) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_frameType_unknown");
    const size_t hdr = 4;
    if (n - hdr < 3) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
    const uint8_t flg = src[hdr], bd = src[hdr + 1];
    if ((flg >> 6) != 1) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_headerVersion_wrong");
    if (flg & 0x02) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_reservedFlag_set");
    if (bd & 0x8F) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_reservedFlag_set");
    fi->block_size_id = (bd >> 4) & 7;
    if (fi->block_size_id < 4) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_maxBlockSize_invalid");
    fi->max_block = (size_t)1 << (8 + 2 * fi->block_size_id);  // 4: 64 KiB ... 7: 4 MiB
    fi->independent = flg & 0x20;
    fi->block_checksum = flg & 0x10;
    fi->has_size = flg & 0x08;
    fi->content_checksum = flg & 0x04;
    const bool has_dict = flg & 0x01;
    size_t p = hdr + 2;
    if (fi->has_size) {
        if (n - p < 8) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
        fi->content_size = (uint64_t)rd32(src + p) | ((uint64_t)rd32(src + p + 4) << 32);
        p += 8;
    }
    if (has_dict) {
        if (n - p < 4) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
        p += 4;
    }
    if (n - p < 1) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
    if (src[p] != ((xxh32(src + hdr, p - hdr) >> 8) & 0xFF)) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_headerChecksum_invalid");
    fi->payload = p + 1;
    return Status::Ok();
}

static Status frame_core(const uint8_t *src, size_t n, size_t need, size_t unit, Out *out) {
    if (need == 0) return Status::Ok();  // zero reads: the Decoder is never polled
    Lz4FrameInfo fi;
    Status hst = lz4_frame_descriptor(src, n, &fi);
    if (!hst.ok()) return hst;
    const size_t max_block = fi.max_block;
    const bool independent = fi.independent, block_checksum = fi.block_checksum, has_size = fi.has_size, content_checksum = fi.content_checksum;
    const uint64_t content_size = fi.content_size;
    size_t p = fi.payload;
    bool after_stored = false;
    while (out->size < need) {
        if (n - p < 4) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
        const uint32_t bs = rd32(src + p);
        p += 4;
        if (bs == 0) {  // EndMark before `need` bytes: the frame is closed, then the next read returns 0
            if (has_size && content_size != out->size) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_frameSize_wrong");
            if (content_checksum) {
                if (n - p < 4) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
                if (rd32(src + p) != xxh32(out->p, out->size)) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_contentChecksum_invalid");
            }
            return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
        }
        const bool stored = bs & 0x80000000u;
        const size_t sz = bs & 0x7FFFFFFFu;
        if (sz > max_block) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_maxBlockSize_invalid");
        const size_t block_out_start = out->size;
        after_stored = stored;
        if (stored) {
            // A stored block is passed through as its bytes arrive (LZ4F's copyDirect stage): what is there is
            // delivered even if the block is cut short, and its checksum is only looked at by the read after
            // the one that handed out its last byte (with the usual full-size stored block the crate's
            // 32 KiB input buffer ends exactly at the block end).
            const size_t avail = std::min(sz, n - p);
            if (!out->room(avail)) return overflowed();
            memcpy(out->p + out->size, src + p, avail);
            out->size += avail;
            if (out->size >= need) return Status::Ok();
            const bool complete = avail == sz && (!block_checksum || n - p - sz >= 4);
            if (!complete) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
            if (block_checksum && rd32(src + p + sz) != xxh32(src + p, sz)) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_blockChecksum_invalid");
        } else {
            if (n - p < sz || (block_checksum && n - p - sz < 4)) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
            if (block_checksum && rd32(src + p + sz) != xxh32(src + p, sz)) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_blockChecksum_invalid");
            Status st = lz4_block(src + p, sz, out, independent ? block_out_start : 0, max_block);
            if (!st.ok()) return st;
            // A compressed block is inflated into liblz4's own buffer and handed out `unit` bytes per read.
            // lz4::Decoder only calls LZ4F_decompress while it holds unread INPUT bytes or can fetch some
            // (decoder.rs: `if self.pos >= self.len { self.len = self.r.read(..)?; if self.len == 0 { break } }`),
            // so when the input ends exactly behind this block, the read that inflated it is the last one to
            // deliver anything: the rest of the block is lost to an UnexpectedEof.
            if (p + sz + (block_checksum ? 4 : 0) == n) {
                const size_t u = unit ? unit : need;
                const size_t deliverable = std::min(out->size, (block_out_start / u + 1) * u);
                if (need > deliverable) return Status::Err(PCQ_ERR_EOF, "failed to fill whole buffer");
                return Status::Ok();
            }
        }
        p += sz + (block_checksum ? 4 : 0);
    }
    // The call that hands out the last byte of a block goes on to the next block header when it is there
    // (LZ4F_decompress keeps changing stage while it needs no output space, and the crate feeds it the
    // 4 header bytes together with the block): an EndMark gets the content-size check — but not the
    // content checksum, whose 4 bytes are only fetched by a read that never comes — and an oversized
    // block header is rejected.
    // (Behind a stored block the crate's input buffering decides what else liblz4 gets to see — not modelled.)
    if (out->size == need && !after_stored && n - p >= 4) {
        const uint32_t bs = rd32(src + p);
        if (bs == 0) {
            if (has_size && content_size != out->size) return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_frameSize_wrong");
        } else if ((bs & 0x7FFFFFFFu) > max_block) {
            return Status::Err(PCQ_ERR_HEADER, "LZ4 error: ERROR_maxBlockSize_invalid");
        }
    }
    return Status::Ok();
}

Status lz4_frame_decode(const uint8_t *src, size_t n, size_t need, size_t unit, std::vector<uint8_t> *out) {
    out->clear();
    out->resize(std::min(need, n * 255) + 64);  // LZ4 cannot expand by more than 255x
    Out o;
    o.p = out->data();
    o.cap = out->size();
    o.grow = out;
    Status st = frame_core(src, n, need, unit, &o);
    out->resize(st.ok() ? o.size : 0);
    return st;
}

// Straight into dst[0, need): the usual case, where the block that holds byte need-1 ends there too.  A
// frame that carries more than asked for spills; that (rare) case is redone through a vector.
Status lz4_frame_decode_into(const uint8_t *src, size_t n, size_t need, size_t unit, uint8_t *dst) {
    Out o;
    o.p = dst;
    o.cap = need;
    Status st = frame_core(src, n, need, unit, &o);
    if (!o.overflow) return st;
    std::vector<uint8_t> tmp;
    st = lz4_frame_decode(src, n, need, unit, &tmp);
    if (st.ok()) memcpy(dst, tmp.data(), need);
    return st;
}

}  // namespace pcq
