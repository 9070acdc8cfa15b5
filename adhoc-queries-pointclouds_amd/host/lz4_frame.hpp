// lz4_frame.hpp — LZ4 Frame decoding for LAZER column blobs (see lz4_frame.cpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "pcq_host.hpp"

namespace pcq {

uint32_t xxh32(const uint8_t *p, size_t len);  // seed 0

// The frame descriptor, checked the way LZ4F_decompress checks it (magic, version, reserved bits, block
// size id, header checksum); `payload` is the offset of the first block header.
struct Lz4FrameInfo {
    size_t payload = 0, max_block = 0;
    unsigned block_size_id = 0;
    bool independent = false, block_checksum = false, has_size = false, content_checksum = false;
    uint64_t content_size = 0;
};
Status lz4_frame_descriptor(const uint8_t *src, size_t n, Lz4FrameInfo *out);

// The first `need` bytes (at least; whole LZ4 blocks) of the frame stored in src[0, n), as a reader that
// pulls `unit` bytes per read_exact call would get them (0: one call for everything).
Status lz4_frame_decode(const uint8_t *src, size_t n, size_t need, size_t unit, std::vector<uint8_t> *out);
// Same, inflated directly into dst[0, need).
Status lz4_frame_decode_into(const uint8_t *src, size_t n, size_t need, size_t unit, uint8_t *dst);

}  // namespace pcq
