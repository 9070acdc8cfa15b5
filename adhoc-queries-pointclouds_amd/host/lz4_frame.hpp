// lz4_frame.hpp — LZ4 Frame decoding for LAZER column blobs (see lz4_frame.cpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "pcq_host.hpp"

namespace pcq {

uint32_t xxh32(const uint8_t *p, size_t len);  // seed 0

// The first `need` bytes (at least; whole LZ4 blocks) of the frame stored in src[0, n), as a reader that
// pulls `unit` bytes per read_exact call would get them (0: one call for everything).
Status lz4_frame_decode(const uint8_t *src, size_t n, size_t need, size_t unit, std::vector<uint8_t> *out);
// Same, inflated directly into dst[0, need).
Status lz4_frame_decode_into(const uint8_t *src, size_t n, size_t need, size_t unit, uint8_t *dst);

}  // namespace pcq
