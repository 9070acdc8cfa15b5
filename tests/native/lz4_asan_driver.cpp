// Memory-safety driver for the product's LZ4 Frame reader (host code, no GPU): built with
// -fsanitize=address,undefined by tests/test_lz4_lazer.py and fed a corpus of damaged frames.
// Corpus record: u32 frame_len | u32 need | u32 unit | frame bytes.  Prints "ok <cases> <errors>".
#include <cstdio>
#include <cstring>
#include <vector>

#include "lz4_frame.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    unsigned long cases = 0, errors = 0;
    for (;;) {
        uint32_t hdr[3];
        if (fread(hdr, 4, 3, f) != 3) break;
        std::vector<uint8_t> frame(hdr[0]);
        if (hdr[0] && fread(frame.data(), 1, hdr[0], f) != hdr[0]) return 3;
        // exact-size heap copies so that any over-read / over-write trips the sanitizer
        uint8_t *src = new uint8_t[hdr[0] ? hdr[0] : 1];
        if (hdr[0]) memcpy(src, frame.data(), hdr[0]);
        uint8_t *dst = new uint8_t[hdr[1] ? hdr[1] : 1];
        pcq::Status a = pcq::lz4_frame_decode_into(src, hdr[0], hdr[1], hdr[2], dst);
        std::vector<uint8_t> v;
        pcq::Status b = pcq::lz4_frame_decode(src, hdr[0], hdr[1], hdr[2], &v);
        if (a.code != b.code) {
            fprintf(stderr, "case %lu: into=%d vector=%d\n", cases, a.code, b.code);
            return 4;
        }
        if (a.ok() && hdr[1] && memcmp(dst, v.data(), hdr[1]) != 0) return 5;
        errors += !a.ok();
        cases++;
        delete[] src;
        delete[] dst;
    }
    fclose(f);
    printf("ok %lu %lu\n", cases, errors);
    return 0;
}
