// Memory-safety + differential driver for the host layer's file parsers (no GPU call is made):
// parse_las_header (core.cpp) and LAZERSource::from as restated in lazer.cpp (lazer_file_bounds).
// Built with -fsanitize=address,undefined by tests/test_lz4_lazer.py; prints "<path> <header rc> <lazer rc>".
#include <cstdio>

#include "pcq_host.hpp"

int main(int argc, char **argv) {
    for (int i = 1; i < argc; i++) {
        pcq::MappedFile f;
        pcq::Status st = f.open(argv[i]);
        int hrc = st.code;
        if (st.ok()) {
            pcq::LasHeader h;
            hrc = pcq::parse_las_header(f.data(), f.size(), false, &h).code;
        }
        pcq::AABB b;
        const int lrc = pcq::lazer_file_bounds(argv[i], &b).code;
        printf("%s %d %d\n", argv[i], hrc, lrc);
    }
    return 0;
}
