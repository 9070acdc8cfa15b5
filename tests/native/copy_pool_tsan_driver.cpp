// ThreadSanitizer stress of csrc/copy_pool.h (CPU build; the GPU box has no sanitizers): back-to-back jobs of
// alternating sizes, so that a helper still leaving the previous job meets the next one with a different number of
// slices.  Every job's destination is compared with its source.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "copy_pool.h"

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    const int helpers = argc > 2 ? atoi(argv[2]) : 7;
    CopyPool pool(helpers);
    const size_t big = 24u << 20;
    std::vector<uint8_t> src(big), dst(big);
    for (size_t i = 0; i < big; i++) src[i] = (uint8_t)(i * 2654435761u >> 24);
    const size_t sizes[] = {12u << 20, 2u << 20, 24u << 20, 3u << 20, (5u << 20) + 4097, 1u << 20, 8u << 20};
    unsigned long long moved = 0;
    for (int r = 0; r < rounds; r++) {
        const size_t n = sizes[r % 7];
        const size_t at = (size_t)(r * 7919) % (big - n + 1);
        memset(dst.data() + at, 0, n);
        const int rc = pool.run(-1, dst.data() + at, src.data() + at, n);
        if (rc != 0 || memcmp(dst.data() + at, src.data() + at, n) != 0) {
            printf("mismatch in round %d (rc %d)\n", r, rc);
            return 1;
        }
        moved += n;
    }
    printf("ok %d %llu\n", rounds, moved);
    return 0;
}
