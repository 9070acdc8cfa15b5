// The cross-lane helpers of grid_common.h against their definition, on the GPU (developer check; tools/bin/wave_ops_check).
// hipcc -O3 --offload-arch=gfx950 -I include -I adhoc-queries-pointclouds_amd/csrc tools/bench_src/wave_ops_check.hip -o tools/bin/wave_ops_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "grid_common.h"

__global__ void k_check(const uint32_t *in, uint32_t *scan, uint32_t *next, uint32_t last) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = in[i];
    scan[i] = pcqgrid::wave_max_scan(v);
    next[i] = pcqgrid::wave_next_lane(v, last);
}

int main() {
    const int waves = 4096, n = waves * 64;
    std::vector<uint32_t> in(n), scan(n), next(n);
    srand(7);
    for (int i = 0; i < n; i++) {
        const int kind = (i / 64) % 4;
        in[i] = kind == 0 ? (uint32_t)rand() : kind == 1 ? ((rand() % 8) ? 0u : (uint32_t)(i % 64)) : kind == 2 ? (uint32_t)(rand() % 64) : ((i % 64) == 63 - (i / 64) % 64 ? 5u : 0u);
    }
    uint32_t *d_in, *d_scan, *d_next;
    hipMalloc(&d_in, n * 4), hipMalloc(&d_scan, n * 4), hipMalloc(&d_next, n * 4);
    hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, d_in, d_scan, d_next, 0xabcdu);
    hipMemcpy(scan.data(), d_scan, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(next.data(), d_next, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < waves; w++) {
        uint32_t m = 0;
        for (int l = 0; l < 64; l++) {
            const int i = w * 64 + l;
            m = in[i] > m ? in[i] : m;
            const uint32_t nx = l == 63 ? 0xabcdu : in[i + 1];
            if (scan[i] != m || next[i] != nx) {
                if (bad++ < 10) printf("wave %d lane %d: scan %u want %u, next %u want %u\n", w, l, scan[i], m, next[i], nx);
            }
        }
    }
    printf(bad ? "wave_ops_check: %d MISMATCHES\n" : "wave_ops_check: ok\n", bad);
    return bad ? 1 : 0;
}
