// gather_bench.hip — developer microbenchmark (not part of the product): what the memory system delivers when a
// workgroup reads a "bin" of pass 0's output — one piece of L tuples out of every tile's block, blocks TILE tuples
// apart — as a function of the piece length and the tuple size, and what it takes when a tile of tuples leaves as
// runs of R tuples into many regions (the second level's output side).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/gather_bench tools/bench_src/gather_bench.hip
// usage: gather_bench [tuples=163000000]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
#define GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ uint32_t xcd_order(uint32_t it, uint32_t n) { return n % 8 == 0 ? (it % 8) * (n / 8) + it / 8 : it; }

// Workgroup w reads piece (bin, t) of the tiles t in its range; lane <-> consecutive tuple of the bin.
// TS = tuple bytes (16: one aligned 16-byte load; 20: a 4-byte aligned 16-byte load + a dword, as ld_tuple does).
template <int TS, int NT, int K>
__global__ __launch_bounds__(NT) void k_gather(const uint8_t *__restrict__ buf, uint32_t tile, uint32_t L, uint32_t magic, uint32_t nbins, uint32_t splits,
                                                uint32_t T, int xcd, uint32_t *__restrict__ sink) {
    const uint32_t nwork = nbins * splits;
    uint32_t acc = 0;
    for (uint32_t it = blockIdx.x; it < nwork; it += gridDim.x) {
        const uint32_t w = xcd ? xcd_order(it, nwork) : it;
        const uint32_t bin = w / splits, sp = w % splits;
        const uint32_t t0 = (uint32_t)((uint64_t)T * sp / splits), t1 = (uint32_t)((uint64_t)T * (sp + 1) / splits);
        const uint32_t total = (t1 - t0) * L;
        const uint8_t *base = buf + (uint64_t)t0 * tile * TS + (uint64_t)bin * L * TS;
        for (uint32_t j0 = 0; j0 < total; j0 += NT * K) {
            u32x4 v[K];
            uint32_t e[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                uint32_t j = j0 + k * NT + threadIdx.x;
                j = j < total ? j : total - 1;
                const uint32_t f = __umulhi(j, magic), o = j - f * L;
                const uint8_t *p = base + (uint64_t)f * tile * TS + o * TS;
                if (TS == 16) {
                    v[k] = *(const GLOBAL u32x4 *)p;
                    e[k] = 0;
                } else {
                    const u32x4_a4 a = *(const GLOBAL u32x4_a4 *)p;
                    v[k] = (u32x4){a.x, a.y, a.z, a.w};
                    e[k] = *(const GLOBAL uint32_t *)(p + 16);
                }
            }
#pragma unroll
            for (int k = 0; k < K; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w ^ e[k];
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

// The output side of a partition level: a workgroup's tile of NT * K tuples leaves as runs of R tuples; run q of step s goes
// to region (q * 97 + s * 13) % nreg of the workgroup's own set of regions, behind what the region already holds.
template <int TS, int NT, int K>
__global__ __launch_bounds__(NT) void k_scatter(uint8_t *__restrict__ out, uint32_t R, uint32_t magicR, uint32_t nreg, uint32_t cap, uint32_t steps, uint32_t nwork) {
    for (uint32_t it = blockIdx.x; it < nwork; it += gridDim.x) {
        uint8_t *base = out + (uint64_t)it * nreg * cap * TS;
        const uint32_t runs_per_step = NT * K / R;
        for (uint32_t s = 0; s < steps; s++) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t i = k * NT + threadIdx.x;
                const uint32_t q = __umulhi(i, magicR), o = i - q * R;
                if (q >= runs_per_step) continue;
                const uint32_t reg = (q * 97u + s * 13u) % nreg;
                // runs of one region per step: about runs_per_step / nreg; place them one behind the other
                const uint32_t nth = (q * 97u + s * 13u) / nreg % 4u;  // a few distinct offsets, good enough for a rate
                const uint32_t within = (s * 4u + nth) * R + o;
                if (within >= cap) continue;
                uint8_t *p = base + ((uint64_t)reg * cap + within) * TS;
                u32x4_a4 a = {i, s, reg, within};
                *(GLOBAL u32x4_a4 *)p = a;
                if (TS == 20) *(GLOBAL uint32_t *)(p + 16) = q;
            }
        }
    }
}

static uint32_t magic_of(uint32_t L) { return (uint32_t)(((1ull << 32) + L - 1) / L); }

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 163000000ull;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    uint8_t *buf = nullptr;
    uint32_t *sink = nullptr;
    const uint64_t bytes = n * 20 + (64 << 20);
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("# gather: tuples %llu, CUs %d\n", (unsigned long long)n, cus);
    printf("# ts tile L piece_bytes bins splits wgs threads xcd   ms   TB/s(useful)\n");
    auto run_gather = [&](int ts, uint32_t tile, uint32_t L, uint32_t splits, int nt, int xcd, int wg_per_cu) {
        const uint32_t nbins = tile / L;
        const uint32_t T = (uint32_t)(n / tile);
        const uint32_t nwork = nbins * splits;
        uint32_t grid = (uint32_t)cus * wg_per_cu;
        if (grid > nwork) grid = nwork;
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
#define LAUNCH(TS, NT, K) hipLaunchKernelGGL((k_gather<TS, NT, K>), dim3(grid), dim3(NT), 0, 0, buf, tile, L, magic_of(L), nbins, splits, T, xcd, sink)
            if (ts == 16 && nt == 1024) LAUNCH(16, 1024, 4);
            else if (ts == 20 && nt == 1024) LAUNCH(20, 1024, 4);
            else if (ts == 16 && nt == 256) LAUNCH(16, 256, 8);
            else if (ts == 20 && nt == 256) LAUNCH(20, 256, 8);
            else if (ts == 16 && nt == 512) LAUNCH(16, 512, 4);
            else LAUNCH(20, 512, 4);
#undef LAUNCH
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double useful = (double)T * nbins * L * ts;
        printf("%2d %5u %4u %5u %4u %3u %5u %4d %d  %7.3f  %6.2f\n", ts, tile, L, L * ts, nbins, splits, grid, nt, xcd, best, useful / best / 1e9);
        fflush(stdout);
    };
    // the shape of today's big fold: 512 bins, 5120-tuple tiles, one 1024-thread workgroup per CU
    for (int ts : {20, 16}) {
        for (uint32_t L : {10u, 20u, 40u, 80u, 160u, 320u}) {
            const uint32_t nb = 5120 / L;
            uint32_t splits = 1;
            while (nb * splits < 512) splits *= 2;  // keep 512 units of work
            run_gather(ts, 5120, L, splits, 1024, 1, 1);
        }
    }
    // xcd order off, for L = 10
    run_gather(20, 5120, 10, 1, 1024, 0, 1);
    run_gather(16, 5120, 10, 1, 1024, 0, 1);
    // larger tiles at 16 bytes
    for (uint32_t L : {16u, 32u, 64u, 128u}) {
        const uint32_t nb = 8192 / L;
        uint32_t splits = 1;
        while (nb * splits < 512) splits *= 2;
        run_gather(16, 8192, L, splits, 1024, 1, 1);
    }
    // more, smaller workgroups (no barriers needed by a filter): 256 threads x 8 in flight, 4 per CU
    for (int ts : {20, 16})
        for (uint32_t L : {10u, 40u, 80u, 160u}) {
            const uint32_t nb = 5120 / L;
            uint32_t splits = 1;
            while (nb * splits < 2048) splits *= 2;
            run_gather(ts, 5120, L, splits, 256, 1, 4);
        }
    // 512 threads, 2 per CU
    for (int ts : {20, 16})
        for (uint32_t L : {10u, 40u, 80u}) {
            const uint32_t nb = 5120 / L;
            uint32_t splits = 1;
            while (nb * splits < 1024) splits *= 2;
            run_gather(ts, 5120, L, splits, 512, 1, 2);
        }

    printf("# scatter: a tile of 4096 tuples leaves as runs of R tuples into nreg regions\n");
    printf("# ts R run_bytes nreg wgs   ms   TB/s\n");
    auto run_scatter = [&](int ts, uint32_t R, uint32_t nreg) {
        const uint32_t nwork = 512;
        const uint32_t per_wg = (uint32_t)(n / nwork);
        const uint32_t steps = per_wg / 4096;
        const uint32_t cap = (uint32_t)((uint64_t)steps * 4 * R + R);  // each region gets up to 4 runs a step in this model
        // keep the footprint inside the buffer
        uint32_t use_steps = steps;
        while ((uint64_t)nwork * nreg * ((uint64_t)use_steps * 4 * R + R) * ts > bytes) use_steps /= 2;
        const uint32_t cap2 = use_steps * 4 * R + R;
        (void)cap;
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            if (ts == 16) hipLaunchKernelGGL((k_scatter<16, 1024, 4>), dim3(cus), dim3(1024), 0, 0, buf, R, magic_of(R), nreg, cap2, use_steps, nwork);
            else hipLaunchKernelGGL((k_scatter<20, 1024, 4>), dim3(cus), dim3(1024), 0, 0, buf, R, magic_of(R), nreg, cap2, use_steps, nwork);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double written = (double)nwork * use_steps * (4096 / R) * R * ts;
        printf("%2d %4u %5u %5u %4d  %7.3f  %6.2f\n", ts, R, R * ts, nreg, cus, best, written / best / 1e9);
        fflush(stdout);
    };
    for (int ts : {20, 16})
        for (uint32_t R : {4u, 8u, 16u, 32u, 64u, 128u}) run_scatter(ts, R, 4096 / R < 16 ? 16 : 4096 / R);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    return 0;
}
