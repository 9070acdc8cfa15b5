// Where a fresh process's HIP start-up goes, and what it would cost to hand bytes read BEFORE the context exists to the GPU
// ("start reading before HIP is up").  One process = one measurement; tools/r03_hip_startup.sh runs it
// several times.  Not part of the product.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_touch(unsigned *p) { p[threadIdx.x] = threadIdx.x; }
static void fill(char *p, size_t n, int threads) {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back([=] { memset(p + n / threads * t, t + 1, n / threads); });
    for (auto &t : th) t.join();
}
int main(int argc, char **argv) {
    const size_t GB = argc > 1 ? (size_t)atoll(argv[1]) << 20 : (size_t)1 << 30;
    double t0 = now_ms(), t;
    int n = 0;
    CK(hipGetDeviceCount(&n));
    t = now_ms(); printf("hipGetDeviceCount (hipInit)        %8.1f ms\n", t - t0); t0 = t;
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    t = now_ms(); printf("hipSetDevice + properties          %8.1f ms\n", t - t0); t0 = t;
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    t = now_ms(); printf("two streams                        %8.1f ms\n", t - t0); t0 = t;
    unsigned *d; CK(hipMalloc((void **)&d, 512));
    t = now_ms(); printf("first hipMalloc (512 B)            %8.1f ms\n", t - t0); t0 = t;
    void *h; CK(hipHostMalloc(&h, 512, hipHostMallocDefault));
    t = now_ms(); printf("first hipHostMalloc (512 B)        %8.1f ms\n", t - t0); t0 = t;
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d); CK(hipStreamSynchronize(s));
    t = now_ms(); printf("first kernel (module load) + sync  %8.1f ms\n", t - t0); t0 = t;
    void *ring; CK(hipHostMalloc(&ring, (size_t)64 << 20, hipHostMallocDefault));
    t = now_ms(); printf("hipHostMalloc 64 MiB               %8.1f ms\n", t - t0); t0 = t;
    char *dev; CK(hipMalloc((void **)&dev, GB));
    t = now_ms(); printf("hipMalloc %zu MiB                 %8.1f ms\n", GB >> 20, t - t0); t0 = t;
    // bytes that were read before the context existed: page-aligned, touched
    char *pre = (char *)aligned_alloc(1 << 21, GB);
    fill(pre, GB, 8);
    t = now_ms(); printf("(host fill of %zu MiB, 8 threads  %8.1f ms)\n", GB >> 20, t - t0); t0 = t;
    CK(hipMemcpyAsync(dev, pre, GB, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    t = now_ms(); printf("H2D from pageable memory           %8.1f ms = %5.1f GB/s\n", t - t0, GB / 1e6 / (t - t0)); t0 = t;
    CK(hipHostRegister(pre, GB, hipHostRegisterDefault));
    t = now_ms(); printf("hipHostRegister                    %8.1f ms = %5.1f GB/s\n", t - t0, GB / 1e6 / (t - t0)); t0 = t;
    CK(hipMemcpyAsync(dev, pre, GB, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    t = now_ms(); printf("H2D from the registered range      %8.1f ms = %5.1f GB/s\n", t - t0, GB / 1e6 / (t - t0)); t0 = t;
    CK(hipHostUnregister(pre));
    t = now_ms(); printf("hipHostUnregister                  %8.1f ms\n", t - t0); t0 = t;
    // registering in chunks of 24 MiB (one staging chunk)
    const size_t C = (size_t)24 << 20;
    double reg = 0;
    for (size_t o = 0; o + C <= GB; o += C) { double a = now_ms(); CK(hipHostRegister(pre + o, C, hipHostRegisterDefault)); reg += now_ms() - a; }
    t = now_ms(); printf("hipHostRegister in 24 MiB pieces   %8.1f ms = %5.1f GB/s\n", reg, GB / 1e6 / reg); t0 = t;
    char *pin; CK(hipHostMalloc((void **)&pin, GB, hipHostMallocDefault));
    t = now_ms(); printf("hipHostMalloc %zu MiB             %8.1f ms\n", GB >> 20, t - t0); t0 = t;
    fill(pin, GB, 8);
    t = now_ms(); printf("(fill of the pinned range         %8.1f ms = %5.1f GB/s)\n", t - t0, GB / 1e6 / (t - t0)); t0 = t;
    CK(hipMemcpyAsync(dev, pin, GB, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    t = now_ms(); printf("H2D from hipHostMalloc memory      %8.1f ms = %5.1f GB/s\n", t - t0, GB / 1e6 / (t - t0)); t0 = t;
    return 0;
}
