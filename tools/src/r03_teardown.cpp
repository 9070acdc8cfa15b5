// What a process leaves for the driver to tear down, seen by the NEXT process: run back to back, each prints its own hsa_init time.
// argv[1]: what this process creates before it ends — 0 nothing beyond hsa_init + hipInit, 1 + two streams, 2 + 4 GiB of device memory
// (touched by a memset), 3 + 96 MiB pinned host memory, 4 = 2 + 3;  argv[2]: 0 return from main, 1 _exit(0), 2 free everything then _exit
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int what = argc > 1 ? atoi(argv[1]) : 0, how = argc > 2 ? atoi(argv[2]) : 0;
    double t0 = now_ms();
    hsa_init();
    double t1 = now_ms();
    int n = 0;
    (void)hipGetDeviceCount(&n);
    (void)hipSetDevice(0);
    hipStream_t s[2] = {nullptr, nullptr};
    void *dev = nullptr, *pin = nullptr;
    if (what >= 1) { (void)hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking); }
    if (what == 2 || what == 4) { (void)hipMalloc(&dev, (size_t)4 << 30); (void)hipMemsetAsync(dev, 1, (size_t)4 << 30, s[0]); (void)hipStreamSynchronize(s[0]); }
    if (what == 3 || what == 4) (void)hipHostMalloc(&pin, (size_t)96 << 20, hipHostMallocDefault);
    double t2 = now_ms();
    printf("hsa_init %7.1f ms   set-up %7.1f ms\n", t1 - t0, t2 - t1);
    fflush(stdout);
    if (how == 2) { if (dev) (void)hipFree(dev); if (pin) (void)hipHostFree(pin); if (s[0]) { (void)hipStreamDestroy(s[0]); (void)hipStreamDestroy(s[1]); } }
    if (how >= 1) _exit(0);
    return 0;
}
