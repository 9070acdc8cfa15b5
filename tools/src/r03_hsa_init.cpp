// How much of hipInit is the HSA runtime underneath it (would a HIP-free start-up be faster?): hsa_init alone, then the first HIP call.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <chrono>
#include <cstdio>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double t0 = now_ms();
    hsa_status_t st = hsa_init();
    double t1 = now_ms();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    double t2 = now_ms();
    hipStream_t s;
    e = hipSetDevice(0);
    e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double t3 = now_ms();
    printf("hsa_init %7.1f ms (status %d)   first HIP call behind it %7.1f ms   first queue %7.1f ms\n", t1 - t0, (int)st, t2 - t1, t3 - t2);
    (void)e;
    return 0;
}
