// collective.hip — the one collective of the path: the sum of the per-GPU match counts
// (query/src/main.rs:164-180), as a single RCCL all-reduce of one u64 per GPU over xGMI, for callers
// that drive several GPUs from ONE process (the `query` CLI with --gpus N).  Callers that run one
// process per GPU (bench.py under torch.distributed) all-reduce the same device counter through
// their own communicator instead.
//
// RCCL is bound at run time (dlopen) rather than at link time: a process that already carries its
// own RCCL (PyTorch ships one) must not get a second copy mixed in through libpcq.so's dependencies.
// Message: 8 bytes per rank — latency-bound; ring/tree choice and the 7 x 153 GB/s links are irrelevant.
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "pcq_internal.h"

namespace {

typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;  // 0 = ncclSuccess
constexpr int kNcclUint64 = 5;  // ncclDataType_t::ncclUint64
constexpr int kNcclSum = 0;     // ncclRedOp_t::ncclSum

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};

Rccl g_rccl;
std::mutex g_mu;

int load_rccl() {
    if (g_rccl.lib) return PCQ_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return pcq_fail(PCQ_ERR_HIP, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
#define BIND(field, sym)                                                               \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, sym)); \
    if (!g_rccl.field) return pcq_fail(PCQ_ERR_HIP, "RCCL symbol %s missing", sym);
    BIND(CommInitAll, "ncclCommInitAll")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(AllReduce, "ncclAllReduce")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    return PCQ_OK;
}

#define PCQ_NCCL(expr)                                                                                        \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != 0) return pcq_fail(PCQ_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));         \
    } while (0)

}  // namespace

extern "C" int pcq_allreduce_sum_u64(pcq_ctx *const *ctxs, uint64_t *const *device_counters, int n) {
    if (!ctxs || !device_counters || n < 1) return pcq_fail(PCQ_ERR_ARG, "pcq_allreduce_sum_u64: bad arguments");
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || !device_counters[i]) return pcq_fail(PCQ_ERR_ARG, "pcq_allreduce_sum_u64: null entry %d", i);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (ctxs[i]->device == ctxs[j]->device)
                return pcq_fail(PCQ_ERR_ARG, "pcq_allreduce_sum_u64: entries %d and %d are both on device %d (one rank per GPU)", j, i, ctxs[i]->device);
    if (n == 1 && !ctxs[0]->allreduce_single_rank) {  // a single rank: the sum is the value itself
        PCQ_ON_DEVICE_OF_CTX(ctxs[0]);
        PCQ_HIP(hipStreamSynchronize(ctxs[0]->stream));
        return PCQ_OK;
    }
    DeviceGuard restore(ctxs[0]->device);  // the calls below move the thread from device to device; put it back at the end
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = load_rccl();
    if (rc) return rc;
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) devs[i] = ctxs[i]->device;
    if (devs != g_rccl.devices) {  // (re)build the intra-node communicator for this device list
        for (ncclComm_t c : g_rccl.comms) g_rccl.CommDestroy(c);
        g_rccl.comms.assign(n, nullptr);
        g_rccl.devices.clear();
        PCQ_NCCL(g_rccl.CommInitAll(g_rccl.comms.data(), n, devs.data()));
        g_rccl.devices = devs;
    }
    PCQ_NCCL(g_rccl.GroupStart());
    for (int i = 0; i < n; i++) {
        PCQ_HIP(hipSetDevice(devs[i]));
        PCQ_NCCL(g_rccl.AllReduce(device_counters[i], device_counters[i], 1, kNcclUint64, kNcclSum, g_rccl.comms[i], ctxs[i]->stream));
    }
    PCQ_NCCL(g_rccl.GroupEnd());
    for (int i = 0; i < n; i++) {
        PCQ_HIP(hipSetDevice(devs[i]));
        PCQ_HIP(hipStreamSynchronize(ctxs[i]->stream));
    }
    return PCQ_OK;
}
