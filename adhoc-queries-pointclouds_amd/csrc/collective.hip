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
#include <unistd.h>

#include <chrono>
#include <cstdlib>
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
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
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
    // RCCL writes its banner and its NCCL_DEBUG output to stdout; the stdout of a query is the reference's (main.rs:289,
    // :179, :313-316) and nothing else.  Whatever level the environment asks for goes to stderr unless it names a file.
    // (A caller that builds the communicator beside threads that read the environment sets the variable itself, before
    // those threads exist — host/run_search.cpp does; then nothing is written here.)
    if (!getenv("NCCL_DEBUG_FILE")) setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    const auto t0 = std::chrono::steady_clock::now();
    for (const char *n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return pcq_fail(PCQ_ERR_HIP, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
    // (the library carries code objects for every architecture it supports: loading it registers them all with the HIP
    // runtime, which is where the seconds go — profiles/r03_rccl_cost.log)
    if (getenv("PCQ_TIMING") && getenv("PCQ_TIMING")[0] == '1')
        fprintf(stderr, "[pcq] librccl loaded in %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
#define BIND(field, sym)                                                               \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, sym)); \
    if (!g_rccl.field) return pcq_fail(PCQ_ERR_HIP, "RCCL symbol %s missing", sym);
    BIND(CommInitAll, "ncclCommInitAll")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(CommAbort, "ncclCommAbort")
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

bool timing() {
    static const bool on = getenv("PCQ_TIMING") && getenv("PCQ_TIMING")[0] == '1';
    return on;
}

// (g_mu held) the intra-node communicator for `devs`; leaves the calling thread on whatever device RCCL left it on
int ensure_comm(const std::vector<int> &devs) {
    int rc = load_rccl();
    if (rc) return rc;
    if (devs == g_rccl.devices) return PCQ_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (ncclComm_t c : g_rccl.comms) g_rccl.CommDestroy(c);
    g_rccl.comms.assign(devs.size(), nullptr);
    g_rccl.devices.clear();
    {
        // RCCL prints its version banner to stdout when a communicator is created (plain printf, whatever NCCL_DEBUG_FILE
        // says); the stdout of a query is the reference's and nothing else.  For the duration of the call descriptor 1 is
        // descriptor 2.  (The CLI joins the thread it builds the communicator on before it prints its first line behind the scans.)
        fflush(stdout);
        const int saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
        const ncclResult_t r = g_rccl.CommInitAll(g_rccl.comms.data(), (int)devs.size(), devs.data());
        fflush(stdout);
        if (saved >= 0) {
            (void)dup2(saved, 1);
            close(saved);
        }
        if (r != 0) return pcq_fail(PCQ_ERR_HIP, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
    }
    g_rccl.devices = devs;
    if (timing())
        fprintf(stderr, "[pcq] RCCL communicator over %zu device(s) ready after %.1f ms\n", devs.size(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return PCQ_OK;
}

}  // namespace

// Builds the communicator for `devices` now, so that the all-reduce finds it ready: loading librccl (1.0-5.0 s in a process that
// does not carry it yet: it registers its code objects with the HIP runtime, and context creation and launches on other
// threads wait for the runtime's lock meanwhile) and ncclCommInitAll (0.64 s on one device) — profiles/r03_rccl_cost.log.
// Synchronous and thread-safe: a caller that wants ncclCommInitAll beside its scans calls this from a thread of its own and
// joins it before the all-reduce (host/run_search.cpp does).  The library itself starts no thread: round 3 had the helper
// thread in here, and a process that ended while it was still inside RCCL died in the runtime's exit handlers — or, once
// an exit handler waited for it, hung in them.
extern "C" int pcq_allreduce_prepare(const int *devices, int n) {
    if (!devices || n < 1) return pcq_fail(PCQ_ERR_ARG, "pcq_allreduce_prepare: bad arguments");
    std::vector<int> devs(devices, devices + n);
    int prev = -1;
    (void)hipGetDevice(&prev);
    int rc;
    {
        std::lock_guard<std::mutex> g(g_mu);
        rc = ensure_comm(devs);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

extern "C" int pcq_allreduce_sum_u64(pcq_ctx *const *ctxs, const uint64_t *const *send, uint64_t *const *recv, int n) {
    if (!ctxs || !send || !recv || n < 1) return pcq_fail(PCQ_ERR_ARG, "pcq_allreduce_sum_u64: bad arguments");
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || !send[i] || !recv[i]) return pcq_fail(PCQ_ERR_ARG, "pcq_allreduce_sum_u64: null entry %d", i);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (ctxs[i]->device == ctxs[j]->device)
                return pcq_fail(PCQ_ERR_ARG, "pcq_allreduce_sum_u64: entries %d and %d are both on device %d (one rank per GPU)", j, i, ctxs[i]->device);
    const int inject = ctxs[0]->allreduce_fail;  // test hook: 1 = fail before anything is touched, 2 = fail after the reduction ran,
                                                 // 3 = fail inside the RCCL group with rank 0 already enqueued (as a later rank's failure would)
    if (inject == 1) return pcq_fail(PCQ_ERR_HIP, "pcq_allreduce_sum_u64: injected failure (before the reduction)");
    if (n == 1 && !ctxs[0]->allreduce_single_rank) {  // a single rank: the sum is the value itself
        PCQ_ON_DEVICE_OF_CTX(ctxs[0]);
        if (recv[0] != send[0]) PCQ_HIP(hipMemcpyAsync(recv[0], send[0], 8, hipMemcpyDeviceToDevice, ctxs[0]->stream));
        PCQ_HIP(hipStreamSynchronize(ctxs[0]->stream));
        if (inject == 2) return pcq_fail(PCQ_ERR_HIP, "pcq_allreduce_sum_u64: injected failure (after the reduction)");
        return PCQ_OK;
    }
    DeviceGuard restore(ctxs[0]->device);  // the calls below move the thread from device to device; put it back at the end
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) devs[i] = ctxs[i]->device;
    int rc = ensure_comm(devs);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    // Whatever fails between GroupStart and GroupEnd, the group is closed before this function returns: an open group
    // would swallow the next caller's collectives — but never as a collective some ranks are missing from (below).
    PCQ_NCCL(g_rccl.GroupStart());
    int failed = PCQ_OK;
    for (int i = 0; i < n && !failed; i++) {
        hipError_t he = hipSetDevice(devs[i]);
        if (he != hipSuccess) {
            failed = pcq_fail(PCQ_ERR_HIP, "hipSetDevice(%d) failed: %s", devs[i], hipGetErrorString(he));
            break;
        }
        const ncclResult_t r = g_rccl.AllReduce(send[i], recv[i], 1, kNcclUint64, kNcclSum, g_rccl.comms[i], ctxs[i]->stream);
        if (r != 0) failed = pcq_fail(PCQ_ERR_HIP, "ncclAllReduce (rank %d) failed: %s", i, g_rccl.GetErrorString(r));
        else if (inject == 3 && i == 0) failed = pcq_fail(PCQ_ERR_HIP, "pcq_allreduce_sum_u64: injected failure (inside the group, behind rank 0)");
    }
    if (failed) {
        // Ranks in front of the failing one are already part of the group: closing it as it is would launch a collective
        // that waits for ranks that never come — on the very streams the caller's fallback reads the counts with.  The
        // communicators are aborted first (kernels of an aborted communicator return), the group is closed whatever it
        // says, and the communicator is built again by whoever asks next.
        for (ncclComm_t c : g_rccl.comms)
            if (c) (void)g_rccl.CommAbort(c);
        (void)g_rccl.GroupEnd();
        g_rccl.comms.clear();
        g_rccl.devices.clear();
        return failed;
    }
    const ncclResult_t ge = g_rccl.GroupEnd();
    if (ge != 0) return pcq_fail(PCQ_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(ge));
    for (int i = 0; i < n; i++) {
        PCQ_HIP(hipSetDevice(devs[i]));
        PCQ_HIP(hipStreamSynchronize(ctxs[i]->stream));
    }
    if (timing())
        fprintf(stderr, "[pcq] all-reduce of %d count(s) took %.2f ms\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (inject == 2) return pcq_fail(PCQ_ERR_HIP, "pcq_allreduce_sum_u64: injected failure (after the reduction)");
    return PCQ_OK;
}
