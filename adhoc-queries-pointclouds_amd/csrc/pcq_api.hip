// pcq_api.hip — the C ABI of include/pcq.h: context, device-resident collectors, the scan entry
// points and the host-block streaming pipeline.
//
// Everything here is plumbing around the kernels of scan_count.hip / scan_generic.hip / grid_*.hip.
// There is deliberately no CPU implementation of any scan in this library: if no HIP device is
// usable, pcq_init fails and nothing else can be called.
#include "pcq_internal.h"

#include <unistd.h>

#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <new>

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024];

int pcq_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *pcq_last_error(void) { return g_err; }
// a few 64-bit words from device memory into pinned host memory (pcq_copy_to_host)
__global__ void k_words_to_host(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst_pinned, uint32_t n) {
    if (threadIdx.x < n) dst_pinned[threadIdx.x] = src[threadIdx.x];
}

extern "C" int pcq_abi_version(void) { return PCQ_ABI_VERSION; }
static void join_stage_warm(pcq_ctx *ctx);
static void join_copy_warm(pcq_ctx *ctx);

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
// Which NUMA node is the GPU attached to, and which CPUs belong to it (Linux sysfs; silently unknown elsewhere).
// DMA out of pinned host memory on the GPU's own socket runs at the PCIe rate; across the socket
// interconnect it was measured at about two thirds of it (profiles/r01_file_path_rate.log).
static void detect_numa_node(pcq_ctx *ctx) {
    CPU_ZERO(&ctx->node_cpus);
    ctx->numa_node = -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, ctx->device) != hipSuccess) return;
    for (char *p = bus; *p; p++) *p = (char)tolower((unsigned char)*p);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    if (node < 0) return;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return;
    char list[1024] = {0};
    if (fgets(list, sizeof list, f)) {
        for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
            int a = 0, b = 0;
            const int k = sscanf(tok, "%d-%d", &a, &b);
            if (k == 1) b = a;
            if (k >= 1)
                for (int c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET(c, &ctx->node_cpus);
        }
    }
    fclose(f);
    if (CPU_COUNT(&ctx->node_cpus) > 0) ctx->numa_node = node;
}

extern "C" int pcq_init(int device, pcq_ctx **out_ctx) {
    if (!out_ctx) return pcq_fail(PCQ_ERR_ARG, "pcq_init: out_ctx is null");
    *out_ctx = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return pcq_fail(PCQ_ERR_HIP, "pcq_init: no HIP device available (%s); this library has no CPU path",
                        e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= ndev) return pcq_fail(PCQ_ERR_ARG, "pcq_init: device %d out of range [0,%d)", device, ndev);
    PCQ_HIP(hipSetDevice(device));
    pcq_ctx *ctx = new (std::nothrow) pcq_ctx();
    if (!ctx) return pcq_fail(PCQ_ERR_NOMEM, "pcq_init: out of memory");
    ctx->device = device;
    e = hipGetDeviceProperties(&ctx->prop, device);
    if (e != hipSuccess) {
        delete ctx;
        return pcq_fail(PCQ_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    }
    ctx->num_cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    detect_numa_node(ctx);
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess) {
        pcq_shutdown(ctx);
        return pcq_fail(PCQ_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < 2; i++) {
        if ((e = hipEventCreateWithFlags(&ctx->copied[i], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->consumed[i], hipEventDisableTiming)) != hipSuccess) {
            pcq_shutdown(ctx);
            return pcq_fail(PCQ_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
        }
    }
    if ((e = hipMalloc((void **)&ctx->d_scalars, 64 * sizeof(uint64_t))) != hipSuccess ||
        (e = hipHostMalloc((void **)&ctx->h_scalars, 64 * sizeof(uint64_t), hipHostMallocDefault)) != hipSuccess) {
        pcq_shutdown(ctx);
        return pcq_fail(PCQ_ERR_HIP, "scratch allocation: %s", hipGetErrorString(e));
    }
    int rc = pcq_ensure_partials(ctx, (size_t)ctx->num_cus * 16);
    if (rc) {
        pcq_shutdown(ctx);
        return rc;
    }
    {  // (not more copy threads than this GPU's share of the host's hardware threads)
        const unsigned hw = std::thread::hardware_concurrency();
        const int share = hw ? (int)(hw / (unsigned)ndev) : ctx->copy_threads;
        if (ctx->copy_threads > share) ctx->copy_threads = share < 2 ? 2 : share;
    }
    // tuning / test knobs (same meaning as pcq_set_option)
    if (const char *e = getenv("PCQ_CHUNK_POINTS")) {
        const long long v = atoll(e);
        if (v >= 4) ctx->chunk_points = (uint64_t)v;
    }
    if (const char *e = getenv("PCQ_NUMA_LOCAL")) ctx->numa_local = atoi(e) != 0;
    if (const char *e = getenv("PCQ_HOST_IN_PLACE")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 2) ctx->host_in_place = v;
    }
    if (const char *e = getenv("PCQ_COPY_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) ctx->copy_threads = v;
    }
#ifdef PCQ_LAB
    if (const char *e = getenv("PCQ_BATCH_VARIANT")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 3) ctx->batch_variant = v;
    }
    if (const char *e = getenv("PCQ_BATCH_WAVES_PER_CU")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 32) ctx->batch_waves_per_cu = v;
    }
    if (const char *e = getenv("PCQ_K1_WAVES_PER_CU")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 32) ctx->k1_waves_per_cu = v;
    }
    if (const char *e = getenv("PCQ_K1_VARIANT")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 14) ctx->k1_variant = v;
    }
#endif
    *out_ctx = ctx;
    return PCQ_OK;
}

extern "C" int pcq_shutdown(pcq_ctx *ctx) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx) return PCQ_OK;
    (void)hipSetDevice(ctx->device);
    join_stage_warm(ctx);
    join_copy_warm(ctx);
    if (ctx->copy_warm_h) (void)hipHostFree(ctx->copy_warm_h);
    if (ctx->copy_warm_d) (void)hipFree(ctx->copy_warm_d);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (int i = 0; i < 2; i++) {
        if (ctx->copied[i]) (void)hipEventDestroy(ctx->copied[i]);
        if (ctx->consumed[i]) (void)hipEventDestroy(ctx->consumed[i]);
        if (ctx->h_stage[i]) (void)hipHostFree(ctx->h_stage[i]);
        if (ctx->d_stage[i]) (void)hipFree(ctx->d_stage[i]);
    }
    delete ctx->copy_pool;
    ctx->copy_pool = nullptr;
    pcq_pool_clear(ctx);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_scalars) (void)hipFree(ctx->d_scalars);
    if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
    if (ctx->d_segments) (void)hipFree(ctx->d_segments);
    if (ctx->h_segments) (void)hipHostFree(ctx->h_segments);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
    return PCQ_OK;
}

// ---------------------------------------------------------------------------------------------
// device-memory pool
// ---------------------------------------------------------------------------------------------
int pcq_pool_alloc(pcq_ctx *ctx, size_t bytes, void **out) {
    *out = nullptr;
    if (bytes == 0) bytes = 256;
    PoolBlock *best = nullptr;  // the smallest free block that is large enough, and not wastefully larger
    for (PoolBlock &b : ctx->pool)
        if (!b.used && b.bytes >= bytes && (b.bytes <= 2 * bytes + (64u << 20)) && (!best || b.bytes < best->bytes)) best = &b;
    if (best) {
        best->used = true;
        *out = best->p;
        return PCQ_OK;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {  // give the free blocks back and try once more
        (void)hipGetLastError();
        for (size_t i = 0; i < ctx->pool.size();) {
            if (!ctx->pool[i].used) {
                (void)hipFree(ctx->pool[i].p);
                ctx->pool.erase(ctx->pool.begin() + (long)i);
            } else {
                i++;
            }
        }
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return pcq_fail(PCQ_ERR_NOMEM, "device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    ctx->pool.push_back(PoolBlock{p, bytes, true});
    if (bytes >= (64u << 20)) {
        static const bool timing = getenv("PCQ_TIMING") && getenv("PCQ_TIMING")[0] == '1';
        if (timing) fprintf(stderr, "pcq: pool block %p (%zu MiB, address mod 2 MiB = %zu KiB)\n", p, bytes >> 20, ((size_t)p & ((2u << 20) - 1)) >> 10);
    }
    *out = p;
    return PCQ_OK;
}

void pcq_pool_free(pcq_ctx *ctx, void *p) {
    if (!p) return;
    uint64_t free_bytes = 0;
    for (PoolBlock &b : ctx->pool) {
        if (b.p == p) b.used = false;
        if (!b.used) free_bytes += b.bytes;
    }
    while (free_bytes > ctx->pool_limit) {  // keep the pool bounded: drop the largest free block
        size_t big = ctx->pool.size();
        for (size_t i = 0; i < ctx->pool.size(); i++)
            if (!ctx->pool[i].used && (big == ctx->pool.size() || ctx->pool[i].bytes > ctx->pool[big].bytes)) big = i;
        if (big == ctx->pool.size()) break;
        free_bytes -= ctx->pool[big].bytes;
        (void)hipFree(ctx->pool[big].p);
        ctx->pool.erase(ctx->pool.begin() + (long)big);
    }
}

void pcq_pool_clear(pcq_ctx *ctx) {
    for (PoolBlock &b : ctx->pool) (void)hipFree(b.p);
    ctx->pool.clear();
}

int pcq_ensure_partials(pcq_ctx *ctx, size_t n) {
    if (n <= ctx->partials_cap) return PCQ_OK;
    // The old buffer may still be referenced by kernels enqueued on the context's or a caller's stream.
    if (ctx->d_partials) {
        PCQ_HIP(hipDeviceSynchronize());
        PCQ_HIP(hipFree(ctx->d_partials));
        ctx->d_partials = nullptr;
        ctx->partials_cap = 0;
    }
    size_t cap = 4096;
    while (cap < n && cap < ((size_t)1 << 22)) cap <<= 1;
    if (cap < n) cap = (n + (((size_t)1 << 22) - 1)) & ~(((size_t)1 << 22) - 1);  // beyond 32 MB: whole 32 MB steps, not the next power of two
    PCQ_HIP(hipMalloc((void **)&ctx->d_partials, cap * sizeof(uint64_t)));
    ctx->partials_cap = cap;
    return PCQ_OK;
}

extern "C" int pcq_get_device_info(pcq_ctx *ctx, pcq_device_info *out) {
    if (!ctx || !out) return pcq_fail(PCQ_ERR_ARG, "pcq_get_device_info: null argument");
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", ctx->prop.name);
    snprintf(out->gcn_arch, sizeof out->gcn_arch, "%s", ctx->prop.gcnArchName);
    out->compute_units = ctx->prop.multiProcessorCount;
    out->wavefront_size = ctx->prop.warpSize;
    out->hbm_bytes = (uint64_t)ctx->prop.totalGlobalMem;
    out->lds_bytes_per_block = (uint64_t)ctx->prop.sharedMemPerBlock;
    out->clock_khz = ctx->prop.clockRate;
    return PCQ_OK;
}

extern "C" void *pcq_ctx_stream(pcq_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int pcq_ctx_synchronize(pcq_ctx *ctx) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx) return pcq_fail(PCQ_ERR_ARG, "pcq_ctx_synchronize: null context");
    PCQ_HIP(hipStreamSynchronize(ctx->stream));
    PCQ_HIP(hipStreamSynchronize(ctx->copy_stream));
    ctx->stage_busy[0] = ctx->stage_busy[1] = false;
    return PCQ_OK;
}

extern "C" int pcq_bind_thread_near_device(pcq_ctx *ctx) {
    if (!ctx) return pcq_fail(PCQ_ERR_ARG, "pcq_bind_thread_near_device: null context");
    if (ctx->numa_local && ctx->numa_node >= 0) (void)sched_setaffinity(0, sizeof ctx->node_cpus, &ctx->node_cpus);
    return PCQ_OK;
}

extern "C" int pcq_set_option(pcq_ctx *ctx, const char *key, int64_t value) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx || !key) return pcq_fail(PCQ_ERR_ARG, "pcq_set_option: null argument");
    if (!strcmp(key, "blocks_per_cu")) {
        if (value < 1 || value > 32) return pcq_fail(PCQ_ERR_ARG, "blocks_per_cu must be 1..32");
        ctx->grid_blocks_per_cu = (int)value;
#ifdef PCQ_LAB
        ctx->batch_blocks_per_cu = (int)value;
    } else if (!strcmp(key, "k1_variant")) {
        if (value < 0 || value > 14) return pcq_fail(PCQ_ERR_ARG, "k1_variant must be 0..14");
        ctx->k1_variant = (int)value;
    } else if (!strcmp(key, "class_batch_loads")) {
        if (value != 0 && value != 4 && value != 6 && value != 8 && value != 12) return pcq_fail(PCQ_ERR_ARG, "class_batch_loads must be 0, 4, 6, 8 or 12");
        ctx->class_batch_loads = (int)value;
    } else if (!strcmp(key, "class_batch_pipe")) {
        ctx->class_batch_pipe = value != 0;
    } else if (!strcmp(key, "class_batch_waves_per_cu")) {
        if (value < 1 || value > 32) return pcq_fail(PCQ_ERR_ARG, "class_batch_waves_per_cu must be 1..32");
        ctx->class_batch_waves_per_cu = (int)value;
    } else if (!strcmp(key, "k1_grid")) {
        if (value < 0 || value > (1 << 20)) return pcq_fail(PCQ_ERR_ARG, "k1_grid must be 0..2^20");
        ctx->k1_grid = (int)value;
    } else if (!strcmp(key, "k1_waves_per_cu")) {
        if (value < 1 || value > 32) return pcq_fail(PCQ_ERR_ARG, "k1_waves_per_cu must be 1..32");
        ctx->k1_waves_per_cu = (int)value;
    } else if (!strcmp(key, "batch_variant")) {
        if (value < 0 || value > 3) return pcq_fail(PCQ_ERR_ARG, "batch_variant must be 0..3");
        ctx->batch_variant = (int)value;
    } else if (!strcmp(key, "batch_waves_per_cu")) {
        if (value < 1 || value > 32) return pcq_fail(PCQ_ERR_ARG, "batch_waves_per_cu must be 1..32");
        ctx->batch_waves_per_cu = (int)value;
#endif
    } else if (!strcmp(key, "allreduce_single_rank")) {
        ctx->allreduce_single_rank = value != 0;
    } else if (!strcmp(key, "allreduce_fail")) {
        if (value < 0 || value > 3) return pcq_fail(PCQ_ERR_ARG, "allreduce_fail must be 0, 1, 2 or 3");
        ctx->allreduce_fail = (int)value;
    } else if (!strcmp(key, "grid_pending_budget")) {
        // (points scanned into a grid collector before it folds; a fold's tuple counts and offsets are 32-bit, grid_host.hip clamps to that)
        if (value < 0 || value > (int64_t)1 << 40) return pcq_fail(PCQ_ERR_ARG, "grid_pending_budget must be 0..2^40");
        ctx->grid_pending_budget = value;
    } else if (!strcmp(key, "grid_agg")) {
        if (value < 0 || value > 2) return pcq_fail(PCQ_ERR_ARG, "grid_agg must be 0 (adaptive), 1 (always) or 2 (never)");
        ctx->grid_agg = (int)value;
    } else if (!strcmp(key, "host_in_place")) {
        if (value < 0 || value > 2) return pcq_fail(PCQ_ERR_ARG, "host_in_place must be 0 (never), 1 (always) or 2 (until the copy path is set up)");
        ctx->host_in_place = (int)value;
    } else if (!strcmp(key, "emit_park_max")) {
        if (value < 0 || value > 256) return pcq_fail(PCQ_ERR_ARG, "emit_park_max must be 0..256");
        ctx->emit_park_max = (int)value;
    } else if (!strcmp(key, "emit_sparse_max")) {
        if (value < 0 || value > 2048) return pcq_fail(PCQ_ERR_ARG, "emit_sparse_max must be 0..2048");
        ctx->emit_sparse_max = (int)value;
    } else if (!strcmp(key, "grid_block_pad")) {
        if (value < 0 || value > 65536) return pcq_fail(PCQ_ERR_ARG, "grid_block_pad must be 0..65536");
        ctx->grid_block_pad = (int)value;
    } else if (!strcmp(key, "grid_stream")) {
        if (value < 0 || value > 1) return pcq_fail(PCQ_ERR_ARG, "grid_stream must be 0 or 1");
        ctx->grid_stream = (int)value;
    } else if (!strcmp(key, "grid_tuple16")) {
        if (value < 0 || value > 2) return pcq_fail(PCQ_ERR_ARG, "grid_tuple16 must be 0, 1 or 2");
        ctx->grid_tuple16 = (int)value;
    } else if (!strcmp(key, "grid_f2")) {
        if (value < 0 || value > 4096) return pcq_fail(PCQ_ERR_ARG, "grid_f2 must be 0..4096");
        ctx->grid_f2 = (int)value;
    } else if (!strcmp(key, "numa_local")) {
        ctx->numa_local = value != 0;
        delete ctx->copy_pool;  // helpers are re-created with or without the affinity
        ctx->copy_pool = nullptr;
        if (ctx->stage_bytes) {  // and the staging buffers re-allocated on the next scan
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ctx->copy_stream);
            for (int i = 0; i < 2; i++) {
                if (ctx->h_stage[i]) (void)hipHostFree(ctx->h_stage[i]);
                if (ctx->d_stage[i]) (void)hipFree(ctx->d_stage[i]);
                ctx->h_stage[i] = nullptr;
                ctx->d_stage[i] = nullptr;
            }
            ctx->stage_bytes = 0;
        }
    } else if (!strcmp(key, "copy_threads")) {
        if (value < 1 || value > 64) return pcq_fail(PCQ_ERR_ARG, "copy_threads must be 1..64");
        ctx->copy_threads = (int)value;
    } else if (!strcmp(key, "chunk_points")) {
        if (value < 4) return pcq_fail(PCQ_ERR_ARG, "chunk_points must be >= 4");
        ctx->chunk_points = (uint64_t)value;
    } else {
        return pcq_fail(PCQ_ERR_ARG, "unknown option '%s'", key);
    }
    return PCQ_OK;
}

extern "C" int pcq_get_option(pcq_ctx *ctx, const char *key, int64_t *value) {
    if (!ctx || !key || !value) return pcq_fail(PCQ_ERR_ARG, "pcq_get_option: null argument");
    if (!strcmp(key, "blocks_per_cu")) *value = ctx->grid_blocks_per_cu;
#ifdef PCQ_LAB
    else if (!strcmp(key, "k1_variant")) *value = ctx->k1_variant;
    else if (!strcmp(key, "k1_waves_per_cu")) *value = ctx->k1_waves_per_cu;
    else if (!strcmp(key, "batch_variant")) *value = ctx->batch_variant;
    else if (!strcmp(key, "batch_waves_per_cu")) *value = ctx->batch_waves_per_cu;
#endif
    else if (!strcmp(key, "chunk_points")) *value = (int64_t)ctx->chunk_points;
    else if (!strcmp(key, "copy_threads")) *value = ctx->copy_threads;
    else if (!strcmp(key, "numa_local")) *value = ctx->numa_local;
    else if (!strcmp(key, "numa_node")) *value = ctx->numa_node;
    else if (!strcmp(key, "grid_pending_budget")) *value = ctx->grid_pending_budget;
    else if (!strcmp(key, "allreduce_single_rank")) *value = ctx->allreduce_single_rank;
    else if (!strcmp(key, "allreduce_fail")) *value = ctx->allreduce_fail;
    else if (!strcmp(key, "grid_f2")) *value = ctx->grid_f2;
    else if (!strcmp(key, "grid_agg")) *value = ctx->grid_agg;
    else if (!strcmp(key, "grid_tuple16")) *value = ctx->grid_tuple16;
    else if (!strcmp(key, "grid_stream")) *value = ctx->grid_stream;
    else if (!strcmp(key, "grid_block_pad")) *value = ctx->grid_block_pad;
    else if (!strcmp(key, "host_in_place")) *value = ctx->host_in_place;
    else if (!strcmp(key, "emit_park_max")) *value = ctx->emit_park_max;
    else if (!strcmp(key, "emit_sparse_max")) *value = ctx->emit_sparse_max;
    else if (!strcmp(key, "grid_deferred")) *value = ctx->grid_deferred;
    else if (!strcmp(key, "grid_last_tuples")) *value = ctx->grid_last_tuples;
    else if (!strcmp(key, "grid_folds")) *value = ctx->grid_folds;
    else if (!strcmp(key, "grid_level2")) *value = ctx->grid_level2;
    else if (!strcmp(key, "grid_refolds")) *value = ctx->grid_refolds;
    else if (!strcmp(key, "grid_level2_exact")) *value = ctx->grid_level2_exact;
    else if (!strcmp(key, "grid_last_f2")) *value = ctx->grid_last_f2;
    else if (!strcmp(key, "grid_compactions")) *value = ctx->grid_compactions;
    else return pcq_fail(PCQ_ERR_ARG, "unknown option '%s'", key);
    return PCQ_OK;
}

// ---------------------------------------------------------------------------------------------
// device memory helpers
// ---------------------------------------------------------------------------------------------
extern "C" int pcq_device_alloc(pcq_ctx *ctx, uint64_t bytes, void **out) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx || !out) return pcq_fail(PCQ_ERR_ARG, "pcq_device_alloc: null argument");
    *out = nullptr;
    PCQ_HIP(hipSetDevice(ctx->device));
    PCQ_HIP(hipMalloc(out, bytes ? bytes : 16));
    return PCQ_OK;
}
extern "C" int pcq_device_free(pcq_ctx *ctx, void *p) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx) return pcq_fail(PCQ_ERR_ARG, "pcq_device_free: null context");
    if (p) PCQ_HIP(hipFree(p));
    return PCQ_OK;
}
extern "C" int pcq_copy_to_device(pcq_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx || (!dst && bytes) || (!src && bytes)) return pcq_fail(PCQ_ERR_ARG, "pcq_copy_to_device: null argument");
    if (bytes) PCQ_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return PCQ_OK;
}
extern "C" int pcq_copy_to_host(pcq_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx || (!dst && bytes) || (!src && bytes)) return pcq_fail(PCQ_ERR_ARG, "pcq_copy_to_host: null argument");
    if (bytes) {
        if (bytes <= 64 * sizeof(uint64_t) && bytes % 8 == 0 && ((uintptr_t)src & 7) == 0) {
            // A few words (a count): a kernel stores them into the context's pinned, device-visible scratch.  The first
            // device-to-host hipMemcpy of a process sets up the runtime's copy-engine path — 8 ms in the CLI, where this
            // read is the only one (profiles/r03_cli_e2e.log).
            hipLaunchKernelGGL(k_words_to_host, dim3(1), dim3(64), 0, ctx->stream, (const uint64_t *)src, ctx->h_scalars, (uint32_t)(bytes / 8));
            PCQ_HIP(hipGetLastError());
            PCQ_HIP(hipStreamSynchronize(ctx->stream));
            memcpy(dst, ctx->h_scalars, bytes);
            return PCQ_OK;
        }
        PCQ_HIP(hipStreamSynchronize(ctx->stream));
        PCQ_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    }
    return PCQ_OK;
}
extern "C" int pcq_device_memset(pcq_ctx *ctx, void *dst, int value, uint64_t bytes, void *stream) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx || (!dst && bytes)) return pcq_fail(PCQ_ERR_ARG, "pcq_device_memset: null argument");
    if (bytes) PCQ_HIP(hipMemsetAsync(dst, value, bytes, stream ? (hipStream_t)stream : ctx->stream));
    return PCQ_OK;
}

// ---------------------------------------------------------------------------------------------
// host-side math of the boundary
// ---------------------------------------------------------------------------------------------

// Rust `f64 as i64`: truncate toward zero, saturate, NaN -> 0.
static int64_t rust_f64_as_i64(double v) {
    if (std::isnan(v)) return 0;
    if (v >= 9223372036854775808.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}
static uint64_t rust_f64_as_u64(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)v;
}

// last.rs:98-109 / las.rs:88-99
extern "C" int pcq_box_to_local(const double bmin[3], const double bmax[3], const double scale[3],
                                const double offset[3], int64_t lmin[3], int64_t lmax[3]) {
    if (!bmin || !bmax || !scale || !offset || !lmin || !lmax) return pcq_fail(PCQ_ERR_ARG, "pcq_box_to_local: null argument");
    for (int a = 0; a < 3; a++) {
        lmin[a] = rust_f64_as_i64((bmin[a] - offset[a]) / scale[0]);  // sic: x scale on every axis (last.rs:100-102)
        lmax[a] = rust_f64_as_i64((bmax[a] - offset[a]) / scale[a]);
    }
    for (int a = 0; a < 3; a++)
        if (lmin[a] > lmax[a])
            return pcq_fail(PCQ_ERR_PANIC, "AABB::from_min_max: Minimum position must be <= maximum position!");
    return PCQ_OK;
}

int pcq_make_dev_pred(const pcq_predicate *p, DevPred *out) {
    memset(out, 0, sizeof *out);
    out->kind = p->kind;
    if (p->kind == PCQ_PRED_CLASS) {
        out->cls = p->cls;
        return PCQ_OK;
    }
    if (p->kind == PCQ_PRED_BOUNDS_F64) {
        for (int a = 0; a < 3; a++) out->wmin[a] = p->wmin[a], out->wmax[a] = p->wmax[a];
        return PCQ_OK;
    }
    if (p->kind != PCQ_PRED_BOUNDS) return pcq_fail(PCQ_ERR_ARG, "unknown predicate kind %d", p->kind);
    for (int a = 0; a < 3; a++) {
        const int64_t lo = p->lmin[a] < INT32_MIN ? (int64_t)INT32_MIN : p->lmin[a];
        const int64_t hi = p->lmax[a] > INT32_MAX ? (int64_t)INT32_MAX : p->lmax[a];
        if (lo > hi) {  // covers lmin > lmax as well as boxes outside the i32 value range
            out->empty = 1;
            out->lo[a] = 0;
            out->width[a] = 0;
        } else {
            out->lo[a] = (int32_t)lo;
            out->width[a] = (uint32_t)(hi - lo);
        }
    }
    return PCQ_OK;
}

// ---------------------------------------------------------------------------------------------
// collectors
// ---------------------------------------------------------------------------------------------
static int new_collector(pcq_ctx *ctx, int kind, pcq_collector **out) {
    if (!ctx || !out) return pcq_fail(PCQ_ERR_ARG, "collector: null argument");
    *out = nullptr;
    PCQ_HIP(hipSetDevice(ctx->device));  // the collector's memory belongs to the context's device, whatever the thread used before
    pcq_collector *c = new (std::nothrow) pcq_collector();
    if (!c) return pcq_fail(PCQ_ERR_NOMEM, "collector: out of memory");
    c->kind = kind;
    c->ctx = ctx;
    *out = c;
    return PCQ_OK;
}

extern "C" int pcq_collector_new_count(pcq_ctx *ctx, pcq_collector **out) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    int rc = new_collector(ctx, COLL_COUNT, out);
    if (rc) return rc;
    pcq_collector *c = *out;
    hipError_t e = hipMalloc((void **)&c->d_count, 16);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_count, 0, 16, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // scans may be enqueued on a caller's stream
    if (e != hipSuccess) {
        delete c;
        *out = nullptr;
        return pcq_fail(PCQ_ERR_HIP, "count collector: %s", hipGetErrorString(e));
    }
    c->owns_count = true;
    return PCQ_OK;
}

extern "C" int pcq_collector_new_count_at(pcq_ctx *ctx, uint64_t *device_counter, pcq_collector **out) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!device_counter) return pcq_fail(PCQ_ERR_ARG, "pcq_collector_new_count_at: null counter");
    int rc = new_collector(ctx, COLL_COUNT, out);
    if (rc) return rc;
    (*out)->d_count = device_counter;
    (*out)->owns_count = false;
    return PCQ_OK;
}

extern "C" int pcq_collector_new_buffer(pcq_ctx *ctx, pcq_collector **out) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    int rc = new_collector(ctx, COLL_BUFFER, out);
    if (rc) return rc;
    pcq_collector *c = *out;
    hipError_t e = hipMalloc((void **)&c->d_count, 16);  // the point count (see pcq_internal.h)
    if (e == hipSuccess) e = hipMemsetAsync(c->d_count, 0, 16, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // scans may be enqueued on a caller's stream
    if (e != hipSuccess) {
        delete c;
        *out = nullptr;
        return pcq_fail(PCQ_ERR_HIP, "buffer collector: %s", hipGetErrorString(e));
    }
    c->owns_count = true;
    return PCQ_OK;
}

// SparseGrid::new — grid_sampling.rs:18-47
extern "C" int pcq_collector_new_grid(pcq_ctx *ctx, const double bmin[3], const double bmax[3], double cell_size,
                                      pcq_collector **out) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!bmin || !bmax) return pcq_fail(PCQ_ERR_ARG, "pcq_collector_new_grid: null bounds");
    int rc = new_collector(ctx, COLL_GRID, out);
    if (rc) return rc;
    pcq_collector *c = *out;
    uint64_t bitsum = 0;
    for (int a = 0; a < 3; a++) {
        c->bmin[a] = bmin[a];
        c->bmax[a] = bmax[a];
        const double extent = bmax[a] - bmin[a];          // :19-23
        const double ncells = std::ceil(extent / cell_size);  // :24-28
        c->bits[a] = rust_f64_as_u64(std::ceil(std::log2(ncells)));  // :29-31
        c->dims[a] = rust_f64_as_u64(ncells);             // :39-43
        bitsum += c->bits[a];
    }
    c->cell_size = cell_size;
    if (bitsum > 64) {  // :32-34
        delete c;
        *out = nullptr;
        return pcq_fail(PCQ_ERR_GRID, "Too many cells ({}*{}*{}) in SparseGrid! The number of cells exceeds the capacity of a u64 index!");
    }
    if (bitsum == 64) {  // all-ones is a legal key then, which this table reserves as "empty"
        delete c;
        *out = nullptr;
        return pcq_fail(PCQ_ERR_UNSUPPORTED, "SparseGrid with exactly 64 key bits is not supported by the device hash table");
    }
    if (!std::isfinite(cell_size) || !std::isfinite(bmin[0]) || !std::isfinite(bmin[1]) || !std::isfinite(bmin[2]) ||
        !std::isfinite(bmax[0]) || !std::isfinite(bmax[1]) || !std::isfinite(bmax[2])) {
        delete c;
        *out = nullptr;
        return pcq_fail(PCQ_ERR_UNSUPPORTED, "SparseGrid with non-finite bounds or cell size is not supported");
    }
    DevGrid &g = c->grid;
    for (int a = 0; a < 3; a++) {
        g.bmin[a] = bmin[a];
        g.bmax[a] = bmax[a];
        g.dims_f[a] = (double)c->dims[a];
        g.inv_extent[a] = 1.0 / (bmax[a] - bmin[a]);
        g.qk[a] = g.dims_f[a] / (bmax[a] - bmin[a]);
        g.qmax[a] = g.dims_f[a] + 2.0 < 0x1p31 ? g.dims_f[a] + 2.0 : 0x1p31;  // every point inside the bounds is below dims + 1
        g.guard[a] = g.qmax[a] * 0x1p-50;
        g.mask[a] = (1ull << (c->bits[a] & 63)) - 1;  // Rust release `1u64 << n` masks n to 6 bits
    }
    g.cell_size = cell_size;
    g.shift[0] = 0;
    g.shift[1] = (uint32_t)(c->bits[0] & 63);
    g.shift[2] = (uint32_t)((c->bits[0] + c->bits[1]) & 63);
    g.keys_wide = bitsum > 32 ? 1u : 0u;
    return PCQ_OK;
}

extern "C" int pcq_collector_free(pcq_collector *c) {
    PCQ_ON_DEVICE_OF_COLLECTOR(c);
    if (!c) return PCQ_OK;
    if (c->ctx) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        if (c->last_stream && c->last_stream != c->ctx->stream) (void)hipStreamSynchronize(c->last_stream);  // scans enqueued on a caller's stream
    }
    if (c->owns_count && c->d_count) (void)hipFree(c->d_count);
    if (c->d_points && c->ctx) pcq_pool_free(c->ctx, c->d_points);
    if (c->kind == COLL_GRID) pcq_grid_release(c);
    delete c;
    return PCQ_OK;
}

extern "C" int pcq_collector_reset(pcq_collector *c) {
    PCQ_ON_DEVICE_OF_COLLECTOR(c);
    if (!c) return pcq_fail(PCQ_ERR_ARG, "pcq_collector_reset: null collector");
    hipStream_t s = c->ctx->stream;
    c->next_index = 0;
    // scans enqueued on a caller's stream may still be reading and moving the counters
    if (c->kind != COLL_GRID && c->last_stream && c->last_stream != s) PCQ_HIP(hipStreamSynchronize(c->last_stream));
    if (c->kind == COLL_COUNT) PCQ_HIP(hipMemsetAsync(c->d_count, 0, 8, s));
    if (c->kind == COLL_BUFFER) {
        PCQ_HIP(hipMemsetAsync(c->d_count, 0, 16, s));
        c->n_upper = 0;
        c->count_slot = 0;
    }
    if (c->kind == COLL_GRID) {
        PCQ_HIP(hipStreamSynchronize(s));
        if (c->last_stream && c->last_stream != s) PCQ_HIP(hipStreamSynchronize(c->last_stream));
        pcq_grid_release(c);
    }
    return PCQ_OK;
}

extern "C" int pcq_collector_has_points(const pcq_collector *c) { return c && c->kind != COLL_COUNT; }

extern "C" int pcq_collector_flush(pcq_collector *c) {
    PCQ_ON_DEVICE_OF_COLLECTOR(c);
    if (!c) return pcq_fail(PCQ_ERR_ARG, "pcq_collector_flush: null collector");
    pcq_ctx *ctx = c->ctx;
    if (c->last_stream && c->last_stream != ctx->stream) PCQ_HIP(hipStreamSynchronize(c->last_stream));
    if (c->kind == COLL_GRID) return pcq_grid_flush(c);  // (synchronises)
    PCQ_HIP(hipStreamSynchronize(ctx->stream));
    return PCQ_OK;
}

extern "C" int pcq_collector_point_count(pcq_collector *c, uint64_t *out) {
    PCQ_ON_DEVICE_OF_COLLECTOR(c);
    if (!c || !out) return pcq_fail(PCQ_ERR_ARG, "pcq_collector_point_count: null argument");
    pcq_ctx *ctx = c->ctx;
    if (c->last_stream && c->last_stream != ctx->stream) PCQ_HIP(hipStreamSynchronize(c->last_stream));
    switch (c->kind) {
    case COLL_COUNT:
        PCQ_HIP(hipMemcpyAsync(ctx->h_scalars, c->d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
        PCQ_HIP(hipStreamSynchronize(ctx->stream));
        *out = ctx->h_scalars[0];
        return PCQ_OK;
    case COLL_BUFFER:
        PCQ_HIP(hipMemcpyAsync(ctx->h_scalars, c->d_count + c->count_slot, 8, hipMemcpyDeviceToHost, ctx->stream));
        PCQ_HIP(hipStreamSynchronize(ctx->stream));
        *out = ctx->h_scalars[0];
        c->n_upper = ctx->h_scalars[0];
        return PCQ_OK;
    default:
        return pcq_grid_drain(c, nullptr, nullptr, 0, out);
    }
}

extern "C" int pcq_collector_points(pcq_collector *c, pcq_point *out, uint64_t cap, uint64_t *out_n) {
    PCQ_ON_DEVICE_OF_COLLECTOR(c);
    if (!c || !out_n) return pcq_fail(PCQ_ERR_ARG, "pcq_collector_points: null argument");
    pcq_ctx *ctx = c->ctx;
    *out_n = 0;
    if (c->kind == COLL_COUNT) return PCQ_OK;  // points() is None (collect_points.rs:87-93)
    if (c->last_stream && c->last_stream != ctx->stream) PCQ_HIP(hipStreamSynchronize(c->last_stream));
    if (c->kind == COLL_BUFFER) {
        PCQ_HIP(hipMemcpyAsync(ctx->h_scalars, c->d_count + c->count_slot, 8, hipMemcpyDeviceToHost, ctx->stream));
        PCQ_HIP(hipStreamSynchronize(ctx->stream));
        const uint64_t n_points = ctx->h_scalars[0];
        c->n_upper = n_points;
        *out_n = n_points;
        if (!out || n_points == 0) return PCQ_OK;
        if (cap < n_points)
            return pcq_fail(PCQ_ERR_CAPACITY, "buffer collector holds %llu points, capacity %llu",
                            (unsigned long long)n_points, (unsigned long long)cap);
        PCQ_HIP(hipMemcpy(out, c->d_points, n_points * 31, hipMemcpyDeviceToHost));
        return PCQ_OK;
    }
    return pcq_grid_drain(c, out, nullptr, cap, out_n);
}

extern "C" int pcq_collector_grid_cells(pcq_collector *c, uint64_t *out, uint64_t cap, uint64_t *out_n) {
    PCQ_ON_DEVICE_OF_COLLECTOR(c);
    if (!c || !out_n) return pcq_fail(PCQ_ERR_ARG, "pcq_collector_grid_cells: null argument");
    if (c->kind != COLL_GRID) return pcq_fail(PCQ_ERR_ARG, "pcq_collector_grid_cells: not a grid collector");
    return pcq_grid_drain(c, nullptr, out, cap, out_n);
}

extern "C" int pcq_collector_grid_params(const pcq_collector *c, uint64_t dims[3], uint64_t bits[3]) {
    if (!c || c->kind != COLL_GRID || !dims || !bits) return pcq_fail(PCQ_ERR_ARG, "pcq_collector_grid_params: not a grid collector");
    for (int a = 0; a < 3; a++) dims[a] = c->dims[a], bits[a] = c->bits[a];
    return PCQ_OK;
}

// ---------------------------------------------------------------------------------------------
// scan over device-resident columns
// ---------------------------------------------------------------------------------------------
static int validate_scan(const pcq_columns *cols, const pcq_predicate *pred, const pcq_collector *c) {
    if (!cols || !pred || !c) return pcq_fail(PCQ_ERR_ARG, "scan: null argument");
    if (pred->kind != PCQ_PRED_BOUNDS && pred->kind != PCQ_PRED_CLASS && pred->kind != PCQ_PRED_BOUNDS_F64)
        return pcq_fail(PCQ_ERR_ARG, "scan: bad predicate kind %d", pred->kind);
    if (cols->n == 0) return PCQ_OK;
    // index arithmetic (n * stride, first_index + n) must stay far from 2^64: a LAS record length is a u16
    // and 2^40 points is ~3 orders of magnitude beyond the largest dataset of the reference
    if (cols->n > (1ull << 40) || cols->first_index > (1ull << 62))
        return pcq_fail(PCQ_ERR_ARG, "scan: %llu points (first index %llu) is out of range", (unsigned long long)cols->n,
                        (unsigned long long)cols->first_index);
    if (cols->xyz_stride > 65535 || cols->cls_stride > 65535 || cols->rgb_stride > 65535)
        return pcq_fail(PCQ_ERR_ARG, "scan: column stride above 65535");
    const bool need_xyz = pred->kind != PCQ_PRED_CLASS || c->kind != COLL_COUNT;
    const bool need_cls = pred->kind == PCQ_PRED_CLASS || c->kind != COLL_COUNT;
    if (need_xyz && (!cols->xyz || cols->xyz_stride < 12)) return pcq_fail(PCQ_ERR_ARG, "scan: positions column missing or stride < 12");
    if (need_cls && (!cols->cls || cols->cls_stride < 1)) return pcq_fail(PCQ_ERR_ARG, "scan: classification column missing");
    if (cols->rgb && cols->rgb_stride < 6) return pcq_fail(PCQ_ERR_ARG, "scan: colour stride < 6");
    return PCQ_OK;
}

static DevCols to_dev_cols(const pcq_columns *cols) {
    DevCols d;
    d.xyz = (const uint8_t *)cols->xyz;
    d.cls = (const uint8_t *)cols->cls;
    d.rgb = (const uint8_t *)cols->rgb;
    d.xyz_stride = cols->xyz_stride;
    d.cls_stride = cols->cls_stride;
    d.rgb_stride = cols->rgb_stride;
    d.n = cols->n;
    d.first_index = cols->first_index;
    for (int a = 0; a < 3; a++) d.scale[a] = cols->scale[a], d.offset[a] = cols->offset[a];
    return d;
}

// Count of matches into *d_count (+=), choosing the fast kernels where the layout allows.
static int count_into(pcq_ctx *ctx, const DevCols &dc, const DevPred &dp, uint64_t *d_count, hipStream_t s) {
    if (dc.n == 0) return PCQ_OK;
    if (dp.kind == PCQ_PRED_BOUNDS) {
        if (dp.empty) return PCQ_OK;
        if (dc.xyz_stride == 12 && ((uintptr_t)dc.xyz & 3) == 0) {
            // peel the (at most 3) points in front of the first 16-byte aligned point boundary
            uint64_t head = ((uintptr_t)dc.xyz & 15) / 4;  // 12*head == -addr (mod 16)
            if (head > dc.n) head = dc.n;
            if (head) {
                DevCols h = dc;
                h.n = head;
                int rc = pcq_launch_generic_count(ctx, h, dp, d_count, s);
                if (rc) return rc;
            }
            return pcq_launch_bounds_count_xyz12(ctx, dc.xyz + 12 * head, dc.n - head, dp, d_count, s);
        }
        return pcq_launch_generic_count(ctx, dc, dp, d_count, s);
    }
    if (dp.kind == PCQ_PRED_BOUNDS_F64) return pcq_launch_generic_count(ctx, dc, dp, d_count, s);
    if (dc.cls_stride == 1) return pcq_launch_class_count_u8(ctx, dc.cls, dc.n, (uint8_t)dp.cls, d_count, s);
    return pcq_launch_generic_count(ctx, dc, dp, d_count, s);
}

// Room for `incoming` more points.  The host knows only an upper bound of the points held (every scanned point may have
// matched); while that bound fits the buffer nothing is asked of the device.  When it does not, the true count is read
// (one synchronisation), and the buffer grows only if the truth needs it.
static int buffer_reserve(pcq_collector *c, uint64_t incoming, hipStream_t s) {
    if (c->n_upper + incoming <= c->cap_points) return PCQ_OK;
    pcq_ctx *ctx = c->ctx;
    if (c->last_stream && c->last_stream != s) PCQ_HIP(hipStreamSynchronize(c->last_stream));
    PCQ_HIP(hipMemcpyAsync(ctx->h_scalars, c->d_count + c->count_slot, 8, hipMemcpyDeviceToHost, s));
    PCQ_HIP(hipStreamSynchronize(s));
    const uint64_t have = ctx->h_scalars[0];
    c->n_upper = have;
    if (have + incoming <= c->cap_points) return PCQ_OK;
    uint64_t cap = 2 * c->cap_points;  // geometric growth, but never beyond what is asked for when that is more
    if (cap < have + incoming) cap = have + incoming;
    if (cap < 4096) cap = 4096;
    void *nb = nullptr;
    int rc = pcq_pool_alloc(ctx, cap * 31 + 16, &nb);
    if (rc) return rc;
    if (have) PCQ_HIP(hipMemcpyAsync(nb, c->d_points, have * 31, hipMemcpyDeviceToDevice, s));
    PCQ_HIP(hipStreamSynchronize(s));
    pcq_pool_free(ctx, c->d_points);
    c->d_points = (uint8_t *)nb;
    c->cap_points = cap;
    return PCQ_OK;
}

// The per-context scratch (partial counts, tile offsets, the grid's count table, the segment table) is shared by all
// scans of the context and ordered only by the stream they run on: when a scan arrives on a different stream than the
// previous one, the previous stream is drained first (one stream in flight per context).
int pcq_scratch_stream(pcq_ctx *ctx, hipStream_t s) {
    if (ctx->scratch_stream && ctx->scratch_stream != s) PCQ_HIP(hipStreamSynchronize(ctx->scratch_stream));
    ctx->scratch_stream = s;
    return PCQ_OK;
}

static int scan_dev_impl(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c, hipStream_t s) {
    int rc = validate_scan(cols, pred, c);
    if (rc) return rc;
    if (cols->n == 0) return PCQ_OK;
    rc = pcq_scratch_stream(ctx, s);
    if (rc) return rc;
    DevPred dp;
    rc = pcq_make_dev_pred(pred, &dp);
    if (rc) return rc;
    DevCols dc = to_dev_cols(cols);
    c->last_stream = s;
    switch (c->kind) {
    case COLL_COUNT:
        return count_into(ctx, dc, dp, c->d_count, s);
    case COLL_BUFFER: {
        if (dp.kind == PCQ_PRED_BOUNDS && dp.empty) return PCQ_OK;
        rc = buffer_reserve(c, dc.n, s);
        if (rc) return rc;
        rc = pcq_launch_emit_points(ctx, dc, dp, c->d_points, c->d_count + c->count_slot, c->d_count + (c->count_slot ^ 1), s);  // asynchronous: one pass, no count first
        if (rc) return rc;
        c->count_slot ^= 1;
        c->n_upper += dc.n;
        return PCQ_OK;
    }
    case COLL_GRID: {
        if (dp.kind == PCQ_PRED_BOUNDS && dp.empty) return PCQ_OK;
        return pcq_grid_scan(ctx, c, dc, dp, s);  // asynchronous: the matches are partitioned now and folded when a result is asked for
    }
    }
    return pcq_fail(PCQ_ERR_ARG, "scan: unknown collector kind");
}

extern "C" int pcq_scan_dev(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c, void *stream) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx) return pcq_fail(PCQ_ERR_ARG, "pcq_scan_dev: null context");
    return scan_dev_impl(ctx, cols, pred, c, stream ? (hipStream_t)stream : ctx->stream);
}

// ---------------------------------------------------------------------------------------------
// scan over host-resident columns: pinned double buffers + hipMemcpyAsync overlapped with kernels
// ---------------------------------------------------------------------------------------------
// The staging ring: two pinned host buffers and their device twins.  `upto` = how many of the pairs the caller needs NOW: a
// scan asks for the first pair, issues its first chunk, and only then for the second — pinning 24 MB is 5 ms
// (profiles/r03_hip_startup.log: 11 ms for the ring), and the second pair's 5 ms then run under the first chunk's transfer
// instead of in front of it (the first file of a process cost 15-24 ms where the others cost 1: profiles/r03_cli_e2e.log).
static void join_stage_warm(pcq_ctx *ctx) {
    if (ctx->stage_warm.joinable()) ctx->stage_warm.join();
}
// The first LARGE host-to-device copy of a process takes 8 ms inside the call (the runtime sets its copy path up; the later
// ones take microseconds; a copy of 8 bytes does not do it: profiles/r04_cli_first_file.log).  start_copy_warm() spends them
// on a thread of its own — one pinned megabyte through hipMemcpyAsync on the copy stream — while the context's first scan
// reads its chunks in place; whoever uses the copy stream next joins it first.
static void join_copy_warm(pcq_ctx *ctx) {
    if (ctx->copy_warm.joinable()) ctx->copy_warm.join();
}
static void start_copy_warm(pcq_ctx *ctx) {
    int expected = 0;
    if (!ctx->copy_warm_state.compare_exchange_strong(expected, 1)) return;
    ctx->copy_warm = std::thread([ctx] {
        (void)hipSetDevice(ctx->device);
        const size_t bytes = 1u << 20;
        if (hipHostMalloc(&ctx->copy_warm_h, bytes, hipHostMallocDefault) == hipSuccess && hipMalloc(&ctx->copy_warm_d, bytes) == hipSuccess &&
            hipMemcpyAsync(ctx->copy_warm_d, ctx->copy_warm_h, bytes, hipMemcpyHostToDevice, ctx->copy_stream) == hipSuccess)
            (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipGetLastError();
        ctx->copy_warm_state.store(2);
    });
}
static int ensure_stage_now(pcq_ctx *ctx, size_t bytes, int upto);
static int ensure_stage(pcq_ctx *ctx, size_t bytes, int upto = 2) {
    join_stage_warm(ctx);
    return ensure_stage_now(ctx, bytes, upto);
}
static int ensure_stage_now(pcq_ctx *ctx, size_t bytes, int upto) {
    if (ctx->stage_bytes < bytes && (ctx->h_stage[0] || ctx->h_stage[1])) {  // too small: drop what there is
        PCQ_HIP(hipStreamSynchronize(ctx->stream));
        PCQ_HIP(hipStreamSynchronize(ctx->copy_stream));
        ctx->stage_busy[0] = ctx->stage_busy[1] = false;
        for (int i = 0; i < 2; i++) {
            if (ctx->h_stage[i]) PCQ_HIP(hipHostFree(ctx->h_stage[i]));
            if (ctx->d_stage[i]) PCQ_HIP(hipFree(ctx->d_stage[i]));
            ctx->h_stage[i] = nullptr;
            ctx->d_stage[i] = nullptr;
        }
        ctx->stage_bytes = 0;
    }
    if (ctx->stage_bytes < bytes) ctx->stage_bytes = bytes;  // (the size the pairs are allocated with from here on)
    bool need = false;
    for (int i = 0; i < upto; i++) need |= !ctx->h_stage[i];
    if (!need) return PCQ_OK;
    const auto t0 = std::chrono::steady_clock::now();
    // pinned pages are allocated where the allocating thread runs (default "local" policy): run on the GPU's node for it
    cpu_set_t saved;
    const bool rebind = ctx->numa_local && ctx->numa_node >= 0 && sched_getaffinity(0, sizeof saved, &saved) == 0 &&
                        sched_setaffinity(0, sizeof ctx->node_cpus, &ctx->node_cpus) == 0;
    hipError_t e = hipSuccess;
    for (int i = 0; i < upto && e == hipSuccess; i++) {
        if (ctx->h_stage[i]) continue;
        e = hipHostMalloc((void **)&ctx->h_stage[i], ctx->stage_bytes, hipHostMallocDefault);  // (pinned = resident: the pages exist when this returns)
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_stage[i], ctx->stage_bytes);
    }
    if (rebind) (void)sched_setaffinity(0, sizeof saved, &saved);
    if (e != hipSuccess) return pcq_fail(PCQ_ERR_HIP, "staging allocation failed: %s", hipGetErrorString(e));
    static const bool timing = getenv("PCQ_TIMING") && getenv("PCQ_TIMING")[0] == '1';
    if (timing)
        fprintf(stderr, "[pcq] staging pair(s) up to %d of %zu MB pinned + device in %.1f ms\n", upto, ctx->stage_bytes >> 20,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return PCQ_OK;
}

static int fetch(pcq_ctx *ctx, int fd, uint8_t *dst, const uint8_t *src, size_t bytes);

int pcq_stream_fd_to_device(pcq_ctx *ctx, int fd, uint64_t offset, uint64_t bytes, uint8_t *d_dst) {
    const size_t chunk = 32u << 20;
    int rc = ensure_stage(ctx, (bytes < chunk ? (size_t)bytes : chunk) + 64);
    if (rc) return rc;
    join_copy_warm(ctx);
    if (ctx->stage_busy[0] || ctx->stage_busy[1]) {  // a nowait scan may still be reading the staging buffers
        PCQ_HIP(hipStreamSynchronize(ctx->stream));
        ctx->stage_busy[0] = ctx->stage_busy[1] = false;
    }
    hipStream_t cs = ctx->copy_stream;
    const uint64_t nchunks = (bytes + chunk - 1) / chunk;
    // events reused as "staging buffer b has been copied out"
    for (uint64_t k = 0; k < nchunks; k++) {
        const int b = (int)(k & 1);
        const uint64_t at = k * chunk, len = bytes - at < chunk ? bytes - at : chunk;
        if (k >= 2) PCQ_HIP(hipEventSynchronize(ctx->copied[b]));
        rc = fetch(ctx, fd, ctx->h_stage[b], (const uint8_t *)(uintptr_t)(offset + at), (size_t)len);
        if (rc) {
            (void)hipStreamSynchronize(cs);
            return rc;
        }
        PCQ_HIP(hipMemcpyAsync(d_dst + at, ctx->h_stage[b], (size_t)len, hipMemcpyHostToDevice, cs));
        PCQ_HIP(hipEventRecord(ctx->copied[b], cs));
    }
    PCQ_HIP(hipStreamSynchronize(cs));
    return PCQ_OK;
}

extern "C" int pcq_read_fd_to_device(pcq_ctx *ctx, int fd, uint64_t file_offset, uint64_t bytes, void *d_dst) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx || fd < 0 || (!d_dst && bytes)) return pcq_fail(PCQ_ERR_ARG, "pcq_read_fd_to_device: bad argument");
    if (bytes == 0) return PCQ_OK;
    return pcq_stream_fd_to_device(ctx, fd, file_offset, bytes, (uint8_t *)d_dst);
}

struct StagePlan {
    bool aos;                  // LAS records: one interleaved range
    bool need_xyz, need_cls, need_rgb;
    uint64_t bytes_per_point;  // staged bytes per point (without per-region alignment slack)
    const uint8_t *aos_base;   // lowest needed column pointer of the first record
    uint64_t stride;           // aos stride
    uint64_t span;             // bytes from aos_base to the end of the last needed column of a record
};

static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }


// Copies `bytes` from the host source into pinned memory: memcpy from caller memory, or — when the
// columns are given as offsets into an open file (pcq_scan_fd) — pread straight from the page cache
// (no mmap page-table work: measured ~2x the rate of memcpy from a freshly mmapped file).
// The copy is split over the context's helper threads (copy_pool.h).
static void ensure_copy_pool(pcq_ctx *ctx) {
    if (!ctx->copy_pool || ctx->copy_pool->helpers() != ctx->copy_threads - 1) {
        delete ctx->copy_pool;
        ctx->copy_pool = new CopyPool(ctx->copy_threads - 1, ctx->numa_local && ctx->numa_node >= 0 ? &ctx->node_cpus : nullptr);
    }
}
extern "C" int pcq_prepare_host_scans(pcq_ctx *ctx) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx) return pcq_fail(PCQ_ERR_ARG, "pcq_prepare_host_scans: null context");
    if (ctx->stage_warm.joinable() || ctx->h_stage[0]) return PCQ_OK;  // under way, or nothing left to prepare
    ctx->stage_warm = std::thread([ctx] {
        (void)hipSetDevice(ctx->device);
        // (what a scan of positions + classes asks for: scan_host_impl.  BOTH pairs: with only the first one pinned here the scan pins
        // the second on a thread of its own while a third sets the copy path up, and its first launch waits for the two of them inside
        // the runtime — first file 14 -> 17.7 ms, profiles/r04_cli_first_file.log)
        (void)ensure_stage_now(ctx, (size_t)ctx->chunk_points * 12 + 4096, 2);
        ensure_copy_pool(ctx);
    });
    return PCQ_OK;
}
static int fetch(pcq_ctx *ctx, int fd, uint8_t *dst, const uint8_t *src, size_t bytes) {
    join_stage_warm(ctx);
    ensure_copy_pool(ctx);
    const int r = ctx->copy_pool->run(fd, dst, src, bytes);
    if (r < 0) return pcq_fail(PCQ_ERR_IO, "pread failed: %s", strerror(-r));
    if (r > 0) return pcq_fail(PCQ_ERR_EOF, "failed to fill whole buffer");
    return PCQ_OK;
}

static int scan_host_impl(pcq_ctx *ctx, int fd, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c, bool wait) {
    if (!ctx) return pcq_fail(PCQ_ERR_ARG, "pcq_scan_host: null context");
    int rc = validate_scan(cols, pred, c);
    if (rc) return rc;
    if (cols->n == 0) return PCQ_OK;
    PCQ_HIP(hipSetDevice(ctx->device));

    StagePlan pl{};
    pl.need_xyz = pred->kind != PCQ_PRED_CLASS || c->kind != COLL_COUNT;
    pl.need_cls = pred->kind == PCQ_PRED_CLASS || c->kind != COLL_COUNT;
    pl.need_rgb = c->kind != COLL_COUNT && cols->rgb != nullptr;
    const uint8_t *hx = (const uint8_t *)cols->xyz, *hc = (const uint8_t *)cols->cls, *hr = (const uint8_t *)cols->rgb;
    // AoS (LAS): every needed column has the same stride and lives inside one record
    {
        const uint64_t st = pl.need_xyz ? cols->xyz_stride : cols->cls_stride;
        bool same = st > 12 || (!pl.need_xyz && st > 1);
        if (pl.need_xyz && cols->xyz_stride != st) same = false;
        if (pl.need_cls && cols->cls_stride != st) same = false;
        if (pl.need_rgb && cols->rgb_stride != st) same = false;
        const uint8_t *lo = nullptr, *hi = nullptr;
        auto upd = [&](const uint8_t *p, uint64_t sz) {
            if (!lo || p < lo) lo = p;
            if (!hi || p + sz > hi) hi = p + sz;
        };
        if (pl.need_xyz) upd(hx, 12);
        if (pl.need_cls) upd(hc, 1);
        if (pl.need_rgb) upd(hr, 6);
        if (same && lo && (uint64_t)(hi - lo) <= st && st > 1) {
            pl.aos = true;
            pl.aos_base = lo;
            pl.stride = st;
            pl.span = (uint64_t)(hi - lo);
            pl.bytes_per_point = st;
        }
    }
    if (!pl.aos) {
        if ((pl.need_xyz && cols->xyz_stride != 12) || (pl.need_cls && cols->cls_stride != 1) ||
            (pl.need_rgb && cols->rgb_stride != 6))
            return pcq_fail(PCQ_ERR_ARG, "pcq_scan_host: columns must be packed blocks (LAST) or one interleaved record (LAS)");
        pl.bytes_per_point = (pl.need_xyz ? 12 : 0) + (pl.need_cls ? 1 : 0) + (pl.need_rgb ? 6 : 0);
    }

    // "chunk_points" is given in points of a positions column (12 B each); what matters to the pipeline is the BYTES per
    // chunk, so a class-only scan (1 B per point) takes 12 x as many points per chunk and a record scan of a wide LAS
    // format fewer — otherwise a class query would move 2 MB per chunk and drown in per-chunk overhead
    uint64_t chunk = ctx->chunk_points * 12 / (pl.bytes_per_point ? pl.bytes_per_point : 1);
    if (chunk < 4) chunk = 4;
    if (chunk > cols->n) chunk = cols->n;
    // keep each staging buffer <= 512 MiB
    const uint64_t max_stage = 512ull << 20;
    if (chunk * pl.bytes_per_point > max_stage) chunk = max_stage / pl.bytes_per_point;
    if (chunk < 1) chunk = 1;
    chunk = (chunk + 3) & ~3ull;  // multiples of 4 points keep 12-byte blocks 16-byte aligned per chunk
    const size_t stage_need = (size_t)(chunk * pl.bytes_per_point) + 64;
    static const bool timing = getenv("PCQ_TIMING") && getenv("PCQ_TIMING")[0] == '1';
    const bool first_scan = timing && !ctx->scanned_before;
    ctx->scanned_before = true;
    const auto t_scan = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (first_scan) fprintf(stderr, "[pcq] first scan of the context: %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scan).count());
    };
    rc = ensure_stage(ctx, stage_need, 1);  // (the second pair: behind the first chunk, below)
    if (rc) return rc;
    stamp("first staging pair ready");

    hipStream_t s = ctx->stream, cs = ctx->copy_stream;
    const uint64_t nchunks = (cols->n + chunk - 1) / chunk;

    // region offsets inside a staging buffer (SoA case)
    const size_t off_xyz = 0;
    const size_t off_cls = pl.need_xyz ? align16((size_t)chunk * 12) : 0;
    const size_t off_rgb = off_cls + (pl.need_cls ? align16((size_t)chunk) : 0);

    // A scan that reads every byte ONCE — count and grid collectors — reads the pinned ring in place: the kernels stream host
    // memory over PCIe at the rate the copy engine moves it (59 against 54.5 GB/s for a count, 56.5 against 50.1 for a grid
    // scan: profiles/r04_zero_copy.log), the chunk is not written to and read from HBM in between, and the process never sets
    // up its copy path (8 ms inside the first large hipMemcpyAsync: profiles/r04_cli_first_file.log) — but on a stream of files the
    // copy engine is a tenth faster (median file of 240 MB: 5.5 against 6.6 ms).  So, by default (host_in_place 2), the scans of
    // a context read in place WHILE a thread sets the copy path up, and copy from then on.  The buffer collector reads the
    // positions twice (count pass, emit pass): it always keeps the device twin.
    bool in_place = false;
    if (c->kind != COLL_BUFFER && ctx->host_in_place == 1) in_place = true;
    if (c->kind != COLL_BUFFER && ctx->host_in_place == 2 && ctx->copy_warm_state.load() != 2) {
        in_place = true;
        start_copy_warm(ctx);  // (the NEXT scan copies: 5.5 ms per 240 MB file against 6.6 in place, once the copy path exists)
    }
    if (!in_place) join_copy_warm(ctx);  // (nobody else is on the copy stream)
    auto stage = [&](uint64_t k) -> int {
        const int b = (int)(k & 1);
        const uint64_t first = k * chunk;
        const uint64_t cnt = cols->n - first < chunk ? cols->n - first : chunk;
        if (ctx->stage_busy[b]) {  // the kernels of the chunk that used staging pair b last (this call's or an earlier nowait call's) are done with it
            PCQ_HIP(hipEventSynchronize(ctx->consumed[b]));
            ctx->stage_busy[b] = false;
        }
        uint8_t *h = ctx->h_stage[b];
        size_t bytes;
        if (pl.aos) {
            // up to the last needed byte of the last record (never past the caller's mapping)
            bytes = (size_t)((cnt - 1) * pl.stride + pl.span);
            int frc = fetch(ctx, fd, h, pl.aos_base + first * pl.stride, bytes);
            if (frc) return frc;
        } else {
            int frc = PCQ_OK;
            if (pl.need_xyz) frc = fetch(ctx, fd, h + off_xyz, hx + first * 12, (size_t)cnt * 12);
            if (!frc && pl.need_cls) frc = fetch(ctx, fd, h + off_cls, hc + first, (size_t)cnt);
            if (!frc && pl.need_rgb) frc = fetch(ctx, fd, h + off_rgb, hr + first * 6, (size_t)cnt * 6);
            if (frc) return frc;
            bytes = off_rgb + (pl.need_rgb ? (size_t)cnt * 6 : 0);
            if (!pl.need_rgb) bytes = off_cls + (pl.need_cls ? (size_t)cnt : 0);
            if (!pl.need_cls && !pl.need_rgb) bytes = (size_t)cnt * 12;
        }
        if (k == 0) stamp("first chunk read into the staging buffer");
        if (in_place) return PCQ_OK;  // (the kernels read it where it is)
        PCQ_HIP(hipMemcpyAsync(ctx->d_stage[b], h, bytes, hipMemcpyHostToDevice, cs));
        PCQ_HIP(hipEventRecord(ctx->copied[b], cs));
        if (k == 0) stamp("first transfer issued");
        return PCQ_OK;
    };

    // on any failure the copies and kernels already queued must drain before the staging buffers (or the
    // caller's memory) can be touched again
    // every error exit below goes through fail(): queued copies and kernels drain before the staging buffers (or the caller's
    // memory) can be touched again, and no staging pair stays marked busy
#define PCQ_HIP_OR_FAIL(expr)                                                                                             \
    do {                                                                                                                  \
        hipError_t _e = (expr);                                                                                           \
        if (_e != hipSuccess) return fail(pcq_fail(PCQ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__)); \
    } while (0)
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(cs);
        (void)hipStreamSynchronize(s);
        ctx->stage_busy[0] = ctx->stage_busy[1] = false;
        return code;
    };
    // The second pair is pinned by a thread of this call WHILE the first chunk is read into the first pair (4 ms each, the
    // first scan of a context only; joined before anything else happens).
    int rc2 = PCQ_OK;
    std::thread second_pair;
    if (nchunks > 1 && !ctx->h_stage[1])
        second_pair = std::thread([&] {
            (void)hipSetDevice(ctx->device);
            rc2 = ensure_stage(ctx, stage_need, 2);
        });
    rc = stage(0);
    if (second_pair.joinable()) second_pair.join();
    if (rc) return fail(rc);
    if (rc2) return fail(pcq_fail(PCQ_ERR_HIP, "staging allocation failed (second pair)"));
    stamp("first chunk read and its transfer issued, second staging pair ready");
    if (nchunks > 1) {
        rc = ensure_stage(ctx, stage_need, 2);  // (no-op unless the pair above was not asked for)
        if (rc) return fail(rc);
    }
    for (uint64_t k = 0; k < nchunks; k++) {
        const int b = (int)(k & 1);
        const uint64_t first = k * chunk;
        const uint64_t cnt = cols->n - first < chunk ? cols->n - first : chunk;
        if (!in_place) PCQ_HIP_OR_FAIL(hipStreamWaitEvent(s, ctx->copied[b], 0));
        pcq_columns dcols = *cols;
        const uint8_t *d = in_place ? ctx->h_stage[b] : ctx->d_stage[b];
        if (pl.aos) {
            dcols.xyz = pl.need_xyz ? d + (hx - pl.aos_base) : nullptr;
            dcols.cls = pl.need_cls ? d + (hc - pl.aos_base) : nullptr;
            dcols.rgb = pl.need_rgb ? d + (hr - pl.aos_base) : nullptr;
        } else {
            dcols.xyz = pl.need_xyz ? d + off_xyz : nullptr;
            dcols.cls = pl.need_cls ? d + off_cls : nullptr;
            dcols.rgb = pl.need_rgb ? d + off_rgb : nullptr;
        }
        dcols.n = cnt;
        dcols.first_index = cols->first_index + first;
        rc = scan_dev_impl(ctx, &dcols, pred, c, s);
        if (rc) return fail(rc);
        PCQ_HIP_OR_FAIL(hipEventRecord(ctx->consumed[b], s));
        ctx->stage_busy[b] = true;
        if (k == 0) stamp("first chunk's kernels launched");
        if (k + 1 < nchunks) {  // the next chunk is read while this one's kernels run (in place: while they read this one over PCIe)
            rc = stage(k + 1);
            if (rc) return fail(rc);
        }
    }
    if (wait) {
        PCQ_HIP_OR_FAIL(hipStreamSynchronize(s));
        ctx->stage_busy[0] = ctx->stage_busy[1] = false;
    }
    stamp(wait ? "last chunk done" : "last chunk's kernels launched (not waited for)");
    return PCQ_OK;
#undef PCQ_HIP_OR_FAIL
}

extern "C" int pcq_scan_host(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    return scan_host_impl(ctx, -1, cols, pred, c, true);
}

extern "C" int pcq_scan_host_nowait(pcq_ctx *ctx, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    return scan_host_impl(ctx, -1, cols, pred, c, false);
}

extern "C" int pcq_scan_fd(pcq_ctx *ctx, int fd, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (fd < 0) return pcq_fail(PCQ_ERR_ARG, "pcq_scan_fd: bad file descriptor");
    return scan_host_impl(ctx, fd, cols, pred, c, true);
}

extern "C" int pcq_scan_fd_nowait(pcq_ctx *ctx, int fd, const pcq_columns *cols, const pcq_predicate *pred, pcq_collector *c) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (fd < 0) return pcq_fail(PCQ_ERR_ARG, "pcq_scan_fd_nowait: bad file descriptor");
    return scan_host_impl(ctx, fd, cols, pred, c, false);
}
