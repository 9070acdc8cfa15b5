// synth_kernels.hip — device-side synthetic LAST column generator (bench / test support; see
// include/pcq_synth.h).  Same integer arithmetic as the host generator used by the oracle side.
#include "pcq_internal.h"
#include "pcq_synth.h"

namespace {

struct DevSynth {
    uint64_t seed;
    int32_t lo[3];
    uint32_t span[3];
    uint32_t zo_prob16;
    int32_t zo_lo;
    uint32_t zo_span;
    uint32_t n_classes;
    uint32_t cls_cum16[PCQ_SYNTH_MAX_CLASSES];
    uint32_t cls_val[PCQ_SYNTH_MAX_CLASSES];
};

__device__ __forceinline__ uint64_t mix(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t mulhi(uint64_t h, uint32_t span) { return (uint32_t)__umul64hi(h, (uint64_t)span); }

__global__ __launch_bounds__(256) void k_synth_fill(DevSynth s, uint64_t first, uint64_t count, int32_t *__restrict__ xyz,
                                                    uint8_t *__restrict__ cls) {
    const uint64_t nthreads = (uint64_t)gridDim.x * 256;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < count; k += nthreads) {
        const uint64_t i = first + k;
        const uint64_t h3 = mix(s.seed, 8 * i + 3);
        if (xyz) {
            const uint64_t h0 = mix(s.seed, 8 * i + 0), h1 = mix(s.seed, 8 * i + 1), h2 = mix(s.seed, 8 * i + 2);
            const int32_t x = (int32_t)((int64_t)s.lo[0] + (int64_t)mulhi(h0, s.span[0]));
            const int32_t y = (int32_t)((int64_t)s.lo[1] + (int64_t)mulhi(h1, s.span[1]));
            int32_t z;
            if (((h3 >> 16) & 0xFFFF) < s.zo_prob16) z = (int32_t)((int64_t)s.zo_lo + (int64_t)mulhi(h2, s.zo_span));
            else z = (int32_t)((int64_t)s.lo[2] + (int64_t)mulhi(h2, s.span[2]));
            xyz[3 * k + 0] = x;
            xyz[3 * k + 1] = y;
            xyz[3 * k + 2] = z;
        }
        if (cls) {
            const uint32_t u = (uint32_t)(h3 & 0xFFFF);
            uint32_t v = 0;
            if (s.n_classes) {
                v = s.cls_val[s.n_classes - 1];
                for (int j = (int)s.n_classes - 1; j >= 0; j--)
                    if (u < s.cls_cum16[j]) v = s.cls_val[j];
            }
            cls[k] = (uint8_t)v;
        }
    }
}

}  // namespace

extern "C" int pcq_synth_fill_dev(pcq_ctx *ctx, const pcq_synth_spec *spec, uint64_t first, uint64_t count, void *d_xyz,
                                  void *d_cls, void *stream) {
    PCQ_ON_DEVICE_OF_CTX(ctx);
    if (!ctx || !spec) return pcq_fail(PCQ_ERR_ARG, "pcq_synth_fill_dev: null argument");
    if (spec->n_classes > PCQ_SYNTH_MAX_CLASSES) return pcq_fail(PCQ_ERR_ARG, "pcq_synth_fill_dev: too many classes");
    if (first > spec->n || count > spec->n - first) return pcq_fail(PCQ_ERR_ARG, "pcq_synth_fill_dev: range outside the spec");
    if (count == 0) return PCQ_OK;
    DevSynth s;
    s.seed = spec->seed;
    for (int a = 0; a < 3; a++) s.lo[a] = spec->lo[a], s.span[a] = spec->span[a];
    s.zo_prob16 = spec->zo_prob16;
    s.zo_lo = spec->zo_lo;
    s.zo_span = spec->zo_span;
    s.n_classes = spec->n_classes;
    for (int j = 0; j < PCQ_SYNTH_MAX_CLASSES; j++) s.cls_cum16[j] = spec->cls_cum16[j], s.cls_val[j] = spec->cls_val[j];
    uint64_t blocks = (count + 255) / 256;
    const uint64_t cap = (uint64_t)ctx->num_cus * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_synth_fill, dim3((unsigned)blocks), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream, s, first,
                       count, (int32_t *)d_xyz, (uint8_t *)d_cls);
    PCQ_HIP(hipGetLastError());
    return PCQ_OK;
}
