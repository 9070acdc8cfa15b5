// alias_sort.hip — orders the tuples of aliased grid keys by (key, file order) for the exact replay of grid_finish.hip.
//
// The list is normally a handful of tuples (a point exactly on the grid's far face when dims is a power of two), and
// grid_finish.hip ranks those with a quadratic kernel.  Inputs that alias massively (a grid box much smaller than the data it
// is fed) make the list as long as the scan; for those the order comes from rocPRIM's radix sort — two stable passes
// over (file order, then key) — so the replay stays O(n log n) instead of O(n^2).  Not a hot path: library sort.
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "pcq_internal.h"

namespace {
__global__ __launch_bounds__(256) void k_iota(uint32_t *idx, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)i;
}
// keys_out[i] = field(items[idx[i]])
__global__ __launch_bounds__(256) void k_gather_u64(const uint8_t *items, size_t stride, size_t field_offset, const uint32_t *idx, uint64_t n,
                                                    uint64_t *keys_out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys_out[i] = *reinterpret_cast<const uint64_t *>(items + (size_t)idx[i] * stride + field_offset);
}
__global__ __launch_bounds__(256) void k_gather_items(const uint8_t *items, size_t stride, const uint32_t *idx, uint64_t n, uint8_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t *s = reinterpret_cast<const uint64_t *>(items + (size_t)idx[i] * stride);
    uint64_t *d = reinterpret_cast<uint64_t *>(out + i * stride);
    for (size_t k = 0; k < stride / 8; k++) d[k] = s[k];
}
}  // namespace

// items: n records of `stride` bytes (a multiple of 8) with a u64 key at offset 0 and a u64 order at offset 8.
// sorted[i] = the i-th record by (key, order).  n < 2^32.  Synchronous on `s` only through stream order.
int pcq_sort_by_key_then_order(pcq_ctx *ctx, const void *items, size_t stride, uint64_t n, void *sorted, hipStream_t s) {
    if (n == 0) return PCQ_OK;
    if (n >= (1ull << 32) || stride % 8) return pcq_fail(PCQ_ERR_ARG, "alias sort: bad arguments");
    void *p_keys = nullptr, *p_keys2 = nullptr, *p_idx = nullptr, *p_idx2 = nullptr, *p_tmp = nullptr;
    int rc = pcq_pool_alloc(ctx, n * 8, &p_keys);
    if (!rc) rc = pcq_pool_alloc(ctx, n * 8, &p_keys2);
    if (!rc) rc = pcq_pool_alloc(ctx, n * 4, &p_idx);
    if (!rc) rc = pcq_pool_alloc(ctx, n * 4, &p_idx2);
    size_t tmp_bytes = 0;
    hipError_t e = hipSuccess;
    if (!rc) {
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t *)p_keys, (uint64_t *)p_keys2, (uint32_t *)p_idx, (uint32_t *)p_idx2, n, 0, 64, s);
        if (e == hipSuccess) rc = pcq_pool_alloc(ctx, tmp_bytes ? tmp_bytes : 8, &p_tmp);
    }
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const uint8_t *it = (const uint8_t *)items;
    if (!rc && e == hipSuccess) {
        hipLaunchKernelGGL(k_iota, dim3(blocks), dim3(256), 0, s, (uint32_t *)p_idx, n);
        hipLaunchKernelGGL(k_gather_u64, dim3(blocks), dim3(256), 0, s, it, stride, (size_t)8, (const uint32_t *)p_idx, n, (uint64_t *)p_keys);
        e = rocprim::radix_sort_pairs(p_tmp, tmp_bytes, (uint64_t *)p_keys, (uint64_t *)p_keys2, (uint32_t *)p_idx, (uint32_t *)p_idx2, n, 0, 64, s);  // by file order
    }
    if (!rc && e == hipSuccess) {
        hipLaunchKernelGGL(k_gather_u64, dim3(blocks), dim3(256), 0, s, it, stride, (size_t)0, (const uint32_t *)p_idx2, n, (uint64_t *)p_keys);
        e = rocprim::radix_sort_pairs(p_tmp, tmp_bytes, (uint64_t *)p_keys, (uint64_t *)p_keys2, (uint32_t *)p_idx2, (uint32_t *)p_idx, n, 0, 64, s);  // stable, by key
    }
    if (!rc && e == hipSuccess) {
        hipLaunchKernelGGL(k_gather_items, dim3(blocks), dim3(256), 0, s, it, stride, (const uint32_t *)p_idx, n, (uint8_t *)sorted);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);  // the scratch goes back to the pool
    }
    for (void *p : {p_keys, p_keys2, p_idx, p_idx2, p_tmp}) pcq_pool_free(ctx, p);
    if (rc) return rc;
    if (e != hipSuccess) return pcq_fail(PCQ_ERR_HIP, "alias sort failed: %s", hipGetErrorString(e));
    return PCQ_OK;
}
