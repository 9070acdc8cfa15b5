/*
 * pcq_synth.h — bench / test support, NOT part of the drop-in boundary.
 *
 * Fills device-resident LAST column blocks (positions N x {i32 x,y,z}; classification N x u8) with
 * the deterministic synthetic data of SURVEY.md §8(d), directly in HBM, so that bench.py can hold
 * the full synthetic ca13 dataset (2608 Mpoints, 31.3 GB of positions) resident without pushing it
 * through PCIe.  The generator is integer-only (counter-based splitmix64 + 64x64 multiply-high) and
 * is bit-identical to the host generator oracle/synth.c (checked by tests/test_synth.py), so the CPU
 * baseline and the GPU scan the same points.
 */
#ifndef PCQ_SYNTH_H
#define PCQ_SYNTH_H

#include "pcq.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PCQ_SYNTH_MAX_CLASSES 8
typedef struct pcq_synth_spec {
    uint64_t seed;
    uint64_t n;
    uint32_t format;            /* LAS point format 0..3 (header only; columns do not depend on it) */
    uint32_t n_classes;
    double scale[3];
    double offset[3];
    int32_t lo[3];              /* inclusive integer lower corner                      */
    uint32_t span[3];           /* number of distinct integer values per axis (>= 1)   */
    uint32_t zo_prob16;         /* P(z outlier) * 65536                                */
    int32_t zo_lo;              /* outlier z range                                     */
    uint32_t zo_span;
    uint32_t cls_cum16[PCQ_SYNTH_MAX_CLASSES]; /* cumulative thresholds over 65536    */
    uint8_t cls_val[PCQ_SYNTH_MAX_CLASSES];
} pcq_synth_spec;

/* Fills points [first, first+count) of the spec: d_xyz receives count*3 int32 (may be NULL),
 * d_cls receives count bytes (may be NULL).  Asynchronous on `stream` (NULL = context stream). */
int pcq_synth_fill_dev(pcq_ctx *ctx, const pcq_synth_spec *spec, uint64_t first, uint64_t count,
                       void *d_xyz, void *d_cls, void *stream);

/* Developer tool: a read-only streaming kernel over `bytes` of device memory (16-byte aligned) in one
 * of the access shapes the scan kernels use (0 = K1's 3 KiB wave tiles, 1/2/3 = 1/4/8 independent
 * 16-byte loads per lane, grid-stride), optionally non-temporal.  Asynchronous; time it with events.
 * Gives the measured read-stream ceiling the scan kernels are compared with (tools/hbm_read_ceiling.py). */
int pcq_membench_read(pcq_ctx *ctx, const void *d_buf, uint64_t bytes, int shape, int nontemporal,
                      int blocks_per_cu, void *stream);

/* Same, as wave tiles of `loads` x 1 KiB with an explicit launch geometry (threads per block, blocks). */
int pcq_membench_read_tiles(pcq_ctx *ctx, const void *d_buf, uint64_t bytes, int loads, int threads, int blocks,
                            void *stream);

/* Same 3 KiB wave tiles with an XCD-aware workgroup -> tile mapping (0 = K1's, 1 = XCD-contiguous inside
 * each grid-wide window, 2 = one contiguous eighth of the buffer per XCD); blocks % 8 == 0.
 * tools/xcd_mapping_sweep.py. */
int pcq_membench_read_xcd(pcq_ctx *ctx, const void *d_buf, uint64_t bytes, int mapping, int threads, int blocks,
                          void *stream);

#ifdef __cplusplus
}
#endif
#endif
