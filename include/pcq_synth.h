/*
 * pcq_synth.h — bench / test support, NOT part of the drop-in boundary.
 *
 * Fills device-resident LAST column blocks (positions N x {i32 x,y,z}; classification N x u8) with
 * the deterministic synthetic data of SURVEY.md §8(d), directly in HBM, so that bench.py can hold
 * the full synthetic ca13 dataset (2608 Mpoints, 31.3 GB of positions) resident without pushing it
 * through PCIe.  The generator is integer-only (counter-based splitmix64 + 64x64 multiply-high) and
 * is bit-identical to the host generator oracle/synth.c (checked by tests/test_gpu_scan.py::test_synth_device_generator_is_bit_identical), so the CPU
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

#ifdef __cplusplus
}
#endif
#endif
