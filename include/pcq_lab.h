/*
 * pcq_lab.h — entry points that exist only in libpcq_lab.so (make -C adhoc-queries-pointclouds_amd/csrc lab): the product's
 * sources plus csrc/lab/, for the measurement tools in tools/.  Nothing here is part of the drop-in boundary (pcq.h).
 *
 * libpcq_lab.so also accepts the experiment options the product rejects: "k1_variant" (bounds-count kernel shape
 * 0..14), "k1_waves_per_cu", "k1_grid", "batch_variant" (0..3), "batch_waves_per_cu", "class_batch_loads",
 * "class_batch_waves_per_cu", "class_batch_pipe" — the shapes measured on the way to the shipped kernels
 * (profiles/r01_k1_*.log, r01_k2_sweep.log).
 *
 * (Round 2's experimental shapes of the grid collector — option "grid_variant" — left with the kernels they varied:
 * round 3 replaced pass 0 and the way the fold reads it; their measurements are in profiles/r02_grid_progress.txt.)
 */
#ifndef PCQ_LAB_H
#define PCQ_LAB_H

#include "pcq.h"

#ifdef __cplusplus
extern "C" {
#endif

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
