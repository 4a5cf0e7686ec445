/*
 * bvcf_bench.h — measurement hooks of libbvcf.so.  NOT part of the drop-in ABI (include/bvcf.h): only bench.py,
 * tools/ and the size-independent property tests call these.  They replace nothing in the reference; they run the
 * same kernel chain bvcf_submit launches (launch_chain in bvcf_core.hip) over blocks that are already resident in
 * HBM, and time it with HIP events on the launch streams.
 */
#ifndef BVCF_BENCH_H
#define BVCF_BENCH_H

#include "bvcf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Runs the kernel chain `iters` times back to back, step i on resident block i % n_blocks (each owning
 * BVCF_DEVICE_PAD bytes past its nbytes), leaving the results in device memory.  HIP events on the launch stream
 * give, per step, the time of the whole chain (chain_ms[i]) and of its dominant kernel (gt_ms[i]): k_gt on the
 * census path, k_stream / k_stream_gen on the streaming path, k_sites on the sites-only path.  counts receives {lines, alleles,
 * errs, class-map bytes, tasks} of the last step.  Returns after the last step has finished. */
int bvcf_bench_device(bvcf_ctx *ctx, const void *const *dblocks, const size_t *nbytes, int n_blocks, int iters,
                      float *chain_ms, float *gt_ms, uint64_t counts[5]);
/* the same with batch i on slot i % slots_in_use (0 = every slot of the ctx, which is what bvcf_bench_device does):
 * 1 times the chains strictly one after the other, 2 lets consecutive batches overlap as bvcf_submit would */
int bvcf_bench_device_slots(bvcf_ctx *ctx, const void *const *device_blocks, const size_t *nbytes, int n_blocks, int iters,
                            uint32_t slots_in_use, float *chain_ms, float *scan_ms, uint64_t counts[5]);

/* which kernel the streaming path will launch for the next batch: 0 = k_stream (made for the 4-byte sample grid),
 * 1 = k_stream_gen (any sample fields); the ctx picks it from the shape of the lines of the batch before
 * (-1: the ctx is not on the streaming path) */
int bvcf_bench_stream_kernel(const bvcf_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* BVCF_BENCH_H */
