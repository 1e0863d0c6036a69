/* EXPERIMENTAL entry points: exported only by libqspec_hip_experimental.so (-DQS_EXPERIMENTAL), NOT part of the product ABI
 * (include/qspec_hip.h).  Measured and left out of the engine: DESIGN.md section 4, "Measured and rejected (round 2)". */
#pragma once
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Cache hint: read `bytes` at p (16-byte aligned) with `workgroups` x 256 threads and discard them, which leaves
 * the lines in the 256 MiB Infinity Cache.  Meant for a side stream, concurrently with the latency-bound kernels
 * between two GEMMs, so the next GEMM finds its weights on-die.  No reference counterpart; no effect on results. */
int qspec_prefetch(const void* p, size_t bytes, int workgroups, void* stream);
/* Tile-aligned form: workgroup b touches tiles b, b + workgroups, ... (tile_bytes each, from first_tile, ntiles of them)
 * -- the ranges the streaming GEMM's workgroup b reads -- so the lines wait in that workgroup's own XCD's L2. */
int qspec_prefetch_tiles(const void* p, size_t tile_bytes, int first_tile, int ntiles, int workgroups, void* stream);

#ifdef __cplusplus
}
#endif
