/*
 * nna_memory.h -- buffers the accelerator can reach.
 *
 * Reference include/nna_memory.h:28-119 / src/memory.c:76-274 hand out
 * ioctl-allocated DMA pages mapped through /dev/mem.  Here nna_malloc returns
 * pinned, device-mapped host memory (hipHostMalloc): the caller fills it with
 * plain stores and kernels / DMA engines read it without a staging copy.
 * Behaviour kept: NULL + a stderr line before nna_init (memory.c:77-81);
 * nna_memalign ignores `alignment` (memory.c:138-142; blocks are page aligned);
 * freeing an unknown pointer only logs (memory.c:195); the ORAM calls are an
 * accounting shim that returns the dummy pointer (void*)1 (memory.c:198-246).
 */
#ifndef THINGINO_ACCEL_NNA_MEMORY_H
#define THINGINO_ACCEL_NNA_MEMORY_H

#include "nna_types.h"

#ifdef __cplusplus
extern "C" {
#endif

void *nna_malloc(size_t size);
void *nna_memalign(size_t alignment, size_t size);
void *nna_calloc(size_t nmemb, size_t size);
void nna_free(void *ptr);

void *nna_oram_malloc(size_t size);
void nna_oram_free(void *ptr);
int nna_oram_get_stats(size_t *total, size_t *used, size_t *free_bytes);

/* Host caches are coherent with pinned memory on this platform: no-ops. */
void nna_cache_flush(void *ptr, size_t size);
void nna_cache_invalidate(void *ptr, size_t size);

#ifdef __cplusplus
}
#endif
#endif
