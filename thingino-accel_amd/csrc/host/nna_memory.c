/*
 * nna_memory.c -- nna_malloc & co. as pinned, device-mapped host memory.
 *
 * Contract of reference src/memory.c:76-274: NULL plus a stderr line before
 * nna_init, a tracking list so that nna_free of a foreign pointer only logs,
 * nna_memalign that ignores its alignment (blocks are page aligned anyway),
 * an ORAM "allocator" that only keeps statistics and hands out (void*)1, and
 * empty cache maintenance calls.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../mhip.h"
#include "nna.h"
#include "nna_memory.h"

typedef struct block {
    void *ptr;
    size_t size;
    struct block *next;
} block_t;

static block_t *g_blocks = NULL;
static struct { size_t total, used; int init; } g_oram;

void *nna_malloc(size_t size) {
    if (!nna_is_ready()) {
        fprintf(stderr, "nna_malloc: NNA not initialized\n");
        return NULL;
    }
    void *p = mhip_host_alloc(size);
    if (!p) {
        fprintf(stderr, "nna_malloc: pinned allocation of %zu bytes failed: %s\n", size, mhip_last_error());
        return NULL;
    }
    block_t *b = (block_t *)malloc(sizeof(*b));
    if (!b) {
        mhip_host_free(p);
        return NULL;
    }
    b->ptr = p;
    b->size = size;
    b->next = g_blocks;
    g_blocks = b;
    return p;
}

void *nna_memalign(size_t alignment, size_t size) {
    (void)alignment;
    return nna_malloc(size);
}

void *nna_calloc(size_t nmemb, size_t size) {
    size_t total = nmemb * size;
    void *p = nna_malloc(total);
    if (p) memset(p, 0, total);
    return p;
}

void nna_free(void *ptr) {
    if (!ptr) return;
    if (!nna_is_ready()) {
        fprintf(stderr, "nna_free: NNA not initialized\n");
        return;
    }
    for (block_t **pp = &g_blocks; *pp; pp = &(*pp)->next) {
        if ((*pp)->ptr == ptr) {
            block_t *b = *pp;
            *pp = b->next;
            mhip_host_free(b->ptr);
            free(b);
            return;
        }
    }
    fprintf(stderr, "nna_free: Pointer %p not found in allocation list\n", ptr);
}

void nna_memory_release_all(void) {
    while (g_blocks) {
        block_t *b = g_blocks;
        g_blocks = b->next;
        mhip_host_free(b->ptr);
        free(b);
    }
    g_oram.init = 0;
    g_oram.used = 0;
}

static int oram_init(void) {
    if (g_oram.init) return 0;
    nna_hw_info_t hw;
    if (nna_get_hw_info(&hw) != NNA_SUCCESS) return -1;
    g_oram.total = hw.oram_size;
    g_oram.used = 0;
    g_oram.init = 1;
    return 0;
}

void *nna_oram_malloc(size_t size) {
    if (oram_init() != 0) return NULL;
    size_t aligned = (size + 63) & ~(size_t)63;
    if (g_oram.used + aligned > g_oram.total) {
        fprintf(stderr, "nna_oram_malloc: Out of ORAM memory (requested %zu, available %zu)\n", aligned,
                g_oram.total - g_oram.used);
        return NULL;
    }
    g_oram.used += aligned;
    return (void *)0x1; /* accounting only, as in the reference (memory.c:222) */
}

void nna_oram_free(void *ptr) { (void)ptr; }

int nna_oram_get_stats(size_t *total, size_t *used, size_t *free_bytes) {
    if (oram_init() != 0) return NNA_ERROR_INIT;
    if (total) *total = g_oram.total;
    if (used) *used = g_oram.used;
    if (free_bytes) *free_bytes = g_oram.total - g_oram.used;
    return NNA_SUCCESS;
}

void nna_cache_flush(void *ptr, size_t size) { (void)ptr; (void)size; }
void nna_cache_invalidate(void *ptr, size_t size) { (void)ptr; (void)size; }
