/*
 * nna_device.c -- nna_init()/nna_deinit() and friends on MI355X.
 *
 * Same entry points and return codes as reference src/device.c:133-443; what
 * that file does with /dev/mem, /dev/soc-nna, ORAM and NNDMA mappings becomes
 * HIP device selection and one stream (mhip_init).  The reference's global
 * singleton, no-lock threading contract is kept (device.c:105-131).
 */
#include <stdio.h>
#include <stdlib.h>

#include "../mhip.h"
#include "nna.h"
#include "nna_memory.h"

void nna_memory_release_all(void); /* nna_memory.c */

static int g_initialized = 0;

static int pick_device(void) {
    const char *e = getenv("MARS_HIP_DEVICE");
    if (e && *e) return atoi(e);
    e = getenv("LOCAL_RANK"); /* one process per GPU under torch.distributed.run */
    if (e && *e) return atoi(e);
    return 0;
}

int nna_init(void) {
    if (g_initialized) return NNA_SUCCESS; /* idempotent, device.c:134 */
    int rc = mhip_init(pick_device());
    if (rc != 0) {
        fprintf(stderr, "nna_init: %s\n", mhip_last_error());
        return rc == -3 ? NNA_ERROR_INIT : NNA_ERROR_DEVICE;
    }
    g_initialized = 1;
    return NNA_SUCCESS;
}

void nna_deinit(void) {
    if (!g_initialized) return;
    nna_memory_release_all();
    mhip_shutdown();
    g_initialized = 0;
}

int nna_get_hw_info(nna_hw_info_t *info) {
    if (info == NULL) return NNA_ERROR_INVALID;
    if (!g_initialized) return NNA_ERROR_INIT;
    int cus = 0, lds = 0, gfx = 0;
    size_t hbm = 0;
    if (mhip_device_info(&cus, &lds, &gfx, &hbm) != 0) return NNA_ERROR_DEVICE;
    info->oram_vbase = 0; /* LDS has no host mapping */
    info->oram_pbase = 0;
    info->oram_size = (uint32_t)lds;
    info->version = (uint32_t)gfx;
    return NNA_SUCCESS;
}

int nna_is_ready(void) { return g_initialized; }

const char *nna_get_version(void) { return "0.1.0-dev"; }

int nna_lock(void) { return NNA_SUCCESS; }
int nna_unlock(void) { return NNA_SUCCESS; }

/* Source compatibility with code that linked the reference's internal getters
 * (reference src/device_internal.h:13-31).  There is no fd, no /dev/mem and no
 * host-mapped arena on this platform. */
int nna_device_get_fd(void) { return g_initialized ? 0 : -1; }
int nna_device_get_memfd(void) { return -1; }
void *nna_device_get_oram(void) { return NULL; }
void *nna_device_get_nndma_io(void) { return NULL; }
void *nna_device_get_nndma_desram(void) { return NULL; }
void *nna_device_get_ddr(void) { return NULL; }
uint32_t nna_device_get_ddr_pbase(void) { return 0; }
