/*
 * nna.h -- device bring-up for the MI355X build of the thingino-accel C API.
 *
 * Entry points and return codes of reference include/nna.h:26-80; the body
 * that opened /dev/soc-nna and mapped ORAM/NNDMA (reference src/device.c:
 * 133-302) is replaced by HIP device selection + stream creation.
 * Device choice: env MARS_HIP_DEVICE, else LOCAL_RANK, else 0.
 */
#ifndef THINGINO_ACCEL_NNA_H
#define THINGINO_ACCEL_NNA_H

#include "nna_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Idempotent.  NNA_ERROR_DEVICE when no usable gfx950 GPU is visible,
 * NNA_ERROR_INIT when stream/context creation fails. */
int nna_init(void);

/* Frees every nna_malloc block still alive, destroys the stream. */
void nna_deinit(void);

/* NNA_ERROR_INVALID for NULL, NNA_ERROR_INIT before nna_init. */
int nna_get_hw_info(nna_hw_info_t *info);

int nna_is_ready(void);

/* Same string as the reference ("0.1.0-dev", src/device.c:396-398). */
const char *nna_get_version(void);

/* Single-threaded contract as in the reference: both are no-op successes. */
int nna_lock(void);
int nna_unlock(void);

#ifdef __cplusplus
}
#endif
#endif
