/*
 * stub_dev.c -- TEST SCAFFOLDING (not reference code, not product code).
 *
 * The reference executor (reference src/mars/mars_runtime.c:205-210) and its
 * test programs need five symbols from reference src/device.c, which cannot
 * run off-camera (/dev/mem, /dev/soc-nna).  This file provides them on top of
 * a zero-filled host block so the reference sources can be compiled where they
 * lie and used as the parity oracle.
 */
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define REF_DDR_BYTES (64u << 20) /* 8 MiB is what the executor assumes; slack
                                     because two shipped models write past it */
static uint8_t *g_ddr;
static uint8_t g_oram[384 * 1024];

int nna_init(void) {
    if (!g_ddr) g_ddr = (uint8_t *)aligned_alloc(4096, REF_DDR_BYTES);
    if (!g_ddr) return -1;
    memset(g_ddr, 0, REF_DDR_BYTES);
    return 0;
}
void nna_deinit(void) {}
void *nna_device_get_ddr(void) { return g_ddr; }
uint32_t nna_device_get_ddr_pbase(void) { return 0x06000000u; }
void *nna_device_get_oram(void) { return g_oram; }

/* The reference prints on every load step and every conv; tests silence it. */
static int g_saved_out = -1, g_saved_err = -1, g_quiet_users = 0;
static pthread_mutex_t g_quiet_lock = PTHREAD_MUTEX_INITIALIZER;
/* counted: the frames-parallel CPU baseline runs one model per thread; the first user redirects, the last restores */
void ref_quiet(int on) {
    pthread_mutex_lock(&g_quiet_lock);
    g_quiet_users += on ? 1 : -1;
    if ((on && g_quiet_users != 1) || (!on && g_quiet_users != 0)) {
        pthread_mutex_unlock(&g_quiet_lock);
        return;
    }
    fflush(stdout);
    fflush(stderr);
    if (on && g_saved_out < 0) {
        int nul = open("/dev/null", O_WRONLY);
        g_saved_out = dup(1);
        g_saved_err = dup(2);
        dup2(nul, 1);
        dup2(nul, 2);
        close(nul);
    } else if (!on && g_saved_out >= 0) {
        dup2(g_saved_out, 1);
        dup2(g_saved_err, 2);
        close(g_saved_out);
        close(g_saved_err);
        g_saved_out = g_saved_err = -1;
    }
    pthread_mutex_unlock(&g_quiet_lock);
}
