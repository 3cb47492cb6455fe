/* fuzz_loader.c -- CPU-only robustness driver for the .mars loader / launch planner.
 *
 * Built by tests/test_sanitize.py with  gcc -fsanitize=address,undefined  together with csrc/host/*.c (the HIP side
 * comes from the ordinary libnna_mars.so; without a GPU mars_load_memory plans the whole graph on the host, then
 * refuses with MARS_ERR_NNA_INIT_FAILED -- everything a hostile file can reach before the device is covered).
 * The reference loader trusts every count and offset (reference src/mars/mars_runtime.c:172-200, 220); this build
 * validates, so no input may crash, hang or trip a sanitizer.
 *
 *   fuzz_loader <iterations> <seed> file.mars [file.mars ...]
 * For each file: load it as is, then `iterations` mutated copies (random bytes / interesting integers written into
 * the header, tensor and layer tables; truncations).  Prints a summary; exit code 0 unless a load misbehaves
 * (returns MARS_OK without a device, or hands back a model pointer on failure).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mars_runtime.h"

static uint64_t rng_state;
static uint32_t rnd(void) {
    rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return (uint32_t)(rng_state >> 33);
}

static const uint32_t k_interesting[] = {0, 1, 2, 3, 4, 7, 15, 16, 17, 31, 32, 63, 64, 65, 255, 256, 1000, 4096, 65535, 65536,
                                         65537, 0x4000001u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0xfffffff0u, 0x10000000u};

static int try_load(const uint8_t *buf, size_t n) {
    mars_model_t *m = NULL;
    mars_error_t e = mars_load_memory(buf, n, &m);
    if (e == MARS_OK || m != NULL) {
        if (m) mars_free(m);
        return e == MARS_OK ? 1 : 2; /* 1: loaded (only possible with a GPU), 2: model returned with an error */
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s iterations seed file.mars...\n", argv[0]);
        return 64;
    }
    const long iters = atol(argv[1]);
    rng_state = strtoull(argv[2], NULL, 0) * 2 + 1;
    long loads = 0, bad = 0;
    for (int a = 3; a < argc; a++) {
        FILE *fp = fopen(argv[a], "rb");
        if (!fp) { perror(argv[a]); return 66; }
        fseek(fp, 0, SEEK_END);
        long sz = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        uint8_t *orig = (uint8_t *)malloc((size_t)sz + 1), *buf = (uint8_t *)malloc((size_t)sz + 1);
        if (!orig || !buf || fread(orig, 1, (size_t)sz, fp) != (size_t)sz) return 66;
        fclose(fp);
        if (try_load(orig, (size_t)sz) == 2) bad++;
        loads++;
        mars_header_t h;
        memcpy(&h, orig, sizeof(h) < (size_t)sz ? sizeof(h) : (size_t)sz);
        size_t tables = sizeof(h) + (size_t)h.num_tensors * sizeof(mars_tensor_t) + (size_t)h.num_layers * sizeof(mars_layer_t);
        if (tables > (size_t)sz) tables = (size_t)sz;
        for (long it = 0; it < iters; it++) {
            memcpy(buf, orig, (size_t)sz);
            size_t n = (size_t)sz;
            const int nmut = 1 + (int)(rnd() % 4);
            for (int k = 0; k < nmut; k++) {
                const uint32_t how = rnd() % 8;
                /* most mutations land in the descriptor tables, 4-byte aligned (that is where the integers live) */
                size_t pos = (rnd() % (tables ? tables : 1)) & ~(size_t)3;
                if (how == 7) pos = rnd() % (n ? n : 1);
                if (pos + 4 > n) continue;
                uint32_t v;
                switch (how) {
                    case 0: case 1: case 2: v = k_interesting[rnd() % (sizeof(k_interesting) / 4)]; memcpy(buf + pos, &v, 4); break;
                    case 3: v = rnd(); memcpy(buf + pos, &v, 4); break;
                    case 4: memcpy(&v, buf + pos, 4); v += (rnd() % 9) - 4; memcpy(buf + pos, &v, 4); break;
                    case 5: buf[pos + rnd() % 4] ^= (uint8_t)(1u << (rnd() % 8)); break;
                    case 6: n = rnd() % (n + 1); break; /* truncation */
                    default: buf[pos] = (uint8_t)rnd(); break;
                }
            }
            if (try_load(buf, n) == 2) bad++;
            loads++;
        }
        free(orig);
        free(buf);
    }
    printf("fuzz_loader: %ld loads, %ld misbehaved\n", loads, bad);
    return bad ? 1 : 0;
}
