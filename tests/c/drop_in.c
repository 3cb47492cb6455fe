/* A C caller written against the reference's headers only (nna.h, mars_runtime.h, nna_model.h), in the shape of
 * the reference's src/mars/mars_test.c:33-148 and examples/test_inference.c:142-238, linked against
 * libnna_mars.so: proves the drop-in C linkage.  Prints a checksum of the output the test compares with the
 * golden vector of the same model. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "mars_runtime.h"
#include "nna.h"
#include "nna_model.h"

static unsigned long long fnv(const unsigned char *p, size_t n) {
    unsigned long long h = 0xCBF29CE484222325ull;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    if (nna_init() != NNA_SUCCESS) { fprintf(stderr, "nna_init failed\n"); return 3; }
    mars_model_t *m = NULL;
    if (mars_load_file(argv[1], &m) != MARS_OK) return 4;
    mars_runtime_tensor_t *in = mars_get_input(m, 0);
    if (!in || !in->vaddr) return 5;
    for (size_t i = 0; i < in->alloc_size; i++) ((int8_t *)in->vaddr)[i] = (int8_t)(i % 127); /* mars_test.c:82-84 */
    if (mars_run(m) != MARS_OK) return 6;
    mars_runtime_tensor_t *out = mars_get_output(m, 0);
    if (!out || !out->vaddr) return 7;
    printf("mars %zu %016llx\n", out->alloc_size, fnv((const unsigned char *)out->vaddr, out->alloc_size));
    mars_free(m);

    nna_model_t *nm = nna_model_load(argv[1], NULL);
    if (!nm) return 8;
    nna_tensor_t *tin = nna_model_get_input(nm, 0);
    if (!tin) return 9;
    for (size_t i = 0; i < nna_tensor_bytes(tin); i++) ((int8_t *)nna_tensor_data(tin))[i] = (int8_t)(i % 127);
    if (nna_model_run(nm) != NNA_SUCCESS) return 10;
    const nna_tensor_t *tout = nna_model_get_output(nm, 0);
    if (!tout) return 11;
    printf("nna %zu %016llx\n", nna_tensor_bytes(tout), fnv((const unsigned char *)nna_tensor_data(tout), nna_tensor_bytes(tout)));
    nna_model_unload(nm);
    nna_deinit();
    return 0;
}
