/* multi_gpu.c -- the 8-GPU recipe of INTEGRATION.md section 4 as a plain C program: one process per GPU, frames
 * sharded, ONE collective (RCCL broadcast of the packed parameter arena from rank 0), no exchange in the forward pass.
 * It replaces, for a node of MI355X, what reference src/device.c:133-302 does for the one NNA of a camera: bring the
 * device(s) up and make the weights resident.
 *
 *   multi_gpu <model.mars> <ranks> <frames_per_rank>
 *
 * The parent forks `ranks` children BEFORE anything touches a GPU (a process that has initialised HIP must not fork
 * workers).  Rank r uses device r (MARS_HIP_DEVICE); rank 0 creates the ncclUniqueId and hands it to the others
 * through a pipe; rank 0 loads the file, the others only its descriptors (weight blob zeroed: what a remote rank
 * would be sent) with MARS_HIP_LOAD_DEFER_WEIGHTS; after the broadcast every rank runs its own shard of frames
 * (frame f of the job = rank * frames_per_rank + local index; frame content depends on f only) and reports an
 * FNV-1a checksum per frame to the parent, which prints them in job order:
 *     frame <f> rank <r> <checksum>
 * and exits 0 when every rank succeeded.  With ranks == 1 the RCCL group has one member (what a one-GPU box can run).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "mars_hip.h"
#include "nna.h"

static unsigned long long fnv(const unsigned char *p, size_t n) {
    unsigned long long h = 0xCBF29CE484222325ull;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}

/* frame f: the 32-bit LCG of SURVEY.md section 8d, seed 0x5EED0000 + f, top byte of every state */
static void fill_frame(int8_t *dst, size_t n, unsigned f) {
    uint32_t s = 0x5EED0000u + f;
    for (size_t i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        dst[i] = (int8_t)(s >> 24);
    }
}

static int rank_main(int rank, int ranks, int per_rank, const unsigned char *file, size_t size, int id_rd, int id_wr, int out_wr) {
    char dev[16];
    snprintf(dev, sizeof(dev), "%d", rank);
    setenv("MARS_HIP_DEVICE", dev, 1);
    if (nna_init() != NNA_SUCCESS) return 10;
    ncclUniqueId id;
    if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) return 11;
        for (int r = 1; r < ranks; r++)
            if (write(id_wr, &id, sizeof(id)) != (ssize_t)sizeof(id)) return 12;
    } else if (read(id_rd, &id, sizeof(id)) != (ssize_t)sizeof(id)) {
        return 12;
    }
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, ranks, id, rank) != ncclSuccess) return 13;

    mars_model_t *m = NULL;
    mars_error_t e;
    if (rank == 0) {
        e = mars_load_memory(file, size, &m);
    } else { /* descriptors only: zero the blob, as if only the tables had been sent */
        unsigned char *desc = (unsigned char *)malloc(size);
        if (!desc) return 14;
        memcpy(desc, file, size);
        uint64_t woff, wsz;
        memcpy(&woff, file + 28, 8);
        memcpy(&wsz, file + 36, 8);
        if (woff < size) memset(desc + woff, 0, wsz < size - woff ? wsz : size - woff);
        e = mars_hip_load_memory_ex(desc, size, MARS_HIP_LOAD_DEFER_WEIGHTS, &m);
        free(desc);
    }
    if (e != MARS_OK) return 15;
    if (mars_hip_set_batch(m, per_rank) != MARS_OK) return 16;
    size_t nbytes = 0;
    void *arena = mars_hip_param_arena(m, &nbytes);
    hipStream_t stream = (hipStream_t)mars_hip_stream();
    if (!arena || ncclBroadcast(arena, arena, nbytes, ncclUint8, 0, comm, stream) != ncclSuccess) return 17;
    if (mars_hip_sync() != MARS_OK) return 18;

    mars_runtime_tensor_t *in = mars_get_input(m, 0), *out = mars_get_output(m, 0);
    if (!in || !out || !in->vaddr || !out->vaddr) return 19;
    const size_t fin = in->alloc_size / (size_t)per_rank, fout = out->alloc_size / (size_t)per_rank;
    if (per_rank == 1) { /* a single frame reports the reference's working-buffer size: use the frame's own */
        return 20;
    }
    for (int k = 0; k < per_rank; k++) fill_frame((int8_t *)in->vaddr + (size_t)k * fin, fin, (unsigned)(rank * per_rank + k));
    if (mars_run(m) != MARS_OK) return 21;
    for (int k = 0; k < per_rank; k++) {
        char line[96];
        int n = snprintf(line, sizeof(line), "frame %d rank %d %016llx\n", rank * per_rank + k, rank,
                         fnv((const unsigned char *)out->vaddr + (size_t)k * fout, fout));
        if (write(out_wr, line, (size_t)n) != n) return 22;
    }
    mars_free(m);
    ncclCommDestroy(comm);
    nna_deinit();
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s model.mars ranks frames_per_rank(>= 2)\n", argv[0]);
        return 2;
    }
    const int ranks = atoi(argv[2]), per_rank = atoi(argv[3]);
    if (ranks < 1 || ranks > 8 || per_rank < 2) return 2;
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) return 3;
    fseek(fp, 0, SEEK_END);
    long size = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    unsigned char *file = (unsigned char *)malloc((size_t)size);
    if (!file || fread(file, 1, (size_t)size, fp) != (size_t)size) return 3;
    fclose(fp);
    int idp[2], outp[2];
    if (pipe(idp) || pipe(outp)) return 4;
    pid_t pids[8];
    for (int r = 0; r < ranks; r++) {
        pids[r] = fork();
        if (pids[r] < 0) return 5;
        if (pids[r] == 0) {
            close(outp[0]);
            _exit(rank_main(r, ranks, per_rank, file, (size_t)size, idp[0], idp[1], outp[1]));
        }
    }
    close(outp[1]);
    close(idp[1]);
    /* collect the lines, print them in job order */
    char *lines[8 * 4096];
    int nlines = 0;
    char buf[65536];
    size_t have = 0;
    ssize_t got;
    while ((got = read(outp[0], buf + have, sizeof(buf) - 1 - have)) > 0) have += (size_t)got;
    buf[have] = 0;
    for (char *p = strtok(buf, "\n"); p && nlines < 8 * 4096; p = strtok(NULL, "\n")) lines[nlines++] = p;
    int bad = 0;
    for (int r = 0; r < ranks; r++) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            fprintf(stderr, "rank %d failed (status %d)\n", r, WIFEXITED(st) ? WEXITSTATUS(st) : -1);
            bad = 1;
        }
    }
    for (int f = 0; f < ranks * per_rank; f++)
        for (int i = 0; i < nlines; i++) {
            int lf = -1;
            if (sscanf(lines[i], "frame %d", &lf) == 1 && lf == f) printf("%s\n", lines[i]);
        }
    if (nlines != ranks * per_rank) bad = 1;
    return bad;
}
