/*
 * o3_harness.c -- TEST SCAFFOLDING: oracle O3 = the reference's detection
 * tail.  parse_output() and nms() are `static` inside the reference's demo
 * program (reference src/mars/mars_yolo_test.c:80-130); this TU includes that
 * file where it lies with its main() renamed, and exports thin wrappers.
 */
#define main ref_yolo_demo_main
#include "mars_yolo_test.c" /* found via -I$(REF)/src/mars */
#undef main

/* det_t = { float x, y, w, h, conf; int cls; } = 24 bytes */
int ref_o3_sizeof_det(void) { return (int)sizeof(det_t); }

int ref_o3_parse_output(const int8_t *data, int npred, float scale, void *dets, int maxd) {
    return parse_output(data, npred, scale, (det_t *)dets, maxd);
}

int ref_o3_nms(void *dets, int n, float thresh) { return nms((det_t *)dets, n, thresh); }
