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

/* Image front-end (reference src/mars/mars_yolo_test.c:40-77): load_image() is static there too and reads a FILE
 * through stb_image; the caller hands this wrapper the path of a binary PPM it wrote, so the reference's own
 * decode -> stbir_resize_uint8 -> letterbox -> (px - 128) sequence runs unmodified.  Returns 0 on success. */
int ref_o3_load_image(const char *ppm_path, int tw, int th, int nhwc, int8_t *out, int *src_w, int *src_h) {
    int ow = 0, oh = 0;
    int8_t *img = load_image(ppm_path, tw, th, nhwc, &ow, &oh);
    if (!img) return -1;
    memcpy(out, img, (size_t)tw * th * 3);
    free(img);
    if (src_w) *src_w = ow;
    if (src_h) *src_h = oh;
    return 0;
}
