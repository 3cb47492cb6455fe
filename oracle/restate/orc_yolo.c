/*
 * orc_yolo.c -- CPU restatement of the detection tail.
 * TEST INFRASTRUCTURE (see orc.h).
 * Follows reference src/mars/mars_yolo_test.c:80-104 (decode) and :107-130
 * (exchange sort + greedy class-wise IoU suppression).
 */
#include <math.h>
#include <stdlib.h>

#include "orc.h"

#define ORC_NUM_CLASSES 80
#define ORC_ROW 85
#define ORC_CONF_MIN 0.25f

int orc_parse_output(const int8_t *pred, int npred, float scale, orc_det_t *dets, int maxd) {
    int n = 0;
    for (int r = 0; r < npred; r++) {
        if (n >= maxd) break; /* the cap is checked before each row (:82) */
        const int8_t *row = pred + (size_t)r * ORC_ROW;
        /* -(float)p[4] * scale : negate first, then scale (:84) */
        float obj = 1.0f / (1.0f + expf((-(float)row[4]) * scale));
        if (obj < ORC_CONF_MIN) continue;
        int arg = 0;
        float top = -1e9f;
        for (int c = 0; c < ORC_NUM_CLASSES; c++) {
            float s = (float)row[5 + c] * scale;
            if (s > top) { /* strict: first maximum wins */
                top = s;
                arg = c;
            }
        }
        float conf = obj / (1.0f + expf(-top));
        if (conf < ORC_CONF_MIN) continue;
        orc_det_t *d = &dets[n++];
        d->x = (float)row[0] * scale;
        d->y = (float)row[1] * scale;
        d->w = (float)row[2] * scale;
        d->h = (float)row[3] * scale;
        d->conf = conf;
        d->cls = arg;
    }
    return n;
}

int orc_nms(orc_det_t *d, int n, float thresh) {
    /* :108-110 -- selection by repeated exchange; NOT stable, and the exact
     * permutation among equal confidences is part of the contract */
    for (int i = 0; i + 1 < n; i++)
        for (int j = i + 1; j < n; j++)
            if (d[j].conf > d[i].conf) {
                orc_det_t t = d[i];
                d[i] = d[j];
                d[j] = t;
            }
    if (n <= 0) return 0;
    unsigned char *dead = (unsigned char *)calloc((size_t)n, 1);
    for (int i = 0; i < n; i++) {
        if (dead[i]) continue;
        const float ax1 = d[i].x - d[i].w / 2, ay1 = d[i].y - d[i].h / 2;
        const float ax2 = d[i].x + d[i].w / 2, ay2 = d[i].y + d[i].h / 2;
        const float aarea = d[i].w * d[i].h;
        for (int j = i + 1; j < n; j++) {
            if (dead[j] || d[j].cls != d[i].cls) continue;
            float x1 = fmaxf(ax1, d[j].x - d[j].w / 2);
            float y1 = fmaxf(ay1, d[j].y - d[j].h / 2);
            float x2 = fminf(ax2, d[j].x + d[j].w / 2);
            float y2 = fminf(ay2, d[j].y + d[j].h / 2);
            float iw = fmaxf(0, x2 - x1), ih = fmaxf(0, y2 - y1);
            float inter = iw * ih;
            float barea = d[j].w * d[j].h;
            float uni = aarea + barea; /* (wa*ha + wb*hb) - inter + 1e-6f, left to right */
            uni = uni - inter;
            uni = uni + 1e-6f;
            if (inter / uni > thresh) dead[j] = 1;
        }
    }
    int kept = 0;
    for (int i = 0; i < n; i++)
        if (!dead[i]) d[kept++] = d[i];
    free(dead);
    return kept;
}
