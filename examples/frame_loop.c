/* A C host program on the C ABI alone (no HIP, no Python): the per-frame loop of the reference's ContinuousVO
 * (continuous.py:73-114) with its image-point arrays resident on the GPU, driven through vofe_framestate_*.
 * The scene is a textured plane at depth Z in front of a camera that translates parallel to it, so every answer has a
 * ground truth: image points move by (-SX, -SY) pixels per frame, and every candidate the library promotes must
 * triangulate to depth Z.  The pose estimation (cv2.solvePnPRansac in the reference) is replaced by the known pose.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/frame_loop.c -Lvisual-odometry_amd -lvofe -Wl,-rpath,$PWD/visual-odometry_amd -lm -o examples/frame_loop
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vofe.h"

#define W 640
#define H 360
#define MARGIN 64
#define CW (W + 2 * MARGIN)
#define CH (H + 2 * MARGIN)
#define FRAMES 24
#define SX 3.0      /* (the reference drops a candidate that travels less than 2 px in a frame: continuous.py:400) */
#define SY 0.8
#define F 400.0
#define Z 10.0

static unsigned rng_state = 12345u;
static double rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return (rng_state >> 8) / 16777216.0; }

/* value noise: random lattices of three cell sizes, bilinearly interpolated and summed */
static void make_canvas(float* c) {
    const int cell[3] = {5, 13, 37};
    const double wgt[3] = {0.5, 0.3, 0.2};
    memset(c, 0, sizeof(float) * CW * CH);
    for (int o = 0; o < 3; ++o) {
        const int gw = CW / cell[o] + 2, gh = CH / cell[o] + 2;
        float* g = (float*)malloc(sizeof(float) * gw * gh);
        for (int i = 0; i < gw * gh; ++i) g[i] = (float)rnd();
        for (int y = 0; y < CH; ++y)
            for (int x = 0; x < CW; ++x) {
                const double fx = (double)x / cell[o], fy = (double)y / cell[o];
                const int ix = (int)fx, iy = (int)fy;
                const double ax = fx - ix, ay = fy - iy;
                const double v = (1 - ay) * ((1 - ax) * g[iy * gw + ix] + ax * g[iy * gw + ix + 1]) +
                                 ay * ((1 - ax) * g[(iy + 1) * gw + ix] + ax * g[(iy + 1) * gw + ix + 1]);
                c[y * CW + x] += (float)(wgt[o] * v);
            }
        free(g);
    }
    float lo = 1e9f, hi = -1e9f;
    for (int i = 0; i < CW * CH; ++i) { if (c[i] < lo) lo = c[i]; if (c[i] > hi) hi = c[i]; }
    for (int i = 0; i < CW * CH; ++i) c[i] = 255.f * (c[i] - lo) / (hi - lo);
}

/* frame t: the canvas seen through a window that has moved by (SX t, SY t) pixels */
static void make_frame(const float* c, int t, uint8_t* img) {
    const double ox = MARGIN + SX * t, oy = MARGIN + SY * t;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const double fx = x + ox, fy = y + oy;
            const int ix = (int)fx, iy = (int)fy;
            const double ax = fx - ix, ay = fy - iy;
            const double v = (1 - ay) * ((1 - ax) * c[iy * CW + ix] + ax * c[iy * CW + ix + 1]) +
                             ay * ((1 - ax) * c[(iy + 1) * CW + ix] + ax * c[(iy + 1) * CW + ix + 1]);
            img[y * W + x] = (uint8_t)(v + 0.5);
        }
}

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vofe_last_error()); return 1; } } while (0)

int main(void) {
    float* canvas = (float*)malloc(sizeof(float) * CW * CH);
    uint8_t* img = (uint8_t*)malloc((size_t)W * H);
    make_canvas(canvas);

    const double cx = 0.5 * (W - 1), cy = 0.5 * (H - 1);
    vofe_framestate_params p;
    memset(&p, 0, sizeof p);
    p.height = H; p.width = W;
    p.win_w = 21; p.win_h = 21; p.max_level = 3;                                  /* continuous.py:24-29 */
    p.crit_type = VOFE_TERM_CRITERIA_COUNT | VOFE_TERM_CRITERIA_EPS; p.max_count = 30; p.eps = 0.01; p.min_eig_thr = 1e-3;
    p.fb_thresh = 1.0f;
    p.max_corners = 500; p.quality_level = 0.01; p.min_distance = 10;             /* constants.py */
    p.thr_new_keypoints = 15; p.thr_triangulation = 15; p.thr_candidates_min = 2; p.thr_candidates_max = 100;
    p.thr_alpha_rad = 1.0 * 3.14159265358979323846 / 180.0;
    p.cap_landmarks = 4096; p.cap_candidates = 4096; p.first_is_f64 = 1;
    const double K[9] = {F, 0, cx, 0, F, cy, 0, 0, 1};
    memcpy(p.K, K, sizeof K);

    vofe_framestate* fs = NULL;
    CHECK(vofe_framestate_create(0, &p, &fs));

    /* the state after a bootstrap: a handful of landmarks on the plane, no candidates yet */
    enum { K0 = 60 };
    float P0[K0 * 2];
    double X0[K0 * 3];
    for (int i = 0; i < K0; ++i) {
        const double x = 40 + (W - 80) * rnd(), y = 40 + (H - 80) * rnd();
        P0[2 * i] = (float)x; P0[2 * i + 1] = (float)y;
        X0[3 * i] = (x - cx) * Z / F; X0[3 * i + 1] = (y - cy) * Z / F; X0[3 * i + 2] = Z;
    }
    make_frame(canvas, 0, img);
    CHECK(vofe_framestate_reset(fs, img, W, P0, X0, K0, NULL, NULL, NULL, 0));

    int promoted_total = 0, depth_bad = 0, flow_bad = 0, flow_n = 0;
    float* prev_xy = (float*)malloc(sizeof(float) * 2 * 4096);
    for (int t = 1; t < FRAMES; ++t) {
        make_frame(canvas, t, img);
        const float* xy; const uint8_t* st; const double* world;
        int k = 0, nf = 0, m1 = 0, kw = 0, prom = 0;
        CHECK(vofe_framestate_download(fs, prev_xy, NULL, NULL, NULL));             /* where the landmarks were (for the check only) */
        CHECK(vofe_framestate_track(fs, img, W, &xy, &st, &k, &nf, &m1));
        CHECK(vofe_framestate_landmarks_world(fs, &world, &kw));
        /* "pose estimation": the camera centre is known; every tracked landmark is an inlier */
        uint8_t keep[4096];
        int tracked = 0;
        for (int i = 0; i < k; ++i) {
            keep[i] = st[i] == 1;
            tracked += keep[i];
            if (keep[i]) {                                                         /* ground truth: (-SX, -SY) pixels per frame */
                ++flow_n;
                if (fabs((xy[2 * i] - prev_xy[2 * i]) + SX) > 0.3 || fabs((xy[2 * i + 1] - prev_xy[2 * i + 1]) + SY) > 0.3) ++flow_bad;
            }
        }
        const double pose[12] = {1, 0, 0, SX * t * Z / F, 0, 1, 0, SY * t * Z / F, 0, 0, 1, 0};
        CHECK(vofe_framestate_update(fs, pose, keep, &prom));
        int kn = 0, mn = 0;
        CHECK(vofe_framestate_counts(fs, &kn, &mn));
        if (prom) {                                                                /* promoted candidates must lie on the plane */
            CHECK(vofe_framestate_landmarks_world(fs, &world, &kw));
            for (int i = kw - prom; i < kw; ++i) if (fabs(world[3 * i + 2] - Z) > 0.1 * Z) ++depth_bad;
        }
        promoted_total += prom;
        printf("frame %2d: %3d landmarks, %3d tracked, %3d corners, %3d candidates kept -> %3d promoted; now %3d landmarks, %3d candidates\n",
               t, k, tracked, nf, m1, prom, kn, mn);
    }
    CHECK(vofe_framestate_destroy(fs));
    free(prev_xy); free(img); free(canvas);
    printf("promoted %d candidates, %d off the plane; %d of %d tracked flows off the ground truth\n", promoted_total, depth_bad, flow_bad, flow_n);
    if (promoted_total < 50 || depth_bad * 20 > promoted_total || flow_bad * 50 > flow_n) { printf("frame_loop FAILED\n"); return 2; }
    printf("frame_loop ok\n");
    return 0;
}
