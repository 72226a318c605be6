/*
 * vofe_oracle.h -- CPU restatement of the feature front end (TEST INFRASTRUCTURE ONLY).
 *
 * This is the *oracle*: a CPU restatement of the OpenCV 4.10 algorithms that the reference
 * reaches through cv2.goodFeaturesToTrack (reference continuous.py:380-385,
 * initialization.py:41-42) and cv2.calcOpticalFlowPyrLK (continuous.py:145-146, 150-151,
 * initialization.py:51-52).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The product path (visual-odometry_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the arithmetic lives in the third-party wheel opencv-python==4.10.0.84
 * (reference vamr_proj.yaml:77) which is absent from the image and from /root/reference, and
 * the reference holds no tests, golden vectors or fixtures.  The restatement follows the
 * published algorithm as written down in SURVEY.md Appendix A; it is pinned only by analytic
 * known-answer tests and by an independent numpy twin (oracle/np_twin.py).
 *
 * All entry points take host pointers and return 0 on success, negative on bad arguments.
 */
#ifndef VOFE_ORACLE_H
#define VOFE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Selects one of the alternative arithmetics for the points SURVEY Appendix A marks as cv2-build dependent (all 0 =
 * canonical, what the HIP kernels follow).  Process-wide, not thread-safe: set between calls.  Used only to measure
 * how far the plausible cv2 arithmetics are from the canonical one (tests/test_variants_cpu.py).
 *   lk_sum      A.5 steps 4/6: 0 exact int64 | 1 f32 raster | 2,3 4 lanes (w/o, with FMA) | 4,5 128-bit universal
 *               intrinsics loop (8 px per step, pair dot products for b) | 6,7 8 lanes (w/o, with FMA)
 *   sobel       A.1 step 2: 0 canonical | 1 FMA in the smoothing passes | 2 integer Sobel scaled afterwards | 3 smoothing
 *               row pass accumulated tap by tap (OpenCV's generic 8u -> 32f RowFilter)
 *   mineig_fma  A.1 step 5;  box_running  A.1 step 4 (running f64 sums);  harris  0 | 1 scalar double-k | 2 vector f32
 *   thresh_ipp  A.2 step 2: v < thr + FLT_EPSILON -> 0 */
int vofe_cpu_set_variant(int lk_sum, int sobel, int mineig_fma, int box_running, int harris, int thresh_ipp);

/* 1: the tracker's window loops run eight pixels at a time (AVX2); same canonical arithmetic, bit-identical results (checked by
 * tests/test_oracle_cpu.py).  For the timed CPU baseline of bench.py; the scalar loops stay the checker's definition. */
int vofe_cpu_set_simd(int on);

/* one pyrDown step, u8 -> u8, dst is ((w+1)/2, (h+1)/2). */
int vofe_cpu_pyr_down_u8(const uint8_t* src, int h, int w, int spitch, uint8_t* dst, int dpitch);

/* Scharr derivative, u8 -> int16 interleaved (Ix,Iy), REFLECT_101, dpitch in int16 units. */
int vofe_cpu_scharr_s16x2(const uint8_t* src, int h, int w, int spitch, int16_t* dst, int dpitch);

/* number of levels buildOpticalFlowPyramid produces (returned maxLevel + 1),
 * and their sizes (level_w/level_h hold up to max_level+1 entries). */
int vofe_cpu_pyramid_levels(int h, int w, int win_w, int win_h, int max_level, int* level_w, int* level_h);

/* Unpadded pyramid level `level` of img (for stage-by-stage parity); dst pitch = level width. */
int vofe_cpu_pyramid_level(const uint8_t* img, int h, int w, int pitch, int level, uint8_t* dst);

/* cv2.calcOpticalFlowPyrLK.  pts are (n,2) f32 xy.  next_pts is read only when
 * flags has VOFE_USE_INITIAL_FLOW (4).  err entries that OpenCV leaves uninitialised are 0. */
int vofe_cpu_lk_track(const uint8_t* prev_img, const uint8_t* next_img, int h, int w, int pitch,
                      const float* prev_pts, float* next_pts, uint8_t* status, float* err, int n,
                      int win_w, int win_h, int max_level, int crit_type, int max_count, double eps,
                      int flags, double min_eig_thr, int nthreads);

/* Twin of ContinuousVO._track_keypoints (reference continuous.py:131-164): forward LK, backward
 * LK on all forward outputs, status &= max|prev-back| < fb_thresh (backward status ignored). */
int vofe_cpu_track_fb(const uint8_t* prev_img, const uint8_t* next_img, int h, int w, int pitch,
                      const float* prev_pts, float* next_pts, uint8_t* status, float* err, int n,
                      int win_w, int win_h, int max_level, int crit_type, int max_count, double eps,
                      double min_eig_thr, float fb_thresh, int nthreads);

/* cornerMinEigenVal (harris=0) / cornerHarris (harris=1) score map, blockSize 3,
 * Sobel aperture 3, u8 -> f32 (pitch = w floats). */
int vofe_cpu_corner_score(const uint8_t* img, int h, int w, int pitch, int harris, double k,
                          float* score, int nthreads);

/* The same for any blockSize in 1..31 and Sobel aperture (gradientSize) in {1, 3, 5, 7}: OpenCV's getSobelKernels taps, scale
 * 1 / (2^(aperture-1) * blockSize * 255) on the smoothing kernel, row pass tap by tap, symmetric / antisymmetric column
 * pass, blockSize x blockSize box sum in f64 with anchor blockSize / 2 (operation order stated at corner_score_generic in the
 * .cpp).  Block 3 / aperture 3 takes the dedicated path above; vofe_cpu_corner_score_generic forces the generic one (tests: both
 * must agree bit for bit). */
int vofe_cpu_corner_score_ex(const uint8_t* img, int h, int w, int pitch, int block_size, int aperture, int harris,
                             double k, float* score, int nthreads);
int vofe_cpu_corner_score_generic(const uint8_t* img, int h, int w, int pitch, int block_size, int aperture, int harris,
                                  double k, float* score, int nthreads);

/* cv2.goodFeaturesToTrack.  corners_xy holds up to max_corners (x,y) pairs when
 * max_corners > 0, else up to h*w pairs.  *n_out = number of corners (0 == cv2 returning None).
 * mask may be NULL. */
int vofe_cpu_gftt(const uint8_t* img, int h, int w, int pitch, int max_corners, double quality,
                  double min_distance, const uint8_t* mask, int harris, double k,
                  float* corners_xy, int capacity, int* n_out, int nthreads);
/* ... with blockSize / gradientSize as cv2.goodFeaturesToTrack takes them (same ranges as vofe_cpu_corner_score_ex). */
int vofe_cpu_gftt_ex(const uint8_t* img, int h, int w, int pitch, int max_corners, double quality,
                     double min_distance, const uint8_t* mask, int block_size, int aperture, int harris, double k,
                     float* corners_xy, int capacity, int* n_out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
