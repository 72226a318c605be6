/*
 * vofe.h -- C ABI of the MI355X (gfx950) visual-odometry feature front end.
 *
 * Drop-in boundary for the ONE hot path of DiaHidvegi/visual-odometry: the three OpenCV calls
 * its Python reaches per frame.  Every entry point is `extern "C"`, takes plain pointers and
 * sizes (device pointers unless a parameter is named h_*), returns 0 on success or a negative
 * VOFE_E_* code, and never throws.  The message for the last failure on the calling thread is
 * returned by vofe_last_error().
 *
 * Ownership: the caller owns every image / point / status / err buffer (device memory, e.g.
 * torch.Tensor.data_ptr()); the library owns only the ctx workspace and vofe_pyr objects.
 * "Device pointer" means any address the device can access: for the few-KB point / status / err
 * arrays, pinned host memory mapped into the device (hipHostMalloc + hipHostGetDevicePointer) works
 * and saves the copy commands -- the shipped Python mirror passes its point sets that way.
 * Threading: a ctx is bound to one device and one stream and is not thread-safe; use one ctx
 * per thread / GPU.  Launches are asynchronous on the ctx stream unless stated otherwise.  Every
 * entry point that touches the device makes the ctx's device current for the calling thread
 * (hipSetDevice) and leaves it current.
 *
 * There is no CPU fallback in this library: without a usable gfx950 device vofe_create fails.
 */
#ifndef VOFE_H
#define VOFE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VOFE_OK 0
#define VOFE_E_ARG (-1)      /* cv2 would raise cv2.error (assert on arguments) */
#define VOFE_E_HIP (-2)      /* a HIP runtime call failed */
#define VOFE_E_UNSUPPORTED (-3)
#define VOFE_E_NOMEM (-4)

/* cv2 constants used by the reference (continuous.py:27-28, initialization.py:49-50) */
#define VOFE_TERM_CRITERIA_COUNT 1
#define VOFE_TERM_CRITERIA_EPS 2
#define VOFE_OPTFLOW_USE_INITIAL_FLOW 4
#define VOFE_OPTFLOW_LK_GET_MIN_EIGENVALS 8

typedef struct vofe_ctx vofe_ctx;
typedef struct vofe_pyr vofe_pyr;

const char* vofe_last_error(void);
int vofe_version(void);

/* Bind a context to `device` and to an existing HIP stream (NULL = the default stream). */
int vofe_create(int device, void* hip_stream, vofe_ctx** out);
int vofe_destroy(vofe_ctx* ctx);
int vofe_synchronize(vofe_ctx* ctx);
/* The VOFE_* environment switches (kernel-path overrides for tests and measurements: VOFE_LK, VOFE_PYRAMID,
 * VOFE_PAD_COPY, VOFE_PYR_BORDERS, VOFE_CORNER_SCORE, VOFE_SCORE_RS, VOFE_NMS_ROWS, VOFE_GFTT_SORT, VOFE_GFTT_CHAIN, VOFE_GFTT_DEBUG, VOFE_SEL_*) are read once, by
 * vofe_create; this re-reads them for an existing context (test aid; no counterpart in the reference). */
int vofe_reload_env(vofe_ctx* ctx);

/* ---- Gaussian pyramid + Scharr derivatives --------------------------------------------------
 * Replaces cv::buildOpticalFlowPyramid + cv::calcScharrDeriv, which cv2.calcOpticalFlowPyrLK runs
 * internally at reference continuous.py:145-146, 150-151 and initialization.py:51-52.
 * A vofe_pyr holds `slots` images of identical geometry (each slot: all levels, REFLECT_101
 * padded by the window, plus zero-padded int16x2 derivative planes when with_deriv != 0).
 * The number of levels follows OpenCV: it stops early when the next level would not exceed the
 * window. */
int vofe_pyramid_create(vofe_ctx* ctx, int slots, int height, int width, int win_w, int win_h,
                        int max_level, int with_deriv, vofe_pyr** out);
int vofe_pyramid_free(vofe_ctx* ctx, vofe_pyr* pyr);
/* Build `count` consecutive slots starting at slot0 from images d_img + i*img_stride (bytes),
 * rows `pitch` bytes apart. */
int vofe_pyramid_build(vofe_ctx* ctx, vofe_pyr* pyr, int slot0, int count, const uint8_t* d_img,
                       int pitch, size_t img_stride);
int vofe_pyramid_levels(const vofe_pyr* pyr);
int vofe_pyramid_level_size(const vofe_pyr* pyr, int level, int* width, int* height);
/* Synchronous copies of one level to host memory (parity tests): unpadded image (w*h bytes),
 * unpadded derivative (w*h*2 int16, may be NULL), or the image with `pad_x`/`pad_y` border
 * columns/rows ((w+2*pad_x)*(h+2*pad_y) bytes; pad_x <= win_w, pad_y <= win_h). */
int vofe_pyramid_download(vofe_ctx* ctx, const vofe_pyr* pyr, int slot, int level,
                          uint8_t* h_image, int16_t* h_deriv);
int vofe_pyramid_download_padded(vofe_ctx* ctx, const vofe_pyr* pyr, int slot, int level,
                                 int pad_x, int pad_y, uint8_t* h_image);

/* ---- pyramidal Lucas-Kanade ------------------------------------------------------------------
 * Replaces cv2.calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts, winSize, maxLevel,
 * criteria, flags, minEigThreshold) as called at reference continuous.py:145-146, 150-151 and
 * initialization.py:51-52.  prev/next pyramids must have the same geometry and window; maxLevel
 * is the pyramids' level count - 1.  Point i reads slot prev_slot + idx and next_slot + idx with
 * idx = d_img_idx ? d_img_idx[i] : 0, so one launch tracks a ragged batch of independent
 * sequences.  n_images bounds the index: slots prev_slot .. prev_slot + n_images - 1 (and the same
 * for next) must exist, which is checked; a point whose index lies outside [0, n_images) is reported
 * lost (status 0, err 0) and never addresses memory outside the batch.  n_images is ignored (1) when
 * d_img_idx is NULL.  d_prev_pts / d_next_pts are (n,2) f32 xy; d_next_pts is read only with
 * VOFE_OPTFLOW_USE_INITIAL_FLOW.  d_status (n) u8, d_err (n) f32 (entries cv2 leaves
 * uninitialised are written as 0).  Any window the pyramids were created for is tracked: up to 64
 * columns (and 64 rows per lane group) by the wavefront-per-keypoint kernels, larger ones by a
 * workgroup-per-keypoint kernel whose template window must fit 64 KiB of LDS (win_w * win_h <=
 * 10 880, about 104x104; VOFE_E_UNSUPPORTED beyond). */
int vofe_lk_track(vofe_ctx* ctx, const vofe_pyr* prev, int prev_slot, const vofe_pyr* next,
                  int next_slot, const float* d_prev_pts, const int* d_img_idx, int n_images, float* d_next_pts,
                  uint8_t* d_status, float* d_err, int n, int crit_type, int max_count, double eps,
                  int flags, double min_eig_thr);

/* Replaces ContinuousVO._track_keypoints (reference continuous.py:131-164) in one launch:
 * forward LK, backward LK from every forward output, status &= max|prev - back| < fb_thresh
 * (the backward status is ignored, as at continuous.py:150,158).  Both pyramids need derivatives.
 * A keypoint the forward pass loses is not tracked backward: its status is 0 whatever the backward
 * pass would find, and no backward result is returned. */
int vofe_track_fb(vofe_ctx* ctx, const vofe_pyr* prev, int prev_slot, const vofe_pyr* next,
                  int next_slot, const float* d_prev_pts, const int* d_img_idx, int n_images, float* d_next_pts,
                  uint8_t* d_status, float* d_err, int n, int crit_type, int max_count, double eps,
                  double min_eig_thr, float fb_thresh);

/* Measurement aid (no counterpart in the reference): when d_counters is non-NULL every later LK
 * launch of this ctx adds {points, point-levels entered, Gauss-Newton iterations} to the three
 * device uint64 words at d_counters (both passes of vofe_track_fb count).  NULL switches it off. */
int vofe_lk_set_counters(vofe_ctx* ctx, unsigned long long* d_counters);

/* Measurement aid (no counterpart in the reference): between vofe_profile_begin and vofe_profile_end every kernel this
 * context launches for the pyramid, Scharr, score-map and detector stages is bracketed by two HIP events on the
 * context's stream (the stream bench.py's roofline figures must be timed on).  vofe_profile_end synchronises the
 * stream and returns, in launch order, up to `capacity` kernel names (VOFE_PROFILE_NAME_LEN bytes each, NUL
 * terminated; pyrDown / per-level kernels carry " L<level>") and their durations in milliseconds. */
#define VOFE_PROFILE_NAME_LEN 48
int vofe_profile_begin(vofe_ctx* ctx);
int vofe_profile_end(vofe_ctx* ctx, int capacity, char* names, float* ms, int* n_out);

/* ---- corner detection --------------------------------------------------------------------------
 * vofe_corner_score replaces cv::cornerMinEigenVal / cv::cornerHarris that cv2.goodFeaturesToTrack
 * runs at reference continuous.py:380-385 and initialization.py:41-42 (there with cv2's defaults,
 * blockSize 3 and Sobel aperture 3, which have dedicated kernels).  block_size 1..31 and aperture
 * (cv2: gradientSize / ksize) 1, 3, 5 or 7 are implemented: VOFE_E_ARG for values cv2 rejects too
 * (block_size < 1, even or non-positive aperture), VOFE_E_UNSUPPORTED beyond the implemented range.
 * d_score is (batch, height, width) f32, contiguous. */
int vofe_corner_score(vofe_ctx* ctx, const uint8_t* d_img, int batch, int height, int width,
                      int pitch, size_t img_stride, int block_size, int aperture, int harris,
                      double k, float* d_score);

/* Replaces cv2.goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, mask, blockSize,
 * useHarrisDetector, k[, gradientSize]) (reference continuous.py:380-385, initialization.py:41-42);
 * block_size / aperture as for vofe_corner_score.
 * d_corners_xy receives min(*h_n_out, capacity) (x,y) f32 pairs, strongest first; *h_n_out (host) the
 * number of corners cv2 would return (0 == cv2 returning None): at most max_corners when that is
 * positive, every accepted corner otherwise -- it exceeds capacity whenever capacity is the smaller.
 * d_mask may be NULL (u8, height*width, contiguous).
 * Synchronous: returns after the corner count is known. */
int vofe_gftt(vofe_ctx* ctx, const uint8_t* d_img, int height, int width, int pitch,
              int max_corners, double quality, double min_distance, const uint8_t* d_mask,
              int block_size, int aperture, int harris, double k, float* d_corners_xy,
              int capacity, int* h_n_out);

/* vofe_gftt for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, set-up code that
 * runs around a HIP graph capture).  NO COUNTERPART IN THE REFERENCE.  Same arguments, same results as vofe_gftt, but the
 * detector's workspace (score map, candidate maps, bucket histograms, keys: 27 bytes per pixel + 129 KiB) is a static
 * device array inside the library instead of a buffer grown with hipMalloc, and the call itself allocates and frees
 * nothing (the array's address, the launch attribute of its last stage and the small pinned count header are set up by
 * vofe_create).  Like vofe_gftt it is SYNCHRONOUS (it waits on the context's stream for the corner count), so it cannot
 * be called INSIDE a stream capture; and with kernel timing switched on (vofe_profile_begin) the timing aid creates HIP
 * events per launch, which is outside this guarantee.
 * Limit: height * width <= VOFE_NOALLOC_MAX_PIXELS (covers the reference's 640x480, 1241x376 and 800x600 frames);
 * larger frames are rejected with VOFE_E_UNSUPPORTED.  The array (14.3 MB) is held by every GPU the library runs on
 * until the process ends, keeps the previous call's contents (the call clears what must start at zero) and is shared
 * by ALL contexts of a GPU: calls that overlap in time on one GPU race -- serialise them. */
#define VOFE_NOALLOC_MAX_PIXELS (512 * 1024)
int vofe_gftt_noalloc(vofe_ctx* ctx, const uint8_t* d_img, int height, int width, int pitch,
                      int max_corners, double quality, double min_distance, const uint8_t* d_mask,
                      int block_size, int aperture, int harris, double k, float* d_corners_xy,
                      int capacity, int* h_n_out);

/* Same call with the corner list delivered to host memory as cv2.goodFeaturesToTrack delivers it
 * (reference continuous.py:380-385 consumes it as a numpy array): h_corners_xy (host, capacity (x,y)
 * pairs) receives min(*h_n_out, capacity) corners together with the count, in the call's single
 * synchronisation.  d_corners_xy (device, capacity pairs) is still filled. */
int vofe_gftt_host(vofe_ctx* ctx, const uint8_t* d_img, int height, int width, int pitch,
                   int max_corners, double quality, double min_distance, const uint8_t* d_mask,
                   int block_size, int aperture, int harris, double k, float* d_corners_xy,
                   float* h_corners_xy, int capacity, int* h_n_out);

/* Batched variant: `batch` images of identical geometry; image b writes its corners to
 * d_corners_xy + b*capacity*2 and its count to h_n_out[b].  No mask. */
int vofe_gftt_batch(vofe_ctx* ctx, const uint8_t* d_img, int batch, int height, int width,
                    int pitch, size_t img_stride, int max_corners, double quality,
                    double min_distance, int block_size, int aperture, int harris, double k,
                    float* d_corners_xy, int capacity, int* h_n_out);

/* ---- candidate bookkeeping ------------------------------------------------
 * Replaces the new-feature de-duplication of reference continuous.py:408-416
 *     cdist(features_current, existing_features) -> np.min(axis=1) -> > THRESHOLD_NEW_KEYPOINTS
 * without building the distance matrix: d_mask[i] = 1 iff every existing point is farther than
 * `threshold` from feature i, decided exactly as scipy's float64 Euclidean cdist decides it
 * (NaN distances give 0).  Points are (x,y) f32 pairs; n_existing == 0 gives all ones (numpy itself
 * raises on the empty reduction; the Python mirror keeps that).  Asynchronous on the ctx stream. */
int vofe_new_feature_mask(vofe_ctx* ctx, const float* d_features_xy, int n_features,
                          const float* d_existing_xy, int n_existing, double threshold, uint8_t* d_mask);
/* The same with the existing points in two arrays (reference continuous.py:408: np.vstack([Ci.T, Pi.T]) -- the
 * candidates and the landmarks live in separate device buffers, the stacked copy is never made). */
int vofe_new_feature_mask2(vofe_ctx* ctx, const float* d_features_xy, int n_features, const float* d_existing_a_xy, int n_a,
                           const float* d_existing_b_xy, int n_b, double threshold, uint8_t* d_mask);

/* Replaces the candidate update of reference continuous.py:391-405 and the triangulation test of :446-448:
 *     valid = status == 1;  Ci = tracked[:, valid];  Fi = Fi[:, valid];  Ti = Ti[:, valid]
 *     d = np.sqrt(np.sum((Ci - Fi) ** 2, axis=0));  good = (d > dist_min) & (d < dist_max)   -> kept, in order
 *     new_landmarks = d > dist_triangulate                                                   -> d_triangulate_out
 * d_tracked_xy (n,2) f32 and d_status (n) are the outputs of vofe_track_fb; d_first_xy is (n,2) float32
 * (first_is_f64 == 0) or float64 — numpy evaluates the distance in that type and so does the kernel, with the
 * same decisions (NaN distances are dropped).  d_transform is (n, transform_dim) float64, transform_dim <= 16
 * (the reference keeps 12 or 16 numbers per candidate), or NULL.  The kept candidates are written densely, in input
 * order, to d_tracked_out / d_first_out / d_transform_out / d_triangulate_out (u8), each of which may be the
 * buffer its input came from; the count goes to *d_count_out (device) and, when h_count_out is not NULL, to the
 * host after a synchronisation (otherwise the call is asynchronous on the ctx stream). */
int vofe_candidate_filter(vofe_ctx* ctx, const float* d_tracked_xy, const uint8_t* d_status,
                          const void* d_first_xy, int first_is_f64, const double* d_transform, int transform_dim,
                          int n, double dist_min, double dist_max, double dist_triangulate, float* d_tracked_out,
                          void* d_first_out, double* d_transform_out, uint8_t* d_triangulate_out,
                          int* d_count_out, int* h_count_out);

/* Replaces the point bookkeeping of the bootstrap loop, reference initialization.py:54-64 (and :69-70):
 *     good_old = good_old[status.flatten() == 1];  good_new = tracked_points[status.flatten() == 1]
 *     distances = np.sqrt(np.sum((good_new - good_old) ** 2, axis=2)).flatten()
 *     drop = np.where(distances < dist_threshold_move);  good_old / good_new = np.delete(.., drop, axis=0)
 * d_new_xy (n,2) f32 are the points vofe_lk_track returned, d_flag (n) its status, d_old_xy (n,2) f32 the positions of the
 * same points in the first bootstrap image.  A pair is kept iff flag == 1 and NOT (distance < dist_move), the distance
 * evaluated in float32 with numpy's decisions (a NaN distance is not below the threshold: the pair stays, as np.delete
 * leaves it).  With d_flag = the inlier mask of cv2.findEssentialMat and dist_move = 0 the same call is
 * good_old[mask.flatten() == 1] / good_new[mask.flatten() == 1] (:69-70).  Kept pairs are written densely, in input order,
 * to d_new_out / d_old_out, each of which may be the buffer its input came from; count as in vofe_candidate_filter. */
int vofe_pair_filter(vofe_ctx* ctx, const float* d_new_xy, const uint8_t* d_flag, const float* d_old_xy, int n,
                     double dist_move, float* d_new_out, float* d_old_out, int* d_count_out, int* h_count_out);

/* Replaces the per-candidate triangulation loop of reference continuous.py:450-485 (cv2.triangulatePoints per
 * candidate, negative-depth test, baseline angle of continuous.py:274-309) with one launch over all candidates.
 * d_K is the 3x3 intrinsic matrix and d_pose the current 3x4 pose (row-major float64, device); d_cur_xy (n,2) f32,
 * d_first_xy (n,2) f32/f64 and d_transform (n, transform_dim >= 12) f64 as produced by vofe_candidate_filter;
 * d_flag_in (n) is its triangulation flag.  d_flag_out[i] = 1 iff candidate i was flagged, its triangulated point
 * X (rounded to the type of d_first_xy, as OpenCV stores points4D) has X[2] >= -t_first[2], and the angle between
 * the bearings from the two camera positions exceeds alpha_min (radians); d_xyz (n,3) f64 receives X for every
 * flagged candidate.  float64 arithmetic; agrees with cv2's SVD to rounding error, not bit for bit.
 * Asynchronous on the ctx stream. */
int vofe_triangulate_candidates(vofe_ctx* ctx, const double* d_K, const double* d_pose, const float* d_cur_xy,
                                const void* d_first_xy, int first_is_f64, const double* d_transform,
                                int transform_dim, const uint8_t* d_flag_in, int n, double alpha_min,
                                uint8_t* d_flag_out, double* d_xyz);

/* Replaces the tail of ContinuousVO._handle_candidates (reference continuous.py:476-495) on device-resident arrays:
 *     Pi = hstack([Pi, Ci[:, new_landmarks]]);  Xi = hstack([Xi, points3D])              -> promoted candidates
 *     Ci, Fi, Ti = Ci[:, no_new_landmark], ...                                           -> removed from the candidates, in order
 *     Ci = hstack([Ci, new_features]); Fi = hstack([Fi, new_features]); Ti = hstack([Ti, tiled_pose])
 * d_cand_xy / d_first_xy / d_transform hold n_candidates candidates (capacity cap_candidates) as vofe_candidate_filter
 * left them and are updated in place; d_promote (n_candidates) is the flag array of vofe_triangulate_candidates and
 * d_xyz its points; d_features_xy / d_feature_mask (n_features) are the detections of the frame and the mask of
 * vofe_new_feature_mask; d_pose the current 3x4 pose.  Promoted candidates are appended to d_landmarks_xy (n_landmarks
 * entries, capacity cap_landmarks) and written with their 3-D points to d_new_landmarks_xy / _xyz (which may be pinned
 * host memory mapped into the device, so the host appends them to its own Pi / Xi).  d_track_next receives the next
 * frame's tracking input [landmarks | candidates] and d_counts = {promoted, candidates, landmarks, overflow flag}.
 * Order-preserving throughout, as numpy's boolean indexing is.  Asynchronous on the ctx stream. */
int vofe_candidate_commit(vofe_ctx* ctx, float* d_cand_xy, void* d_first_xy, int first_is_f64, double* d_transform,
                          int transform_dim, int n_candidates, int cap_candidates, const uint8_t* d_promote, const double* d_xyz,
                          const float* d_features_xy, const uint8_t* d_feature_mask, int n_features, const double* d_pose,
                          float* d_landmarks_xy, int n_landmarks, int cap_landmarks, float* d_track_next, float* d_new_landmarks_xy,
                          double* d_new_landmarks_xyz, int* d_counts);

/* ---- frame state resident on the device (SURVEY.md 8f ranks 1-4 as one object) ------------------
 * Replaces, for one sequence, what ContinuousVO.process_frame (reference continuous.py:73-114) asks of the front end
 * per frame: landmark tracking (:86-92), and _handle_candidates (:349-495: detection, candidate tracking, status
 * compaction, travel-distance band, cdist de-duplication, triangulation, promotion, append).  The image-point arrays
 * of the reference's FrameState (framestate.py:13-78: landmarks Pi, candidates Ci / Fi / Ti) live in device memory
 * across frames; a frame is two calls around the caller's CPU pose estimation (continuous.py:166-272), one host wait
 * each.  All pointers of this group are HOST pointers (the reference holds numpy arrays); arrays are row-major with
 * one point per row: (K,2) f32 image points, (K,3) f64 world points, (M,12) f64 first-observation poses, first
 * observations (M,2) in f64 or f32 as `first_is_f64` says (numpy keeps whichever dtype the caller's FrameState has).
 * The object owns its contexts and streams (non-blocking), is bound to one device and is not thread-safe. */
typedef struct vofe_framestate vofe_framestate;
typedef struct vofe_framestate_params {
    int height, width;
    int win_w, win_h, max_level;                /* LK: continuous.py:24-29, 137-141 */
    int crit_type, max_count;
    double eps, min_eig_thr;
    float fb_thresh;                            /* continuous.py:158: 1.0 */
    int max_corners;                            /* goodFeaturesToTrack: continuous.py:380-385; must be positive */
    double quality_level, min_distance;
    double thr_new_keypoints;                   /* constants.py THRESHOLD_NEW_KEYPOINTS (continuous.py:413) */
    double thr_triangulation;                   /* THRESHOLD_PIXEL_DIST_TRIANGULATION (:448) */
    double thr_candidates_min, thr_candidates_max;   /* MIN / MAX travel distance (:400-401) */
    double thr_alpha_rad;                       /* minimum baseline angle (:475, :274-309), radians */
    int cap_landmarks, cap_candidates;          /* capacities of the device arrays; exceeding one is VOFE_E_NOMEM */
    int first_is_f64;
    double K[9];                                /* intrinsics, row-major */
} vofe_framestate_params;

int vofe_framestate_create(int device, const vofe_framestate_params* params, vofe_framestate** out);
int vofe_framestate_destroy(vofe_framestate* fs);
/* Start from a FrameState on its frame (reference main.py:82-90 builds the first one): uploads the frame, builds its
 * pyramid, uploads the point arrays.  h_img: height rows of width bytes, `pitch` bytes apart. */
int vofe_framestate_reset(vofe_framestate* fs, const uint8_t* h_img, int pitch, const float* h_landmarks_xy,
                          const double* h_landmarks_xyz, int n_landmarks, const float* h_cand_xy, const void* h_first_xy,
                          const double* h_transform, int n_candidates);
/* Pinned staging buffer (height x width, contiguous) the NEXT frame may be written into directly -- e.g. by the image
 * decoder -- so that vofe_framestate_track(fs, that pointer, width, ...) skips its host-side copy. */
int vofe_framestate_frame_buffer(vofe_framestate* fs, uint8_t** h_buf);
/* Phase 1 (continuous.py:86-87, 380-405): upload, ONE pyramid, ONE forward-backward launch over [landmarks | candidates],
 * candidate compaction + distance band, detection on a second stream; one wait.  *h_tracked_xy / *h_status point into
 * pinned memory owned by the object ((K,2) f32 and (K) u8 for the K landmarks, valid until the next call on fs);
 * *n_landmarks = K; *n_features / *n_candidates_kept (may be NULL) report the detections and surviving candidates. */
int vofe_framestate_track(vofe_framestate* fs, const uint8_t* h_img, int pitch, const float** h_tracked_xy,
                          const uint8_t** h_status, int* n_landmarks, int* n_features, int* n_candidates_kept);
/* Phase 2 (continuous.py:90-101 selection by h_keep, a (K) u8 mask = tracked AND kept by the pose estimation;
 * :408-495 de-duplication, triangulation, promotion, append).  h_pose: 3x4 row-major f64.  World points of the kept
 * and the promoted landmarks are maintained on the host side of the object (only the CPU pose estimation reads them).
 * *n_promoted may be NULL. */
int vofe_framestate_update(vofe_framestate* fs, const double* h_pose, const uint8_t* h_keep, int* n_promoted);
/* Phase 2 without its wait: queues the same work and returns; h_pose / h_keep are consumed before the call returns.  The
 * update is collected by vofe_framestate_wait or by the next call that needs its outcome (track, counts,
 * landmarks_world, download); vofe_framestate_track first copies, uploads and builds the pyramid of its frame and
 * only then waits, so the update overlaps whatever the caller does between the two calls and the host copy of the next
 * frame.  A capacity overflow of the queued update is reported (VOFE_E_NOMEM) by the call that collects it.
 * vofe_framestate_update == vofe_framestate_update_async + vofe_framestate_wait. */
int vofe_framestate_update_async(vofe_framestate* fs, const double* h_pose, const uint8_t* h_keep);
int vofe_framestate_wait(vofe_framestate* fs, int* n_promoted);
int vofe_framestate_counts(vofe_framestate* fs, int* n_landmarks, int* n_candidates);
/* World points (K,3) f64 of the current landmarks, in landmark order (pointer valid until the next update/reset).
 * Between track and update these are the landmarks track started from, row for row the points it returned: what the
 * pose estimation pairs the tracked image points with (continuous.py:95-101). */
int vofe_framestate_landmarks_world(vofe_framestate* fs, const double** h_xyz, int* n_landmarks);
/* The FrameState arrays as the reference holds them (any pointer may be NULL); synchronous; not between track and update. */
int vofe_framestate_download(vofe_framestate* fs, float* h_landmarks_xy, float* h_cand_xy, void* h_first_xy,
                             double* h_transform);

#ifdef __cplusplus
}
#endif
#endif
