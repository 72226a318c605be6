// vofe_framestate: the reference's FrameState (framestate.py:13-78) with its image-point arrays resident in HBM, and
// one frame of ContinuousVO.process_frame's front end (continuous.py:86-92 landmark tracking, :349-495
// _handle_candidates) as two host calls around the CPU pose estimation.  Host-side orchestration only: every kernel
// is launched through the entry points of api.hip; this file owns streams, buffers and the order of the launches.
//
//   track  (phase 1)  H2D frame -> pyramid -> ONE forward-backward launch over [landmarks | candidates] -> candidate
//                     compaction + distance band                                   (tracker stream)
//                     goodFeaturesToTrack on the uploaded frame                    (detector stream, behind the upload only)
//   update (phase 2)  triangulation of matured candidates (pose, K as kernel arguments) -> H2D surviving landmarks ->
//                     de-duplication of the detections -> promotion / removal / append, next frame's tracking input
//
// Tracker outputs, counts and promoted landmarks are written by the kernels straight into pinned host memory that is
// mapped into the device, so each phase has exactly one stream wait and no D2H copy command.
#include <cstring>
#include <new>

#include "vofe_internal.h"

struct vofe_framestate {
    vofe_framestate_params p;
    int device;
    hipStream_t s_trk, s_det;
    hipEvent_t ev_up;
    vofe_ctx *c_trk, *c_det;
    vofe_pyr* pyr;                 // two slots: previous and current frame
    uint8_t* d_img[2];             // the frames themselves (the detector reads the current one)
    uint8_t* h_stage;              // pinned staging buffer of one frame
    // device arrays
    float* d_track;                // (capP + capC, 2): [landmarks | candidates] of the last frame = the tracker's input
    float* d_C;                    // (capC, 2) candidates, current positions
    void* d_F;                     // (capC, 2) first observations, f32 or f64
    double* d_T;                   // (capC, 12) first-observation poses
    uint8_t *d_tri, *d_prom;       // (capC) triangulation / promotion flags
    double* d_xyz;                 // (capC, 3) triangulated points
    float* d_feat;                 // (max_corners, 2) detections of the frame
    uint8_t* d_fmask;              // (max_corners)
    uint8_t* d_par;                // phase 2: 192 B unused (pose and K travel as kernel arguments) | surviving landmarks (capP, 2) f32
    uint8_t* h_par;                // pinned source of d_par, mapped into the device (d_par_host): phase 2 fetches it from a kernel
    uint8_t* d_par_host;
    // results in mapped pinned memory
    uint8_t *h_res, *d_res;
    size_t o_cnt, o_newP, o_newX;  // byte offsets of the counters / promoted landmarks in the result buffer
    // host side
    double* world;                 // (capP, 3) world points of the landmarks
    int K, M, m1, nf, phase, frame_index;     // phase: 0 idle, 1 tracked (update is due), 2 update queued (not yet collected)
    int K1, last_promoted;                    // landmarks the pose estimation kept; promoted by the last collected update
};

namespace {

size_t first_size(const vofe_framestate* fs) { return fs->p.first_is_f64 ? sizeof(double) : sizeof(float); }

void release(vofe_framestate* fs) {
    if (!fs) return;
    if (!fs->c_trk) { delete fs; return; }       // nothing was created (the device was never validated: do not touch it)
    hipSetDevice(fs->device);
    if (fs->s_trk) hipStreamSynchronize(fs->s_trk);
    if (fs->s_det) hipStreamSynchronize(fs->s_det);
    if (fs->pyr) vofe_pyramid_free(fs->c_trk, fs->pyr);
    if (fs->c_trk) vofe_destroy(fs->c_trk);
    if (fs->c_det) vofe_destroy(fs->c_det);
    void* dev[] = {fs->d_img[0], fs->d_img[1], fs->d_track, fs->d_C, fs->d_F, fs->d_T, fs->d_tri, fs->d_prom,
                   fs->d_xyz, fs->d_feat, fs->d_fmask, fs->d_par};
    for (void* q : dev) if (q) hipFree(q);
    if (fs->h_stage) hipHostFree(fs->h_stage);
    if (fs->h_par) hipHostFree(fs->h_par);
    if (fs->h_res) hipHostFree(fs->h_res);
    if (fs->ev_up) hipEventDestroy(fs->ev_up);
    if (fs->s_trk) hipStreamDestroy(fs->s_trk);
    if (fs->s_det) hipStreamDestroy(fs->s_det);
    delete[] fs->world;
    delete fs;
}

int bind(const vofe_framestate* fs) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == fs->device) return VOFE_OK;
    VOFE_HIP(hipSetDevice(fs->device));
    return VOFE_OK;
}

// frame -> staging (unless the caller wrote it there) -> device image of the next slot -> that slot's pyramid
int push_frame(vofe_framestate* fs, const uint8_t* h_img, int pitch) {
    const int h = fs->p.height, w = fs->p.width;
    if (h_img != fs->h_stage) {
        if (pitch == w) memcpy(fs->h_stage, h_img, (size_t)h * w);
        else for (int y = 0; y < h; ++y) memcpy(fs->h_stage + (size_t)y * w, h_img + (size_t)y * pitch, (size_t)w);
    } else if (pitch != w) {
        vofe_set_error("vofe_framestate: the staging buffer is contiguous (pitch must equal width)");
        return VOFE_E_ARG;
    }
    const int slot = (fs->frame_index + 1) & 1;
    VOFE_HIP(hipMemcpyAsync(fs->d_img[slot], fs->h_stage, (size_t)h * w, hipMemcpyHostToDevice, fs->s_trk));
    VOFE_HIP(hipEventRecord(fs->ev_up, fs->s_trk));
    int rc = vofe_pyramid_build(fs->c_trk, fs->pyr, slot, 1, fs->d_img[slot], w, 0);
    if (rc) return rc;
    fs->frame_index += 1;
    return VOFE_OK;
}

// the host half of phase 2: wait for the queued update, take its counts, append the promoted world points
int collect_update(vofe_framestate* fs) {
    if (fs->phase != 2) return VOFE_OK;
    VOFE_HIP(hipStreamSynchronize(fs->s_trk));
    const int* cnt = reinterpret_cast<const int*>(fs->h_res + fs->o_cnt) + 4;
    const int n_prom = cnt[0], m_out = cnt[1], k_out = cnt[2], overflow = cnt[3];
    if (overflow) {
        fs->K = fs->M = 0; fs->phase = 0; fs->frame_index = -1;      // the arrays were cut at their capacity: start over
        vofe_set_error("vofe_framestate_update: landmark / candidate capacity exceeded (%d / %d); the state needs vofe_framestate_reset",
                       fs->p.cap_landmarks, fs->p.cap_candidates);
        return VOFE_E_NOMEM;
    }
    if (n_prom) memcpy(fs->world + (size_t)fs->K1 * 3, fs->h_res + fs->o_newX, (size_t)n_prom * 24);
    fs->K = k_out; fs->M = m_out; fs->phase = 0;
    fs->last_promoted = n_prom;
    return VOFE_OK;
}

}  // namespace

extern "C" {

int vofe_framestate_create(int device, const vofe_framestate_params* params, vofe_framestate** out) {
    if (!out || !params) { vofe_set_error("vofe_framestate_create: NULL argument"); return VOFE_E_ARG; }
    *out = nullptr;
    const vofe_framestate_params& p = *params;
    if (p.height <= 0 || p.width <= 0 || p.cap_landmarks <= 0 || p.cap_candidates <= 0 || p.max_corners <= 0) {
        vofe_set_error("vofe_framestate_create: frame size, capacities and max_corners must be positive");
        return VOFE_E_ARG;
    }
    if (!(p.quality_level > 0) || !(p.min_distance >= 0)) {
        vofe_set_error("vofe_framestate_create: need qualityLevel > 0, minDistance >= 0");
        return VOFE_E_ARG;
    }
    vofe_framestate* fs = new (std::nothrow) vofe_framestate();
    if (!fs) { vofe_set_error("vofe_framestate_create: out of host memory"); return VOFE_E_NOMEM; }
    memset(fs, 0, sizeof(*fs));
    fs->p = p;
    fs->device = device;
    fs->frame_index = -1;
    int rc = VOFE_OK;
    hipError_t e = hipSuccess;
    const char* what = "";
#define FS_TRY(call) do { if (!rc && e == hipSuccess) { e = (call); what = #call; } } while (0)
#define FS_RC(call) do { if (!rc && e == hipSuccess) rc = (call); } while (0)
    // the first context validates the device (count, range, gfx950) before anything is allocated on it
    FS_RC(vofe_create(device, nullptr, &fs->c_trk));
    if (rc) { release(fs); return rc; }
    int lo = 0, hi = 0;
    FS_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    FS_TRY(hipStreamCreateWithFlags(&fs->s_trk, hipStreamNonBlocking));
    // the detector is a chain of short dependent launches: its stream goes first where the hardware has to choose
    FS_TRY(hipStreamCreateWithPriority(&fs->s_det, hipStreamNonBlocking, hi));
    FS_TRY(hipEventCreateWithFlags(&fs->ev_up, hipEventDisableTiming));
    if (!rc && e == hipSuccess) fs->c_trk->stream = fs->s_trk;
    FS_RC(vofe_create(device, fs->s_det, &fs->c_det));
    FS_RC(vofe_pyramid_create(fs->c_trk, 2, p.height, p.width, p.win_w, p.win_h, p.max_level, 1, &fs->pyr));
    const size_t px = (size_t)p.height * p.width, capP = p.cap_landmarks, capC = p.cap_candidates, ncap = capP + capC;
    const size_t fsz = p.first_is_f64 ? 8 : 4;
    FS_TRY(hipMalloc((void**)&fs->d_img[0], px));
    FS_TRY(hipMalloc((void**)&fs->d_img[1], px));
    FS_TRY(hipMalloc((void**)&fs->d_track, ncap * 8));
    FS_TRY(hipMalloc((void**)&fs->d_C, capC * 8));
    FS_TRY(hipMalloc((void**)&fs->d_F, capC * 2 * fsz));
    FS_TRY(hipMalloc((void**)&fs->d_T, capC * 12 * 8));
    FS_TRY(hipMalloc((void**)&fs->d_tri, capC));
    FS_TRY(hipMalloc((void**)&fs->d_prom, capC));
    FS_TRY(hipMalloc((void**)&fs->d_xyz, capC * 24));
    FS_TRY(hipMalloc((void**)&fs->d_feat, (size_t)p.max_corners * 8));
    FS_TRY(hipMalloc((void**)&fs->d_fmask, (size_t)p.max_corners));
    FS_TRY(hipMalloc((void**)&fs->d_par, 192 + capP * 8));
    FS_TRY(hipHostMalloc((void**)&fs->h_stage, px, hipHostMallocDefault));
    FS_TRY(hipHostMalloc((void**)&fs->h_par, 192 + capP * 8, hipHostMallocMapped));
    FS_TRY(hipHostGetDevicePointer((void**)&fs->d_par_host, fs->h_par, 0));
    // results: tracked xy (8 B) | err (4 B) | status (1 B) per tracked point, then counters, promoted xy, promoted xyz
    fs->o_cnt = (ncap * 13 + 63) / 64 * 64;
    fs->o_newP = fs->o_cnt + 64;
    fs->o_newX = fs->o_newP + capC * 8;
    FS_TRY(hipHostMalloc((void**)&fs->h_res, fs->o_newX + capC * 24, hipHostMallocMapped));
    FS_TRY(hipHostGetDevicePointer((void**)&fs->d_res, fs->h_res, 0));
#undef FS_TRY
#undef FS_RC
    if (!rc && e != hipSuccess) rc = vofe_hip_fail(e, what);
    if (!rc) {
        fs->world = new (std::nothrow) double[capP * 3];
        if (!fs->world) { vofe_set_error("vofe_framestate_create: out of host memory"); rc = VOFE_E_NOMEM; }
    }
    if (rc) { release(fs); return rc; }
    memset(fs->h_res, 0, fs->o_newX + capC * 24);
    memset(fs->h_par, 0, 192);
    memcpy(fs->h_par + 96, p.K, 9 * sizeof(double));
    *out = fs;
    return VOFE_OK;
}

int vofe_framestate_destroy(vofe_framestate* fs) {
    release(fs);
    return VOFE_OK;
}

int vofe_framestate_frame_buffer(vofe_framestate* fs, uint8_t** h_buf) {
    if (!fs || !h_buf) { vofe_set_error("vofe_framestate_frame_buffer: NULL argument"); return VOFE_E_ARG; }
    *h_buf = fs->h_stage;
    return VOFE_OK;
}

int vofe_framestate_reset(vofe_framestate* fs, const uint8_t* h_img, int pitch, const float* h_landmarks_xy,
                          const double* h_landmarks_xyz, int n_landmarks, const float* h_cand_xy, const void* h_first_xy,
                          const double* h_transform, int n_candidates) {
    if (!fs || !h_img) { vofe_set_error("vofe_framestate_reset: NULL argument"); return VOFE_E_ARG; }
    if (pitch < fs->p.width) { vofe_set_error("vofe_framestate_reset: pitch %d < width %d", pitch, fs->p.width); return VOFE_E_ARG; }
    const int K = n_landmarks, M = n_candidates;
    if (K < 0 || M < 0 || K > fs->p.cap_landmarks || M > fs->p.cap_candidates) {
        vofe_set_error("vofe_framestate_reset: %d landmarks / %d candidates outside the capacities (%d / %d)", K, M,
                       fs->p.cap_landmarks, fs->p.cap_candidates);
        return VOFE_E_ARG;
    }
    if ((K > 0 && (!h_landmarks_xy || !h_landmarks_xyz)) || (M > 0 && (!h_cand_xy || !h_first_xy || !h_transform))) {
        vofe_set_error("vofe_framestate_reset: NULL point array");
        return VOFE_E_ARG;
    }
    int rc = bind(fs);
    if (rc) return rc;
    VOFE_HIP(hipStreamSynchronize(fs->s_trk));       // a queued update is dropped with the state it belongs to
    fs->frame_index = -1;
    fs->phase = 0;
    rc = push_frame(fs, h_img, pitch);
    if (rc) return rc;
    hipStream_t st = fs->s_trk;
    if (K) {
        VOFE_HIP(hipMemcpyAsync(fs->d_track, h_landmarks_xy, (size_t)K * 8, hipMemcpyHostToDevice, st));
        memcpy(fs->world, h_landmarks_xyz, (size_t)K * 24);
    }
    if (M) {
        VOFE_HIP(hipMemcpyAsync(fs->d_track + (size_t)K * 2, h_cand_xy, (size_t)M * 8, hipMemcpyHostToDevice, st));
        VOFE_HIP(hipMemcpyAsync(fs->d_C, h_cand_xy, (size_t)M * 8, hipMemcpyHostToDevice, st));
        VOFE_HIP(hipMemcpyAsync(fs->d_F, h_first_xy, (size_t)M * 2 * first_size(fs), hipMemcpyHostToDevice, st));
        VOFE_HIP(hipMemcpyAsync(fs->d_T, h_transform, (size_t)M * 96, hipMemcpyHostToDevice, st));
    }
    VOFE_HIP(hipStreamSynchronize(st));
    fs->K = K; fs->M = M; fs->m1 = 0; fs->nf = 0;
    return VOFE_OK;
}

// A failure after the frame has been pushed (frame_index advanced, resident arrays possibly half updated) leaves nothing a
// later call could safely continue from: the object then demands vofe_framestate_reset, like after an overflow.
static void invalidate_state(vofe_framestate* fs) {
    hipStreamSynchronize(fs->s_trk);
    hipStreamSynchronize(fs->s_det);
    fs->frame_index = -1;
    fs->phase = 0;
}

static int track_impl(vofe_framestate* fs, const uint8_t* h_img, int pitch, const float** h_tracked_xy,
                      const uint8_t** h_status, int* n_landmarks, int* n_features, int* n_candidates_kept, bool* pushed) {
    int rc = bind(fs);
    if (rc) return rc;
    const vofe_framestate_params& p = fs->p;
    const int prev_slot = fs->frame_index & 1;
    // staging copy, upload and pyramid of the new frame do not depend on the point counts: they are queued (and the
    // host copy made) BEFORE the wait for an update that vofe_framestate_update_async left running
    rc = push_frame(fs, h_img, pitch);
    if (rc) return rc;
    *pushed = true;
    rc = collect_update(fs);
    if (rc) return rc;
    const int slot = fs->frame_index & 1;
    // detection (continuous.py:380-385) needs the uploaded frame only: queue its stream behind the copy, not behind the
    // pyramid or the tracker launch
    VOFE_HIP(hipStreamWaitEvent(fs->s_det, fs->ev_up, 0));
    const int K = fs->K, M = fs->M, n = K + M;
    float* o_next = reinterpret_cast<float*>(fs->d_res);
    float* o_err = reinterpret_cast<float*>(fs->d_res + (size_t)n * 8);
    uint8_t* o_st = fs->d_res + (size_t)n * 12;
    int* o_cnt = reinterpret_cast<int*>(fs->d_res + fs->o_cnt);
    if (n) {
        rc = vofe_track_fb(fs->c_trk, fs->pyr, prev_slot, fs->pyr, slot, fs->d_track, nullptr, 1, o_next, o_st, o_err, n,
                           p.crit_type, p.max_count, p.eps, p.min_eig_thr, p.fb_thresh);
        if (rc) return rc;
    }
    if (M) {
        // status == 1, travel-distance band, triangulation flag (continuous.py:391-405, 446-448): reads the tracker's
        // outputs where the kernel left them and compacts the resident candidate arrays in place
        rc = vofe_candidate_filter(fs->c_trk, o_next + (size_t)K * 2, o_st + K, fs->d_F, p.first_is_f64, fs->d_T, 12, M,
                                   p.thr_candidates_min, p.thr_candidates_max, p.thr_triangulation, fs->d_C, fs->d_F, fs->d_T,
                                   fs->d_tri, o_cnt, nullptr);
        if (rc) return rc;
    }
    int nf = 0;
    rc = vofe_gftt(fs->c_det, fs->d_img[slot], p.height, p.width, p.width, p.max_corners, p.quality_level, p.min_distance,
                   nullptr, 3, 3, 0, 0.04, fs->d_feat, p.max_corners, &nf);       // returns when the detector's stream is done
    if (rc) { hipStreamSynchronize(fs->s_trk); return rc; }
    VOFE_HIP(hipStreamSynchronize(fs->s_trk));
    fs->nf = nf;
    fs->m1 = M ? reinterpret_cast<const int*>(fs->h_res + fs->o_cnt)[0] : 0;
    fs->phase = 1;
    *h_tracked_xy = reinterpret_cast<const float*>(fs->h_res);
    *h_status = fs->h_res + (size_t)n * 12;
    *n_landmarks = K;
    if (n_features) *n_features = nf;
    if (n_candidates_kept) *n_candidates_kept = fs->m1;
    return VOFE_OK;
}

int vofe_framestate_track(vofe_framestate* fs, const uint8_t* h_img, int pitch, const float** h_tracked_xy,
                          const uint8_t** h_status, int* n_landmarks, int* n_features, int* n_candidates_kept) {
    if (!fs || !h_img || !h_tracked_xy || !h_status || !n_landmarks) { vofe_set_error("vofe_framestate_track: NULL argument"); return VOFE_E_ARG; }
    if (fs->frame_index < 0) { vofe_set_error("vofe_framestate_track before vofe_framestate_reset"); return VOFE_E_ARG; }
    if (fs->phase == 1) { vofe_set_error("vofe_framestate_track called twice without vofe_framestate_update"); return VOFE_E_ARG; }
    if (pitch < fs->p.width) { vofe_set_error("vofe_framestate_track: pitch %d < width %d", pitch, fs->p.width); return VOFE_E_ARG; }
    bool pushed = false;
    const int rc = track_impl(fs, h_img, pitch, h_tracked_xy, h_status, n_landmarks, n_features, n_candidates_kept, &pushed);
    if (rc && pushed) invalidate_state(fs);        // the previous frame's points no longer match the pyramid slots
    return rc;
}

static int update_impl(vofe_framestate* fs, const double* h_pose, const uint8_t* h_keep, bool* touched) {
    int rc = bind(fs);
    if (rc) return rc;
    const vofe_framestate_params& p = fs->p;
    const int K = fs->K, m1 = fs->m1, nf = fs->nf;
    int K1 = 0;
    for (int i = 0; i < K; ++i) K1 += h_keep[i] != 0;
    if (nf && m1 + K1 == 0) {
        // the reference's np.min(cdist(features, empty), axis=1) raises here (continuous.py:410-411)
        vofe_set_error("zero-size array to reduction operation minimum which has no identity");
        return VOFE_E_ARG;
    }
    // landmarks the pose estimation kept (continuous.py:90-101): image points into the upload, world points in place
    *touched = true;                               // from here on a failure leaves `world` and the resident arrays out of step
    const float* trk = reinterpret_cast<const float*>(fs->h_res);
    float* P1 = reinterpret_cast<float*>(fs->h_par + 192);
    for (int i = 0, o = 0; i < K; ++i) {
        if (!h_keep[i]) continue;
        P1[2 * o] = trk[2 * i]; P1[2 * o + 1] = trk[2 * i + 1];
        if (o != i) { fs->world[3 * o] = fs->world[3 * i]; fs->world[3 * o + 1] = fs->world[3 * i + 1]; fs->world[3 * o + 2] = fs->world[3 * i + 2]; }
        ++o;
    }
    // One stream: cross-stream events cost more host time here than overlapping the two independent kernels gains
    // (measured: update 41 -> 51 us with the de-duplication on the detector stream).  Pose and intrinsics travel as
    // kernel arguments; the list of surviving landmarks is read from mapped pinned memory by the first launch.
    VofePoseK pk;
    memcpy(pk.pose, h_pose, sizeof(pk.pose));
    memcpy(pk.K, p.K, sizeof(pk.K));
    float* d_P = reinterpret_cast<float*>(fs->d_par + 192);
    // (the surviving landmarks ride along with the triangulation launch: extra blocks fetch them from the mapped pinned
    // buffer instead of a copy command in front of the kernels: update 41-45 -> 38-39 us)
    if (m1 || K1) {
        rc = vofe_launch_triangulate(fs->c_trk, nullptr, nullptr, &pk, fs->d_C, fs->d_F, p.first_is_f64, fs->d_T, 12, fs->d_tri, m1,
                                     p.thr_alpha_rad, fs->d_prom, fs->d_xyz, reinterpret_cast<const float*>(fs->d_par_host + 192), d_P, K1);
        if (rc) return rc;
    }
    if (nf) {
        rc = vofe_new_feature_mask2(fs->c_trk, fs->d_feat, nf, fs->d_C, m1, d_P, K1, p.thr_new_keypoints, fs->d_fmask);
        if (rc) return rc;
    }
    rc = vofe_launch_candidate_commit(fs->c_trk, fs->d_C, fs->d_F, p.first_is_f64, fs->d_T, 12, m1, p.cap_candidates, fs->d_prom, fs->d_xyz,
                                      fs->d_feat, fs->d_fmask, nf, nullptr, &pk, d_P, K1, p.cap_landmarks, fs->d_track,
                                      reinterpret_cast<float*>(fs->d_res + fs->o_newP), reinterpret_cast<double*>(fs->d_res + fs->o_newX),
                                      reinterpret_cast<int*>(fs->d_res + fs->o_cnt) + 4);
    if (rc) return rc;
    fs->K1 = K1;
    fs->phase = 2;
    return VOFE_OK;
}

int vofe_framestate_update_async(vofe_framestate* fs, const double* h_pose, const uint8_t* h_keep) {
    if (!fs || !h_pose) { vofe_set_error("vofe_framestate_update: NULL argument"); return VOFE_E_ARG; }
    if (fs->phase != 1) { vofe_set_error("vofe_framestate_update before vofe_framestate_track"); return VOFE_E_ARG; }
    if (fs->K > 0 && !h_keep) { vofe_set_error("vofe_framestate_update: keep mask is NULL"); return VOFE_E_ARG; }
    bool touched = false;
    const int rc = update_impl(fs, h_pose, h_keep, &touched);
    if (rc && touched) invalidate_state(fs);       // world points compacted, device arrays not: only a reset can continue
    return rc;
}

int vofe_framestate_wait(vofe_framestate* fs, int* n_promoted) {
    if (n_promoted) *n_promoted = 0;
    if (!fs) { vofe_set_error("vofe_framestate_wait: NULL argument"); return VOFE_E_ARG; }
    int rc = bind(fs);
    if (rc) return rc;
    rc = collect_update(fs);
    if (rc) return rc;
    if (n_promoted) *n_promoted = fs->last_promoted;
    return VOFE_OK;
}

int vofe_framestate_update(vofe_framestate* fs, const double* h_pose, const uint8_t* h_keep, int* n_promoted) {
    if (n_promoted) *n_promoted = 0;
    int rc = vofe_framestate_update_async(fs, h_pose, h_keep);
    if (rc) return rc;
    return vofe_framestate_wait(fs, n_promoted);
}

int vofe_framestate_counts(vofe_framestate* fs, int* n_landmarks, int* n_candidates) {
    if (!fs) { vofe_set_error("vofe_framestate_counts: NULL argument"); return VOFE_E_ARG; }
    if (fs->phase == 1) { vofe_set_error("vofe_framestate_counts between track and update"); return VOFE_E_ARG; }
    int rc = collect_update(fs);
    if (rc) return rc;
    if (n_landmarks) *n_landmarks = fs->K;
    if (n_candidates) *n_candidates = fs->M;
    return VOFE_OK;
}

int vofe_framestate_landmarks_world(vofe_framestate* fs, const double** h_xyz, int* n_landmarks) {
    if (!fs || !h_xyz || !n_landmarks) { vofe_set_error("vofe_framestate_landmarks_world: NULL argument"); return VOFE_E_ARG; }
    int rc0 = collect_update(fs);          // (between track and update the array is the one track started from: what PnP pairs the tracked points with)
    if (rc0) return rc0;
    *h_xyz = fs->world;
    *n_landmarks = fs->K;
    return VOFE_OK;
}

int vofe_framestate_download(vofe_framestate* fs, float* h_landmarks_xy, float* h_cand_xy, void* h_first_xy,
                             double* h_transform) {
    if (!fs) { vofe_set_error("vofe_framestate_download: NULL argument"); return VOFE_E_ARG; }
    if (fs->phase == 1) { vofe_set_error("vofe_framestate_download between track and update"); return VOFE_E_ARG; }
    int rc = bind(fs);
    if (rc) return rc;
    rc = collect_update(fs);
    if (rc) return rc;
    hipStream_t st = fs->s_trk;
    const size_t K = fs->K, M = fs->M;
    if (h_landmarks_xy && K) VOFE_HIP(hipMemcpyAsync(h_landmarks_xy, fs->d_track, K * 8, hipMemcpyDeviceToHost, st));
    if (h_cand_xy && M) VOFE_HIP(hipMemcpyAsync(h_cand_xy, fs->d_C, M * 8, hipMemcpyDeviceToHost, st));
    if (h_first_xy && M) VOFE_HIP(hipMemcpyAsync(h_first_xy, fs->d_F, M * 2 * first_size(fs), hipMemcpyDeviceToHost, st));
    if (h_transform && M) VOFE_HIP(hipMemcpyAsync(h_transform, fs->d_T, M * 96, hipMemcpyDeviceToHost, st));
    VOFE_HIP(hipStreamSynchronize(st));
    return VOFE_OK;
}

}  // extern "C"
