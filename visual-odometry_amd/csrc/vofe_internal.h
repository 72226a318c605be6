// Internal declarations shared by the HIP translation units of libvofe.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <climits>
#include <string>
#include <vector>

#include "../../include/vofe.h"

#define VOFE_MAX_LEVELS 16

// One pyramid level as the kernels see it.  I/D point at pixel (0,0) of slot 0; D packs
// (Ix | Iy << 16) per pixel.  pitch is in pixels for both planes (I: bytes, D: 4-byte units).
struct VofeLevel {
    uint8_t* I;
    int* D;
    int w, h;
    int pitch;
    int rows;                 // h + 2*pad_y
    long long slot_stride;    // pixels between consecutive slots (same for I and D)
};

// VOFE_* environment switches (tests, tuning, measurements).  Read ONCE per context at vofe_create -- not on the
// launch paths, whose latency is measured in tens of microseconds -- and again by vofe_reload_env (tests).
struct VofeKnobs {
    char lk[8];               // VOFE_LK: "generic" | "x1" | "x2" | "x4" | "lp" | "wide" | "" (size rule)
    char pyramid;             // VOFE_PYRAMID: 'f'used | 'l'evels | 0 (size rule)
    bool pad_copy_generic;    // VOFE_PAD_COPY=generic: the any-width copy / border kernels
    bool score_tile;          // VOFE_CORNER_SCORE=tile: the LDS-tile score kernel
    bool score_generic;       // VOFE_CORNER_SCORE=generic: the any-blockSize / any-aperture kernel for block 3 / aperture 3 too
    int score_rs;             // VOFE_SCORE_RS: rows per strip (0 = by launch size)
    bool pyr_step_off;        // VOFE_PYR_STEP=0: per-level path without the level pipeline (pyrDown + Scharr sharing launches)
    bool pyr_tail_off;        // VOFE_PYR_TAIL=0: level pipeline without its tail kernel (every level a pyrDown launch, the top level's Scharr pass alone)
    bool pyr_tail_split;      // VOFE_PYR_TAIL=split: the tail's tiles and its Scharr riders as two launches (measurement)
    int pyr_tail_th;          // VOFE_PYR_TAIL_TH: rows of the tail kernel's top-level tile, 8 or 16 (0 = 16)
    int pyr_borders_separate; // VOFE_PYR_BORDERS=separate: the fused pyramid build leaves the border bands to a second launch (tests, A/B)
    int nms_rows;             // VOFE_NMS_ROWS: rows per thread of the NMS kernel, 1 / 2 / 4 / 8 (0 = by launch size)
    bool gftt_bitonic;        // VOFE_GFTT_SORT=bitonic
    bool gftt_debug;          // VOFE_GFTT_DEBUG
    char gftt_chain;          // VOFE_GFTT_CHAIN: 's'plit = the last stage always as gather / rank_count / emit (round 1's chain); 'r'ank = that chain for long prefixes only
    int sel_blocks, sel_launches, sel_passes, sel_spins, sel_prefix, sel_full;   // VOFE_SEL_*
    bool sel_blocks_set;      // VOFE_SEL_BLOCKS was given (otherwise the block count per image shrinks with the batch)
};

// Measurement aid (vofe_profile_begin / vofe_profile_end): while a context profiles, every kernel launch of the
// streaming stages and the detector is bracketed by two HIP events on the context's stream.
struct VofeProfEntry { std::string name; hipEvent_t a, b; };
struct VofeProf { std::vector<VofeProfEntry> entries; };

struct vofe_ctx {
    int device;
    hipStream_t stream;
    VofeKnobs knobs;
    VofeProf* prof;           // non-NULL between vofe_profile_begin and vofe_profile_end
    // GFTT workspace (grown on demand)
    void* ws;
    size_t ws_bytes;
    int* h_pinned;            // small pinned readback buffer (mapped: kernels write results into it directly)
    int* d_pinned;            // the same buffer as the device addresses it
    size_t h_pinned_ints;
    int gftt_clean_batch;     // batch count whose header/histogram layout the last detector call left zeroed (0: none)
    void* gftt_clean_ws;      // ... in this workspace allocation
    unsigned gftt_last_ncand; // candidates of the previous detector call (sizes the last stage when the whole list is taken)
    unsigned long long* lk_counters;   // optional device counters (vofe_lk_set_counters)
};

struct vofe_pyr {
    int slots, height, width, win_w, win_h, levels, with_deriv;
    int pad_x, pad_y;
    VofeLevel lv[VOFE_MAX_LEVELS];     // host copy
    VofeLevel* d_lv;                   // device copy (levels entries)
    uint8_t* d_I;                      // allocation bases
    int* d_D;
};

void vofe_set_error(const char* fmt, ...);
void vofe_read_knobs(VofeKnobs* k);
// per-context set-up of the detector (corners.hip): binds the device's copy of the static scratch array
// (hipGetSymbolAddress, once per device) and allocates the small pinned header buffer every detector call reads back
int vofe_gftt_setup(vofe_ctx* ctx);
int vofe_hip_fail(hipError_t e, const char* what);

// brackets the launches issued in its scope with HIP events when the context profiles; otherwise one pointer test
struct VofeKernelScope {
    vofe_ctx* c;
    long idx;
    VofeKernelScope(vofe_ctx* ctx, const char* name, int level = -1) : c(ctx), idx(-1) {
        if (!ctx->prof) return;
        VofeProfEntry e;
        e.name = name;
        if (level >= 0) e.name += " L" + std::to_string(level);
        if (hipEventCreate(&e.a) != hipSuccess) return;
        if (hipEventCreate(&e.b) != hipSuccess) { hipEventDestroy(e.a); return; }
        hipEventRecord(e.a, ctx->stream);
        ctx->prof->entries.push_back(e);
        idx = (long)ctx->prof->entries.size() - 1;
    }
    ~VofeKernelScope() { if (idx >= 0) hipEventRecord(c->prof->entries[idx].b, c->stream); }
};
#define VOFE_TIMED_CAT2(a, b) a##b
#define VOFE_TIMED_CAT(a, b) VOFE_TIMED_CAT2(a, b)
#define VOFE_TIMED(...) VofeKernelScope VOFE_TIMED_CAT(vofe_scope_, __LINE__)(__VA_ARGS__)

#define VOFE_HIP(call)                                              \
    do {                                                            \
        hipError_t e_ = (call);                                     \
        if (e_ != hipSuccess) return vofe_hip_fail(e_, #call);      \
    } while (0)

// kernels' host launchers (each returns VOFE_OK or an error code after hipGetLastError)
int vofe_launch_pyramid(vofe_ctx* ctx, vofe_pyr* pyr, int slot0, int count, const uint8_t* d_img,
                        int pitch, size_t img_stride);
int vofe_launch_lk(vofe_ctx* ctx, const vofe_pyr* prev, int prev_slot, const vofe_pyr* next,
                   int next_slot, const float* d_prev_pts, const int* d_img_idx, int n_images, float* d_next_pts,
                   uint8_t* d_status, float* d_err, int n, int max_count, double eps2, int flags,
                   float min_eig_thr, int fb, float fb_thresh);
// wide-window fallback (lk_wide.hip): windows the wavefront-per-keypoint kernels cannot hold, or every launch under VOFE_LK=wide
bool vofe_lk_needs_wide(int win_w, int win_h);
int vofe_launch_lk_wide(vofe_ctx* ctx, const vofe_pyr* prev, int prev_slot, const vofe_pyr* next, int next_slot,
                        const float* d_prev_pts, const int* d_img_idx, int n_images, float* d_next_pts, uint8_t* d_status,
                        float* d_err, int n, int max_count, double eps2, int flags, float min_eig_thr, int fb, float fb_thresh);
int vofe_launch_corner_score(vofe_ctx* ctx, const uint8_t* d_img, int batch, int h, int w, int pitch,
                             size_t img_stride, int harris, float k, float* d_score,
                             unsigned* d_max_keys, const uint8_t* d_mask, int block_size = 3, int aperture = 3);
int vofe_launch_new_feature_mask(vofe_ctx* ctx, const float* d_features_xy, int n_features,
                                 const float* d_existing_xy, int n_existing, const float* d_existing2_xy, int n_existing2,
                                 double threshold, uint8_t* d_mask);
int vofe_launch_pair_filter(vofe_ctx* ctx, const float* d_new_xy, const uint8_t* d_flag, const float* d_old_xy, int n,
                            double dist_move, float* d_new_out, float* d_old_out, int* d_count_out);
int vofe_launch_candidate_filter(vofe_ctx* ctx, const float* d_tracked_xy, const uint8_t* d_status, const void* d_first_xy,
                                 int first_is_f64, const double* d_transform, int tdim, int n, double dist_min,
                                 double dist_max, double dist_tri, float* d_tracked_out, void* d_first_out,
                                 double* d_transform_out, uint8_t* d_tri_out, int* d_count_out);
// current pose (3x4) and intrinsics (3x3) handed to a kernel by value: h_pk non-NULL replaces the device pointers
// d_K / d_pose (no upload, nothing to wait for)
struct VofePoseK { double pose[12]; double K[9]; };
int vofe_launch_triangulate(vofe_ctx* ctx, const double* d_K, const double* d_pose, const VofePoseK* h_pk, const float* d_cur_xy,
                            const void* d_first_xy, int first_is_f64, const double* d_transform, int tdim, const uint8_t* d_flag_in,
                            int n, double alpha_min, uint8_t* d_flag_out, double* d_xyz,
                            const float* copy_src = nullptr, float* copy_dst = nullptr, int copy_n = 0);
int vofe_launch_candidate_commit(vofe_ctx* ctx, float* d_C, void* d_F, int first_is_f64, double* d_T, int tdim, int m, int cap_c,
                                 const uint8_t* d_promote, const double* d_xyz, const float* d_features, const uint8_t* d_fmask,
                                 int nf, const double* d_pose, const VofePoseK* h_pk, float* d_P, int k, int cap_p, float* d_track_next,
                                 float* d_new_P, double* d_new_X, int* d_counts);
int vofe_gftt_impl(vofe_ctx* ctx, const uint8_t* d_img, int batch, int h, int w, int pitch,
                   size_t img_stride, int max_corners, double quality, double min_distance,
                   const uint8_t* d_mask, int harris, double k, float* d_corners_xy, int capacity,
                   int* h_n_out, float* h_corners_xy = nullptr, bool noalloc = false, int block_size = 3, int aperture = 3);

// ---- device helpers shared by kernels ------------------------------------------------------
__device__ __forceinline__ int vofe_reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}
