// Candidate bookkeeping on the device: the new-feature de-duplication of
// reference continuous.py:408-416
//     distance_matrix = cdist(features_current, existing_features)      # scipy, float64 Euclidean
//     min_distances   = np.min(distance_matrix, axis=1)
//     new_feature_mask = min_distances > Constants.THRESHOLD_NEW_KEYPOINTS
// as one kernel that never materialises the (N x M) matrix.
//
// Exactness.  scipy computes d = sqrt(dx*dx + dy*dy) in float64 (one rounding per operation, no
// contraction) and compares d > T.  sqrt is monotone and correctly rounded, so
//     min_j d_j > T   <=>   min_j s_j >= s*,   s_j = dx*dx + dy*dy,
// where s* is the smallest double whose correctly rounded square root exceeds T.  The host finds s*
// by stepping through neighbouring doubles around T*T; the kernel only multiplies, adds, takes a
// minimum and compares, in float64, in scipy's operation order.  A NaN distance makes np.min NaN and
// the comparison false; the kernel tracks that separately.
#include <cmath>

#include "vofe_internal.h"

namespace {

// one wavefront per feature: lanes stride over the existing points (coalesced 8-byte loads, the list stays in
// L2), then a butterfly minimum.  Small problems (1000 x 2000) would otherwise occupy a handful of CUs.
__global__ __launch_bounds__(256) void new_feature_mask_kernel(const float2* __restrict__ feat, int nf,
                                                               const float2* __restrict__ exist, int ne,
                                                               double s_star, uint8_t* __restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nf) return;                                     // whole wave
    const float2 f = feat[i];
    const double fx = (double)f.x, fy = (double)f.y;
    double m = INFINITY;
    bool bad = false;
    for (int j = lane; j < ne; j += 64) {
        const float2 e = exist[j];
        const double dx = fx - (double)e.x, dy = fy - (double)e.y;
        const double s = dx * dx + dy * dy;
        bad |= (s != s);
        m = s < m ? s : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(m, o);
        m = other < m ? other : m;
    }
    if (lane == 0) mask[i] = (!__any(bad) && m >= s_star) ? 1 : 0;
}

// ---- status compaction + travel-distance filter (reference continuous.py:391-405, 446-448) ----------------
//     valid_idx = np.where(status == 1)[0];  Ci = tracked[:, valid_idx]; Fi = Fi[:, valid_idx]; Ti = Ti[:, valid_idx]
//     distances = np.sqrt(np.sum((Ci - Fi) ** 2, axis=0))
//     good = (distances > MIN) & (distances < MAX);   Ci, Fi, Ti = Ci[:, good], Fi[:, good], Ti[:, good]
//     new_landmarks = np.sqrt(np.sum((Ci - Fi) ** 2, axis=0)) > TRIANGULATION
// numpy evaluates this in float32 when Fi is float32 and in float64 when it is float64 (Ci is always the float32 LK
// output); F is that type.  As above the comparisons are made on the squared distance against thresholds the host
// derives from the correctly rounded square root: d > t <=> s >= s_gt(t), d < t <=> s < s_ge(t).  One block walks
// the list in chunks of 1024 and keeps the input order; every load of a chunk has returned before the first store of
// that chunk is issued, and stores only go to positions at or before the chunk's own, so each output may be the very
// buffer its input came from.
template <typename F>
__global__ __launch_bounds__(1024) void candidate_filter_kernel(const float2* cur, const uint8_t* __restrict__ status,
                                                                const F* first, const double* transform, int tdim, int n,
                                                                F s_gt_min, F s_ge_max, F s_gt_tri, float2* cur_out,
                                                                F* first_out, double* transform_out,
                                                                uint8_t* __restrict__ tri_out, int* __restrict__ count_out) {
    __shared__ int wave_count[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        bool keep = false;
        float2 c = make_float2(0.f, 0.f);
        F fx = 0, fy = 0, s = 0;
        double t[16];
        if (i < n) {
            c = cur[i];
            fx = first[2 * (size_t)i];
            fy = first[2 * (size_t)i + 1];
            const F dx = (F)c.x - fx, dy = (F)c.y - fy;
            s = dx * dx + dy * dy;
            keep = status[i] == 1 && s >= s_gt_min && s < s_ge_max;
            if (keep && transform)
                for (int k = 0; k < 16; ++k)
                    if (k < tdim) t[k] = transform[(size_t)i * tdim + k];
        }
        const unsigned long long ballot = __ballot(keep);
        if (lane == 0) wave_count[wave] = __popcll(ballot);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the chunk's loads have returned before anyone stores
        __syncthreads();
        int offset = base + __popcll(ballot & ((1ull << lane) - 1)), total = 0;
        for (int k = 0; k < 16; ++k) {
            const int cnt = wave_count[k];
            offset += k < wave ? cnt : 0;
            total += cnt;
        }
        if (keep) {
            cur_out[offset] = c;
            first_out[2 * (size_t)offset] = fx;
            first_out[2 * (size_t)offset + 1] = fy;
            tri_out[offset] = s >= s_gt_tri ? 1 : 0;
            if (transform)
                for (int k = 0; k < 16; ++k)
                    if (k < tdim) transform_out[(size_t)offset * tdim + k] = t[k];
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = base;
}

}  // namespace

// smallest x >= 0 with sqrt(x) > t (host sqrt is correctly rounded); NaN when no distance can pass
template <typename F>
static F square_threshold(F t) {
    if (t != t || t == (F)INFINITY) return (F)NAN;
    if (t < 0) return 0;                                     // every distance (>= 0) exceeds a negative threshold
    F x = t * t;
    if (!(x < (F)INFINITY)) return (F)INFINITY;              // only an infinite distance exceeds it
    while (x > 0 && std::sqrt(std::nextafter(x, (F)0)) > t) x = std::nextafter(x, (F)0);
    while (!(std::sqrt(x) > t)) x = std::nextafter(x, (F)INFINITY);
    return x;
}

// smallest x >= 0 with sqrt(x) >= t, so that d < t <=> s < x; 0 when no distance is below t
template <typename F>
static F square_threshold_ge(F t) {
    if (t != t || t <= 0) return 0;
    if (t == (F)INFINITY) return (F)INFINITY;
    F x = t * t;
    if (!(x < (F)INFINITY)) return (F)INFINITY;              // every finite s has sqrt(s) < t
    while (x > 0 && std::sqrt(std::nextafter(x, (F)0)) >= t) x = std::nextafter(x, (F)0);
    while (!(std::sqrt(x) >= t)) x = std::nextafter(x, (F)INFINITY);
    return x;
}

template <typename F>
static void launch_candidate_filter(vofe_ctx* ctx, const float* d_tracked_xy, const uint8_t* d_status, const void* d_first_xy,
                                    const double* d_transform, int tdim, int n, double dist_min, double dist_max,
                                    double dist_tri, float* d_tracked_out, void* d_first_out, double* d_transform_out,
                                    uint8_t* d_tri_out, int* d_count_out) {
    // numpy compares a float32 array with a Python scalar in float32 (the scalar is converted), a float64 one in float64
    const F s_min = square_threshold<F>((F)dist_min), s_max = square_threshold_ge<F>((F)dist_max),
            s_tri = square_threshold<F>((F)dist_tri);
    hipLaunchKernelGGL(candidate_filter_kernel<F>, dim3(1), dim3(1024), 0, ctx->stream,
                       reinterpret_cast<const float2*>(d_tracked_xy), d_status, static_cast<const F*>(d_first_xy), d_transform,
                       tdim, n, s_min, s_max, s_tri, reinterpret_cast<float2*>(d_tracked_out), static_cast<F*>(d_first_out),
                       d_transform_out, d_tri_out, d_count_out);
}

int vofe_launch_candidate_filter(vofe_ctx* ctx, const float* d_tracked_xy, const uint8_t* d_status, const void* d_first_xy,
                                 int first_is_f64, const double* d_transform, int tdim, int n, double dist_min,
                                 double dist_max, double dist_tri, float* d_tracked_out, void* d_first_out,
                                 double* d_transform_out, uint8_t* d_tri_out, int* d_count_out) {
    if (first_is_f64)
        launch_candidate_filter<double>(ctx, d_tracked_xy, d_status, d_first_xy, d_transform, tdim, n, dist_min, dist_max,
                                        dist_tri, d_tracked_out, d_first_out, d_transform_out, d_tri_out, d_count_out);
    else
        launch_candidate_filter<float>(ctx, d_tracked_xy, d_status, d_first_xy, d_transform, tdim, n, dist_min, dist_max,
                                       dist_tri, d_tracked_out, d_first_out, d_transform_out, d_tri_out, d_count_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "candidate_filter launch");
    return VOFE_OK;
}

int vofe_launch_new_feature_mask(vofe_ctx* ctx, const float* d_features_xy, int n_features,
                                 const float* d_existing_xy, int n_existing, double threshold,
                                 uint8_t* d_mask) {
    if (n_features == 0) return VOFE_OK;
    const double s_star = square_threshold<double>(threshold);
    hipLaunchKernelGGL(new_feature_mask_kernel, dim3((n_features + 3) / 4), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float2*>(d_features_xy), n_features,
                       reinterpret_cast<const float2*>(d_existing_xy), n_existing, s_star, d_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "new_feature_mask launch");
    return VOFE_OK;
}
