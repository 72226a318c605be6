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

}  // namespace

// smallest double x >= 0 with sqrt(x) > t (host sqrt is correctly rounded); NaN when no finite distance can pass
static double square_threshold(double t) {
    if (t != t || t == INFINITY) return NAN;
    if (t < 0) return 0.0;                                   // every distance (>= 0) exceeds a negative threshold
    double x = t * t;
    if (!(x < INFINITY)) return INFINITY;                    // only an infinite distance exceeds it
    while (x > 0 && std::sqrt(std::nextafter(x, 0.0)) > t) x = std::nextafter(x, 0.0);
    while (!(std::sqrt(x) > t)) x = std::nextafter(x, INFINITY);
    return x;
}

int vofe_launch_new_feature_mask(vofe_ctx* ctx, const float* d_features_xy, int n_features,
                                 const float* d_existing_xy, int n_existing, double threshold,
                                 uint8_t* d_mask) {
    if (n_features == 0) return VOFE_OK;
    const double s_star = square_threshold(threshold);
    hipLaunchKernelGGL(new_feature_mask_kernel, dim3((n_features + 3) / 4), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float2*>(d_features_xy), n_features,
                       reinterpret_cast<const float2*>(d_existing_xy), n_existing, s_star, d_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "new_feature_mask launch");
    return VOFE_OK;
}
