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
#include <cstring>

#include "vofe_internal.h"

namespace {

// one wavefront per feature: lanes stride over the existing points (coalesced 8-byte loads, the list stays in
// L2), then a butterfly minimum.  Small problems (1000 x 2000) would otherwise occupy a handful of CUs.
__global__ __launch_bounds__(256) void new_feature_mask_kernel(const float2* __restrict__ feat, int nf,
                                                               const float2* __restrict__ exist, int ne,
                                                               const float2* __restrict__ exist2, int ne2,
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
    for (int j = lane; j < ne2; j += 64) {                   // second set (landmarks kept apart from the candidates)
        const float2 e = exist2[j];
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


// ---- bootstrap pair filter (reference initialization.py:54-64, :69-70) ----------------------------------------
//     good_old = good_old[status.flatten() == 1];  good_new = tracked_points[status.flatten() == 1]
//     distances = np.sqrt(np.sum((good_new - good_old) ** 2, axis=2)).flatten()           # float32
//     drop = np.where(distances < dist_threshold_move);  good_old, good_new = np.delete(.., drop, axis=0)
// and, with flag = the RANSAC inlier mask and no distance rule, good_old[mask == 1] / good_new[mask == 1].
// A pair goes when its distance is BELOW the threshold, so a NaN distance stays (the candidate filter above drops it).
// Same single-block, order-keeping, in-place-safe compaction as candidate_filter_kernel.
__global__ __launch_bounds__(1024) void pair_filter_kernel(const float2* nw, const uint8_t* __restrict__ flag, const float2* old,
                                                           int n, float s_ge_move, float2* nw_out, float2* old_out,
                                                           int* __restrict__ count_out) {
    __shared__ int wave_count[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        bool keep = false;
        float2 a = make_float2(0.f, 0.f), b = a;
        if (i < n) {
            a = nw[i];
            b = old[i];
            const float dx = a.x - b.x, dy = a.y - b.y;
            const float s = dx * dx + dy * dy;
            keep = flag[i] == 1 && !(s < s_ge_move);
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
        if (keep) { nw_out[offset] = a; old_out[offset] = b; }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = base;
}


// ---- batched triangulation of matured candidates (reference continuous.py:450-485, 274-309) ---------------------
//     for idx with new_landmarks[idx]:
//         proj1 = K @ inv(Ti[:, idx].reshape(3, 4));  proj2 = K @ inv(pose)
//         X4 = cv2.triangulatePoints(proj1, proj2, Fi[:, idx], Ci[:, idx]);  X = X4[:3] / X4[3]
//         if X[2] < -t_idx[2]: drop        (t_idx = Ti[:3, 3] of that candidate)
//         alpha = arccos(clip(dot(unit(X - t_cur), unit(X - t_idx)), -1, 1));  keep iff |alpha| > alpha_min
// One thread per candidate, float64 throughout.  cv::triangulatePoints builds the 4x4 DLT matrix
// (x * P[2] - P[0], y * P[2] - P[1] per view) and takes the right singular vector of its smallest singular value
// from a one-sided Jacobi SVD; the same construction and the same kind of SVD run here, and the result is rounded
// to the type of the first-observation points (float32 / float64) exactly where OpenCV stores points4D in that
// type.  Rotation order and stopping differ from OpenCV's implementation, so this stage agrees with cv2 to
// rounding error of the SVD, not bit for bit ("parity unpinned" for this row; everything after X is the
// reference's own float64 numpy arithmetic, evaluated in the same order).
__device__ void invert_rt(const double* T, double* out) {        // T, out: 3x4 row-major; out = [R^T | -R^T t]
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) out[i * 4 + j] = T[j * 4 + i];
        out[i * 4 + 3] = (-T[0 * 4 + i]) * T[3] + (-T[1 * 4 + i]) * T[7] + (-T[2 * 4 + i]) * T[11];
    }
}
__device__ void k_times(const double* K, const double* M, double* P) {   // P (3x4) = K (3x3) @ M (3x4)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) P[i * 4 + j] = K[i * 3 + 0] * M[0 * 4 + j] + K[i * 3 + 1] * M[1 * 4 + j] + K[i * 3 + 2] * M[2 * 4 + j];
}

template <typename F>
__global__ __launch_bounds__(64) void triangulate_kernel(const double* __restrict__ Kmat, const double* __restrict__ pose, VofePoseK pk,
                                                         const float2* __restrict__ cur, const F* __restrict__ first,
                                                         const double* __restrict__ transform, int tdim,
                                                         const uint8_t* __restrict__ flag_in, int n, double alpha_min,
                                                         uint8_t* __restrict__ flag_out, double* __restrict__ xyz,
                                                         const float2* __restrict__ copy_src, float2* __restrict__ copy_dst, int copy_n,
                                                         int tri_blocks) {
    if ((int)blockIdx.x >= tri_blocks) {
        // rider of the frame-state object's phase 2: the surviving landmarks, which the host compacted into pinned memory
        // that this device addresses, are fetched by these extra blocks instead of by a copy command in front of the launch
        const int c = ((int)blockIdx.x - tri_blocks) * 64 + (int)threadIdx.x;
        if (c < copy_n) copy_dst[c] = copy_src[c];
        return;
    }
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    if (!flag_in[i]) { flag_out[i] = 0; return; }
    double K[9], T2[12], T1[12], M[12], P1[12], P2[12];
    // pose and intrinsics from device memory, or (both pointers null) from the kernel arguments
#pragma unroll
    for (int k = 0; k < 9; ++k) K[k] = Kmat ? Kmat[k] : pk.K[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) { T2[k] = pose ? pose[k] : pk.pose[k]; T1[k] = transform[(size_t)i * tdim + k]; }
    invert_rt(T1, M); k_times(K, M, P1);
    invert_rt(T2, M); k_times(K, M, P2);
    const double x1 = (double)first[2 * (size_t)i], y1 = (double)first[2 * (size_t)i + 1];
    const double x2 = (double)cur[i].x, y2 = (double)cur[i].y;
    // A (4x4): columns are orthogonalised in place (one-sided Jacobi), V accumulates the rotations
    double A[4][4], V[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A[0][k] = x1 * P1[8 + k] - P1[k];
        A[1][k] = y1 * P1[8 + k] - P1[4 + k];
        A[2][k] = x2 * P2[8 + k] - P2[k];
        A[3][k] = y2 * P2[8 + k] - P2[4 + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) V[k][j] = (k == j) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool changed = false;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double a = 0, b = 0, c = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a += A[r][p] * A[r][p]; b += A[r][q] * A[r][q]; c += A[r][p] * A[r][q]; }
                if (fabs(c) <= 2.220446049250313e-16 * sqrt(a * b)) continue;
                changed = true;
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double ap = A[r][p], aq = A[r][q];
                    A[r][p] = cs * ap - sn * aq; A[r][q] = sn * ap + cs * aq;
                    const double vp = V[r][p], vq = V[r][q];
                    V[r][p] = cs * vp - sn * vq; V[r][q] = sn * vp + cs * vq;
                }
            }
        if (!changed) break;
    }
    int best = 0;
    double bn = INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double nn = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) nn += A[r][c] * A[r][c];
        if (nn < bn) { bn = nn; best = c; }
    }
    double v4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v4[r] = best == 0 ? V[r][0] : (best == 1 ? V[r][1] : (best == 2 ? V[r][2] : V[r][3]));
    // points4D is stored in the point type, the division happens in that type (point4D_hom[:3] / point4D_hom[3])
    const F h0 = (F)v4[0], h1 = (F)v4[1], h2 = (F)v4[2], h3 = (F)v4[3];
    const double X = (double)(h0 / h3), Y = (double)(h1 / h3), Z = (double)(h2 / h3);
    xyz[3 * (size_t)i] = X; xyz[3 * (size_t)i + 1] = Y; xyz[3 * (size_t)i + 2] = Z;
    bool keep = !(Z < -T1[11]);
    if (keep) {
        double c0 = X - T2[3], c1 = Y - T2[7], c2 = Z - T2[11];
        double f0 = X - T1[3], f1 = Y - T1[7], f2 = Z - T1[11];
        const double nc = sqrt(c0 * c0 + c1 * c1 + c2 * c2), nf = sqrt(f0 * f0 + f1 * f1 + f2 * f2);
        c0 /= nc; c1 /= nc; c2 /= nc; f0 /= nf; f1 /= nf; f2 /= nf;
        double d = c0 * f0 + c1 * f1 + c2 * f2;
        d = d < -1.0 ? -1.0 : (d > 1.0 ? 1.0 : d);           // np.clip; a NaN stays NaN and fails the test below
        keep = fabs(acos(d)) > alpha_min;
    }
    flag_out[i] = keep ? 1 : 0;
}

// ---- tail of _handle_candidates on the device (reference continuous.py:476-495) -------------------------------------
//     Pi = np.hstack([Pi, Ci[:, new_landmarks]]);  Xi = np.hstack([Xi, points3D])          # promoted candidates
//     Ci, Fi, Ti = Ci[:, no_new_landmark], Fi[:, no_new_landmark], Ti[:, no_new_landmark]
//     Ci = np.hstack([Ci, new_features]);  Fi = np.hstack([Fi, new_features]);  Ti = np.hstack([Ti, tiled_pose])
// One block, order-preserving throughout (numpy's boolean indexing keeps order): the candidates are walked in chunks of
// 1024 with ballot / popcount offsets; a chunk's loads have returned before its first store and stores only go to
// positions at or before the chunk's own, so C / F / T are compacted in place.  The promoted points and their 3-D
// positions also go to new_P / new_X (which may be pinned host memory mapped into the device: the host appends them to
// its own landmark arrays), the next frame's tracking input [landmarks | candidates] is laid out in track_next, and
// counts = {promoted, candidates out, landmarks out, overflow} tells the host what happened.
template <typename F>
__global__ __launch_bounds__(1024) void candidate_commit_kernel(float2* C, F* Fp, double* T, int tdim, int m, int cap_c,
                                                                const uint8_t* __restrict__ promote, const double* __restrict__ xyz,
                                                                const float2* __restrict__ feat, const uint8_t* __restrict__ fmask, int nf,
                                                                const double* __restrict__ pose, VofePoseK pk, float2* P, int k, int cap_p,
                                                                float2* __restrict__ track_next, float2* __restrict__ new_P,
                                                                double* __restrict__ new_X, int* __restrict__ counts) {
    __shared__ int wave_a[16], wave_b[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int n_keep = 0, n_prom = 0, overflow = 0;
    for (int c0 = 0; c0 < m; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        bool keep = false, prom = false;
        float2 c = make_float2(0.f, 0.f);
        F fx = 0, fy = 0;
        double t[16], X[3] = {0, 0, 0};
        if (i < m) {
            prom = promote[i] != 0;
            keep = !prom;
            c = C[i];
            if (keep) {
                fx = Fp[2 * (size_t)i]; fy = Fp[2 * (size_t)i + 1];
                for (int q = 0; q < 16; ++q) if (q < tdim) t[q] = T[(size_t)i * tdim + q];
            } else {
                X[0] = xyz[3 * (size_t)i]; X[1] = xyz[3 * (size_t)i + 1]; X[2] = xyz[3 * (size_t)i + 2];
            }
        }
        const unsigned long long bk = __ballot(keep), bp = __ballot(prom);
        if (lane == 0) { wave_a[wave] = __popcll(bk); wave_b[wave] = __popcll(bp); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the chunk's loads have returned before anyone stores
        __syncthreads();
        int ok = n_keep + __popcll(bk & ((1ull << lane) - 1)), op = n_prom + __popcll(bp & ((1ull << lane) - 1)), tk = 0, tp = 0;
        for (int q = 0; q < 16; ++q) {
            ok += q < wave ? wave_a[q] : 0; tk += wave_a[q];
            op += q < wave ? wave_b[q] : 0; tp += wave_b[q];
        }
        if (keep) {
            C[ok] = c;
            Fp[2 * (size_t)ok] = fx; Fp[2 * (size_t)ok + 1] = fy;
            for (int q = 0; q < 16; ++q) if (q < tdim) T[(size_t)ok * tdim + q] = t[q];
        }
        if (prom) {
            new_P[op] = c;
            new_X[3 * (size_t)op] = X[0]; new_X[3 * (size_t)op + 1] = X[1]; new_X[3 * (size_t)op + 2] = X[2];
            if (k + op < cap_p) P[k + op] = c; else overflow = 1;
        }
        n_keep += tk; n_prom += tp;
        __syncthreads();
    }
    // new features: C = F = the detection, T = the current pose (continuous.py:487-493)
    double pz[16];
    for (int q = 0; q < 16; ++q) pz[q] = q < 12 && q < tdim ? (pose ? pose[q] : pk.pose[q]) : 0.0;
    int n_c = n_keep;
    for (int c0 = 0; c0 < nf; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        const bool add = i < nf && fmask[i] != 0;
        const unsigned long long ba = __ballot(add);
        if (lane == 0) wave_a[wave] = __popcll(ba);
        __syncthreads();
        int o = n_c + __popcll(ba & ((1ull << lane) - 1)), tot = 0;
        for (int q = 0; q < 16; ++q) { o += q < wave ? wave_a[q] : 0; tot += wave_a[q]; }
        if (add) {
            if (o < cap_c) {
                const float2 f = feat[i];
                C[o] = f;
                Fp[2 * (size_t)o] = (F)f.x; Fp[2 * (size_t)o + 1] = (F)f.y;
                for (int q = 0; q < 16; ++q) if (q < tdim) T[(size_t)o * tdim + q] = pz[q];
            } else overflow = 1;
        }
        n_c += tot;
        __syncthreads();
    }
    const int k_out = min(k + n_prom, cap_p), m_out = min(n_c, cap_c);
    overflow = __syncthreads_or(overflow || k + n_prom > cap_p || n_c > cap_c);
    // next frame's tracking input: [landmarks | candidates]
    for (int i = threadIdx.x; i < k_out; i += 1024) track_next[i] = P[i];
    for (int i = threadIdx.x; i < m_out; i += 1024) track_next[k_out + i] = C[i];
    if (threadIdx.x == 0) { counts[0] = n_prom; counts[1] = m_out; counts[2] = k_out; counts[3] = overflow; }
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

int vofe_launch_pair_filter(vofe_ctx* ctx, const float* d_new_xy, const uint8_t* d_flag, const float* d_old_xy, int n,
                            double dist_move, float* d_new_out, float* d_old_out, int* d_count_out) {
    // numpy compares the float32 distances with the Python scalar in float32
    const float s_move = square_threshold_ge<float>((float)dist_move);
    hipLaunchKernelGGL(pair_filter_kernel, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<const float2*>(d_new_xy), d_flag,
                       reinterpret_cast<const float2*>(d_old_xy), n, s_move, reinterpret_cast<float2*>(d_new_out),
                       reinterpret_cast<float2*>(d_old_out), d_count_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "pair_filter launch");
    return VOFE_OK;
}

int vofe_launch_new_feature_mask(vofe_ctx* ctx, const float* d_features_xy, int n_features,
                                 const float* d_existing_xy, int n_existing, const float* d_existing2_xy, int n_existing2,
                                 double threshold, uint8_t* d_mask) {
    if (n_features == 0) return VOFE_OK;
    const double s_star = square_threshold<double>(threshold);
    hipLaunchKernelGGL(new_feature_mask_kernel, dim3((n_features + 3) / 4), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float2*>(d_features_xy), n_features,
                       reinterpret_cast<const float2*>(d_existing_xy), n_existing,
                       reinterpret_cast<const float2*>(d_existing2_xy), n_existing2, s_star, d_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "new_feature_mask launch");
    return VOFE_OK;
}

int vofe_launch_triangulate(vofe_ctx* ctx, const double* d_K, const double* d_pose, const VofePoseK* h_pk, const float* d_cur_xy,
                            const void* d_first_xy, int first_is_f64, const double* d_transform, int tdim, const uint8_t* d_flag_in,
                            int n, double alpha_min, uint8_t* d_flag_out, double* d_xyz, const float* copy_src, float* copy_dst,
                            int copy_n) {
    if (!copy_src || !copy_dst || copy_n < 0) copy_n = 0;
    if (n == 0 && copy_n == 0) return VOFE_OK;
    const int tri_blocks = (n + 63) / 64;
    const dim3 grid(tri_blocks + (copy_n + 63) / 64), block(64);
    VofePoseK pk;
    if (h_pk) { pk = *h_pk; d_K = nullptr; d_pose = nullptr; } else memset(&pk, 0, sizeof(pk));
    const float2* cs = reinterpret_cast<const float2*>(copy_src);
    float2* cd = reinterpret_cast<float2*>(copy_dst);
    if (first_is_f64)
        hipLaunchKernelGGL(triangulate_kernel<double>, grid, block, 0, ctx->stream, d_K, d_pose, pk, reinterpret_cast<const float2*>(d_cur_xy),
                           static_cast<const double*>(d_first_xy), d_transform, tdim, d_flag_in, n, alpha_min, d_flag_out, d_xyz,
                           cs, cd, copy_n, tri_blocks);
    else
        hipLaunchKernelGGL(triangulate_kernel<float>, grid, block, 0, ctx->stream, d_K, d_pose, pk, reinterpret_cast<const float2*>(d_cur_xy),
                           static_cast<const float*>(d_first_xy), d_transform, tdim, d_flag_in, n, alpha_min, d_flag_out, d_xyz,
                           cs, cd, copy_n, tri_blocks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "triangulate launch");
    return VOFE_OK;
}

int vofe_launch_candidate_commit(vofe_ctx* ctx, float* d_C, void* d_F, int first_is_f64, double* d_T, int tdim, int m, int cap_c,
                                 const uint8_t* d_promote, const double* d_xyz, const float* d_features, const uint8_t* d_fmask,
                                 int nf, const double* d_pose, const VofePoseK* h_pk, float* d_P, int k, int cap_p, float* d_track_next,
                                 float* d_new_P, double* d_new_X, int* d_counts) {
    VofePoseK pk;
    if (h_pk) { pk = *h_pk; d_pose = nullptr; } else memset(&pk, 0, sizeof(pk));
    if (first_is_f64)
        hipLaunchKernelGGL(candidate_commit_kernel<double>, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<float2*>(d_C),
                           static_cast<double*>(d_F), d_T, tdim, m, cap_c, d_promote, d_xyz, reinterpret_cast<const float2*>(d_features),
                           d_fmask, nf, d_pose, pk, reinterpret_cast<float2*>(d_P), k, cap_p, reinterpret_cast<float2*>(d_track_next),
                           reinterpret_cast<float2*>(d_new_P), d_new_X, d_counts);
    else
        hipLaunchKernelGGL(candidate_commit_kernel<float>, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<float2*>(d_C),
                           static_cast<float*>(d_F), d_T, tdim, m, cap_c, d_promote, d_xyz, reinterpret_cast<const float2*>(d_features),
                           d_fmask, nf, d_pose, pk, reinterpret_cast<float2*>(d_P), k, cap_p, reinterpret_cast<float2*>(d_track_next),
                           reinterpret_cast<float2*>(d_new_P), d_new_X, d_counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "candidate_commit launch");
    return VOFE_OK;
}
