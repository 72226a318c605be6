// Pyramidal Lucas-Kanade tracker, wide-window fallback: one 256-thread workgroup per keypoint.
//
// cv2.calcOpticalFlowPyrLK takes any winSize; the wavefront-per-keypoint kernels of lk.hip map a window column to a lane and hold
// the template window in registers, which ends at 64 columns (and 64 rows per lane group).  Larger windows -- the reference uses
// none (constants.py:34,49,65: 21 / 11 / 21, initialization.py:186-188: 11 / 11 / 41) -- run here: the template window
// (I * 32, Ix, Iy as int16) lives in LDS, the threads stride over the window's pixels, the search image is
// read from the padded plane through the caches, and the window sums are reduced exactly (int64 partials: wavefront shuffles,
// then the four wavefront totals through LDS) so that every thread holds the same totals and the workgroup stays uniform.
// Arithmetic per SURVEY A.5 in the oracle's own statement order (oracle/vofe_oracle.cpp lk_point): integer stages exact, one
// int64 -> f32 rounding per window sum, float steps with -ffp-contract=off: bit-identical results, which the tests also check
// for small windows (VOFE_LK=wide sends every launch here).  A fallback: correctness first, no tuning.
#include <cmath>

#include "vofe_internal.h"

namespace {

constexpr int W_BITS = 14;
constexpr int NT = 256;

struct WideArgs {
    const VofeLevel* lv_a;
    const VofeLevel* lv_b;
    int n_levels;
    int slot_a, slot_b;
    const float2* prev_pts;
    const int* img_idx;
    int n_images;
    float2* next_pts;
    uint8_t* status;
    float* err;
    int n;
    int win_w, win_h;
    int max_count, flags, fb;
    float min_eig_thr, fb_thresh;
    double eps2;
    unsigned long long* counters;
};

__device__ __forceinline__ int floor_i(float v) {
    const int i = (int)__builtin_floorf(v);
    return __builtin_fabsf(v) < 1073741824.f ? i : INT_MIN;              // NaN / huge -> INT_MIN (fails every bounds test)
}
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

struct Weights { int w00, w01, w10, w11; };
__device__ __forceinline__ Weights bilinear_weights(float fx, float fy, int ix, int iy) {
    const float a = fx - (float)ix, b = fy - (float)iy;
    Weights q;
    q.w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    q.w01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    q.w10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    q.w11 = (1 << W_BITS) - q.w00 - q.w01 - q.w10;
    return q;
}

// exact sum over the workgroup; every thread returns with the same totals (two barriers per call)
template <int K>
__device__ __forceinline__ void block_sum(long long (&v)[K], long long* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    __syncthreads();                                 // the previous call's readers are done with `red`
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[k * 4 + wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = ((red[k * 4] + red[k * 4 + 1]) + red[k * 4 + 2]) + red[k * 4 + 3];
}

// bilinear sample of a u8 plane at window pixel (x, y) of a window whose integer origin is (ox, oy): weights * 4 taps
__device__ __forceinline__ int sample_u8(const uint8_t* __restrict__ P, int pitch, int ox, int oy, int x, int y, const Weights& q) {
    const uint8_t* r0 = P + (long long)(oy + y) * pitch + ox + x;
    const uint8_t* r1 = r0 + pitch;
    return (int)r0[0] * q.w00 + (int)r0[1] * q.w01 + (int)r1[0] * q.w10 + (int)r1[1] * q.w11;
}

__device__ void lk_one_wide(const WideArgs& a, const VofeLevel* __restrict__ LA, const VofeLevel* __restrict__ LB, int slotA,
                            int slotB, float px, float py, float& outx, float& outy, int& status, float& err, int flags,
                            bool want_err, short* tI, short* tX, short* tY, long long* red, unsigned& n_lv, unsigned& n_it) {
    const int w = a.win_w, h = a.win_h, npx_win = w * h;
    const int tid = threadIdx.x;
    const float hx = (float)(w - 1) * 0.5f, hy = (float)(h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int top = a.n_levels - 1;
    // (row, column) of a window pixel index walked with the index: one division per loop
    const int y_first = tid / w, x_first = tid - y_first * w, dy_step = NT / w, dx_step = NT - dy_step * w;
    for (int level = top; level >= 0; --level) {
        const VofeLevel A = LA[level];
        const VofeLevel B = LB[level];
        const float sc = __builtin_bit_cast(float, (127 - level) << 23);          // (float)(1. / (1 << level)), exactly
        float ppx = px * sc, ppy = py * sc;
        float npx, npy;
        if (level == top) {
            if (flags & 4) { npx = outx * sc; npy = outy * sc; }
            else { npx = ppx; npy = ppy; }
        } else { npx = outx * 2.f; npy = outy * 2.f; }
        outx = npx; outy = npy;
        ppx -= hx; ppy -= hy;
        const int ipx = floor_i(ppx), ipy = floor_i(ppy);
        if (ipx < -w || ipx >= A.w || ipy < -h || ipy >= A.h) {
            if (level == 0) { status = 0; err = 0.f; }
            continue;
        }
        ++n_lv;
        const uint8_t* Ia = A.I + (long long)slotA * A.slot_stride;
        const int* Da = A.D + (long long)slotA * A.slot_stride;
        const uint8_t* Jb = B.I + (long long)slotB * B.slot_stride;
        const Weights q = bilinear_weights(ppx, ppy, ipx, ipy);
        long long s3[3] = {0, 0, 0};
        __syncthreads();                             // the previous level's readers are done with the template
        for (int idx = tid, y = y_first, x = x_first; idx < npx_win; idx += NT) {
            const int iv = descale(sample_u8(Ia, A.pitch, ipx, ipy, x, y, q), W_BITS - 5);
            const int* d0 = Da + (long long)(ipy + y) * A.pitch + ipx + x;
            const int* d1 = d0 + A.pitch;
            const int p00 = d0[0], p01 = d0[1], p10 = d1[0], p11 = d1[1];         // (Ix | Iy << 16) per pixel
            const int ix = descale((int)(short)p00 * q.w00 + (int)(short)p01 * q.w01 + (int)(short)p10 * q.w10 + (int)(short)p11 * q.w11, W_BITS);
            const int iy = descale((p00 >> 16) * q.w00 + (p01 >> 16) * q.w01 + (p10 >> 16) * q.w10 + (p11 >> 16) * q.w11, W_BITS);
            tI[idx] = (short)iv; tX[idx] = (short)ix; tY[idx] = (short)iy;
            s3[0] += (long long)ix * ix; s3[1] += (long long)ix * iy; s3[2] += (long long)iy * iy;
            x += dx_step; y += dy_step;
            if (x >= w) { x -= w; ++y; }
        }
        block_sum<3>(s3, red);                       // (its barriers also publish the template)
        const float A11 = (float)s3[0] * FLT_SCALE, A12 = (float)s3[1] * FLT_SCALE, A22 = (float)s3[2] * FLT_SCALE;
        float Dd = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * w * h);
        if (flags & 8) err = minEig;
        if (minEig < a.min_eig_thr || Dd < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        Dd = 1.f / Dd;
        npx -= hx; npy -= hy;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; ++j) {
            const int inx = floor_i(npx), iny = floor_i(npy);
            if (inx < -w || inx >= B.w || iny < -h || iny >= B.h) {
                if (level == 0) status = 0;
                break;
            }
            ++n_it;
            const Weights r = bilinear_weights(npx, npy, inx, iny);
            long long b2s[2] = {0, 0};
            for (int idx = tid, y = y_first, x = x_first; idx < npx_win; idx += NT) {
                const int diff = descale(sample_u8(Jb, B.pitch, inx, iny, x, y, r), W_BITS - 5) - (int)tI[idx];
                b2s[0] += (long long)diff * (int)tX[idx];
                b2s[1] += (long long)diff * (int)tY[idx];
                x += dx_step; y += dy_step;
                if (x >= w) { x -= w; ++y; }
            }
            block_sum<2>(b2s, red);
            const float b1 = (float)b2s[0] * FLT_SCALE, b2 = (float)b2s[1] * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * Dd;
            const float dy = (A12 * b1 - A11 * b2) * Dd;
            npx += dx; npy += dy;
            outx = npx + hx; outy = npy + hy;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= a.eps2) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0 && !(flags & 8) && want_err) {
            const float ex = outx - hx, ey = outy - hy;
            const int inx = floor_i(ex), iny = floor_i(ey);
            if (inx < -w || inx >= B.w || iny < -h || iny >= B.h) { status = 0; continue; }
            const Weights r = bilinear_weights(ex, ey, inx, iny);
            long long se[1] = {0};
            for (int idx = tid, y = y_first, x = x_first; idx < npx_win; idx += NT) {
                const int diff = descale(sample_u8(Jb, B.pitch, inx, iny, x, y, r), W_BITS - 5) - (int)tI[idx];
                se[0] += diff < 0 ? -diff : diff;
                x += dx_step; y += dy_step;
                if (x >= w) { x -= w; ++y; }
            }
            block_sum<1>(se, red);
            err = (float)se[0] / (float)(32 * w * h);
        }
    }
}

__global__ __launch_bounds__(NT) void lk_track_kernel_wide(WideArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    long long* red = reinterpret_cast<long long*>(smem);                 // 3 sums x 4 wavefronts
    short* tI = reinterpret_cast<short*>(smem + 128);
    short* tX = tI + a.win_w * a.win_h;
    short* tY = tX + a.win_w * a.win_h;
    const int i = blockIdx.x;
    const float2 p = a.prev_pts[i];
    int idx = a.img_idx ? a.img_idx[i] : 0;
    const bool bad_idx = (unsigned)idx >= (unsigned)a.n_images;          // workgroup-uniform: never address a slot outside the batch
    if (bad_idx) idx = 0;
    float fx = 0.f, fy = 0.f;
    if (a.flags & 4) { const float2 q = a.next_pts[i]; fx = q.x; fy = q.y; }
    int st = bad_idx ? 0 : 1;
    float er = 0.f;
    float bx = 0.f, by = 0.f;
    const int ndir = bad_idx ? 0 : (a.fb ? 2 : 1);
    unsigned n_lv = 0, n_it = 0;
    for (int dir = 0; dir < ndir; ++dir) {            // dir 1 = backward pass from the forward result (continuous.py:150-151)
        const VofeLevel* LA = dir ? a.lv_b : a.lv_a;
        const VofeLevel* LB = dir ? a.lv_a : a.lv_b;
        const int sa = (dir ? a.slot_b : a.slot_a) + idx, sb = (dir ? a.slot_a : a.slot_b) + idx;
        const float qx = dir ? fx : p.x, qy = dir ? fy : p.y;
        float ox = dir ? 0.f : fx, oy = dir ? 0.f : fy, e = 0.f;
        int s = 1;
        lk_one_wide(a, LA, LB, sa, sb, qx, qy, ox, oy, s, e, dir ? 0 : a.flags, dir == 0, tI, tX, tY, red, n_lv, n_it);
        if (dir == 0) { fx = ox; fy = oy; st = s; er = e; }
        else { bx = ox; by = oy; }
        if (!st) break;                               // lost going forward: the backward pass cannot bring it back (continuous.py:158)
    }
    if (a.fb) {
        const float dx = fabsf(p.x - bx), dy = fabsf(p.y - by);
        float d = dx > dy ? dx : dy;
        if (dx != dx || dy != dy) d = __builtin_nanf("");
        st = st & (d < a.fb_thresh ? 1 : 0);
    }
    if (bad_idx) { st = 0; er = 0.f; }
    if (threadIdx.x == 0) {
        a.next_pts[i] = make_float2(fx, fy);
        a.status[i] = (uint8_t)st;
        a.err[i] = er;
        if (a.counters) {
            atomicAdd(a.counters + 0, 1ull);
            atomicAdd(a.counters + 1, (unsigned long long)n_lv);
            atomicAdd(a.counters + 2, (unsigned long long)n_it);
        }
    }
}

}  // namespace

// windows the wavefront-per-keypoint kernels of lk.hip cannot hold
bool vofe_lk_needs_wide(int win_w, int win_h) {
    if (win_w > 64) return true;
    int groups = 64 / win_w;
    if (groups > win_h) groups = win_h;
    return (win_h + groups - 1) / groups > 64;
}

int vofe_launch_lk_wide(vofe_ctx* ctx, const vofe_pyr* prev, int prev_slot, const vofe_pyr* next, int next_slot,
                        const float* d_prev_pts, const int* d_img_idx, int n_images, float* d_next_pts, uint8_t* d_status,
                        float* d_err, int n, int max_count, double eps2, int flags, float min_eig_thr, int fb, float fb_thresh) {
    WideArgs a;
    a.lv_a = prev->d_lv; a.lv_b = next->d_lv;
    a.n_levels = prev->levels;
    a.slot_a = prev_slot; a.slot_b = next_slot;
    a.prev_pts = reinterpret_cast<const float2*>(d_prev_pts);
    a.img_idx = d_img_idx;
    a.n_images = d_img_idx ? n_images : 1;
    a.next_pts = reinterpret_cast<float2*>(d_next_pts);
    a.status = d_status; a.err = d_err; a.n = n;
    a.win_w = prev->win_w; a.win_h = prev->win_h;
    a.max_count = max_count; a.flags = flags; a.fb = fb;
    a.min_eig_thr = min_eig_thr; a.fb_thresh = fb_thresh; a.eps2 = eps2;
    a.counters = ctx->lk_counters;
    const size_t lds = 128 + (size_t)6 * a.win_w * a.win_h;
    if (lds > 64 * 1024 - 256) {
        vofe_set_error("vofe_lk_track: winSize %dx%d needs %zu bytes of LDS for its template window (limit 64 KiB: about 104x104)",
                       a.win_w, a.win_h, lds);
        return VOFE_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(lk_track_kernel_wide, dim3(n), dim3(NT), lds, ctx->stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "lk_track_wide launch");
    return VOFE_OK;
}
