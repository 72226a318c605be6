// Gaussian pyramid + Scharr derivative kernels for gfx950 (HBM-bound streaming stencils).
//
// Replaces cv::buildOpticalFlowPyramid / cv::pyrDown (8U) / cv::calcScharrDeriv, which
// cv2.calcOpticalFlowPyrLK runs at reference continuous.py:145-146,150-151 and
// initialization.py:51-52.  Arithmetic per SURVEY.md A.3 / A.4 (integer, exact).
//
// Layout: every level lives in a plane padded by (pad_x, pad_y) >= (win_w, win_h); pixel (0,0)
// is 16-byte aligned and rows are `pitch` (multiple of 64) pixels apart.  The REFLECT_101 border
// that OpenCV materialises with copyMakeBorder is written by the same kernel that produces the
// interior, so pyrDown / Scharr / LK never need index arithmetic at image edges.
#include "vofe_internal.h"

namespace {

struct U4 { unsigned a, b, c, d; } __attribute__((packed, aligned(4)));
struct U2 { unsigned a, b; } __attribute__((packed, aligned(4)));

// ---------------------------------------------------------------------------------------------
// level 0: copy the caller's image into the padded plane and fill the REFLECT_101 border.
// One thread = 16 consecutive bytes of one padded row.
__global__ __launch_bounds__(256) void pad_copy_level0(const uint8_t* __restrict__ src, int spitch,
                                                       size_t img_stride, VofeLevel L, int slot0,
                                                       int win_w, int win_h, int gx0, int ngx) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngx) return;
    const int y = (int)blockIdx.y - win_h;
    const int x0 = (g + gx0) * 16;
    const uint8_t* s = src + (size_t)blockIdx.z * img_stride;
    uint8_t* d = L.I + (long long)(slot0 + (int)blockIdx.z) * L.slot_stride + (long long)y * L.pitch + x0;
    const int sy = vofe_reflect101(y, L.h);
    const uint8_t* srow = s + (size_t)sy * spitch;
    uint4 v;
    if (x0 >= 0 && x0 + 16 <= L.w) {
        __builtin_memcpy(&v, srow + x0, 16);
    } else {
        uint8_t b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = srow[vofe_reflect101(x0 + i, L.w)];
        __builtin_memcpy(&v, b, 16);
    }
    *reinterpret_cast<uint4*>(d) = v;
}

// ---------------------------------------------------------------------------------------------
// pyrDown: src level (padded, border valid) -> dst level interior + border.
// One thread = 8 consecutive output pixels of one padded output row.
__device__ __forceinline__ unsigned byte_of(const unsigned* w, int i) { return (w[i >> 2] >> ((i & 3) * 8)) & 0xffu; }

__global__ __launch_bounds__(256) void pyr_down_u8(VofeLevel S, VofeLevel Dl, int slot0, int win_w,
                                                   int win_h, int gx0, int ngx) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngx) return;
    const int Y = (int)blockIdx.y - win_h;
    const int X0 = (g + gx0) * 8;
    const long long so = (long long)(slot0 + (int)blockIdx.z) * S.slot_stride;
    const long long dofs = (long long)(slot0 + (int)blockIdx.z) * Dl.slot_stride;
    const uint8_t* sp = S.I + so;
    uint8_t* d = Dl.I + dofs + (long long)Y * Dl.pitch + X0;
    unsigned out[8];
    if (X0 >= 0 && X0 + 8 <= Dl.w && Y >= 0 && Y < Dl.h) {
        int acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 128;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int wk = (k == 0 || k == 4) ? 1 : (k == 2 ? 6 : 4);
            const uint8_t* r = sp + (long long)(2 * Y - 2 + k) * S.pitch + 2 * X0 - 4;
            unsigned wv[6];
            U4 a = *reinterpret_cast<const U4*>(r);
            U2 b = *reinterpret_cast<const U2*>(r + 16);
            wv[0] = a.a; wv[1] = a.b; wv[2] = a.c; wv[3] = a.d; wv[4] = b.a; wv[5] = b.b;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 4 + 2 * j;
                int hs = (int)(byte_of(wv, c - 2) + byte_of(wv, c + 2)) + 4 * (int)(byte_of(wv, c - 1) + byte_of(wv, c + 1)) +
                         6 * (int)byte_of(wv, c);
                acc[j] += wk * hs;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = (unsigned)(acc[j] >> 8);
    } else {
        const int ry = vofe_reflect101(Y, Dl.h);
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            const int rx = vofe_reflect101(X0 + j, Dl.w);
            int acc = 128;
            for (int k = 0; k < 5; ++k) {
                const int wk = (k == 0 || k == 4) ? 1 : (k == 2 ? 6 : 4);
                const uint8_t* r = sp + (long long)(2 * ry - 2 + k) * S.pitch + 2 * rx;
                acc += wk * ((int)r[-2] + (int)r[2] + 4 * ((int)r[-1] + (int)r[1]) + 6 * (int)r[0]);
            }
            out[j] = (unsigned)(acc >> 8);
        }
    }
    uint2 o;
    o.x = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24);
    o.y = out[4] | (out[5] << 8) | (out[6] << 16) | (out[7] << 24);
    *reinterpret_cast<uint2*>(d) = o;
}

// ---------------------------------------------------------------------------------------------
// Scharr derivative of every level in one launch.  One thread = 8 pixels of one row; a block is
// 64 groups x 4 rows.  The source border is valid (REFLECT_101), the destination border stays 0.
struct ScharrPlan {
    int blk_start[VOFE_MAX_LEVELS + 1];
    int blk_x[VOFE_MAX_LEVELS];
    int levels;
};

__global__ __launch_bounds__(256) void scharr_deriv_s16x2(const VofeLevel* __restrict__ lv, ScharrPlan plan, int slot0) {
    int level = 0;
    while (level + 1 < plan.levels && (int)blockIdx.x >= plan.blk_start[level + 1]) ++level;
    const VofeLevel L = lv[level];
    const int b = (int)blockIdx.x - plan.blk_start[level];
    const int bx = b % plan.blk_x[level], by = b / plan.blk_x[level];
    const int x0 = (bx * 64 + (threadIdx.x & 63)) * 8;
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x0 >= L.w || y >= L.h) return;
    const long long so = (long long)(slot0 + (int)blockIdx.y) * L.slot_stride;
    const uint8_t* r1 = L.I + so + (long long)y * L.pitch + x0;
    const uint8_t* r0 = r1 - L.pitch;
    const uint8_t* r2 = r1 + L.pitch;
    int t0[10], t1[10];
    {
        unsigned a[4], bq[4], c[4];
        U4 va = *reinterpret_cast<const U4*>(r0 - 4), vb = *reinterpret_cast<const U4*>(r1 - 4), vc = *reinterpret_cast<const U4*>(r2 - 4);
        a[0] = va.a; a[1] = va.b; a[2] = va.c; a[3] = va.d;
        bq[0] = vb.a; bq[1] = vb.b; bq[2] = vb.c; bq[3] = vb.d;
        c[0] = vc.a; c[1] = vc.b; c[2] = vc.c; c[3] = vc.d;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int pa = (int)byte_of(a, i + 3), pb = (int)byte_of(bq, i + 3), pc = (int)byte_of(c, i + 3);
            t0[i] = 3 * (pa + pc) + 10 * pb;
            t1[i] = pc - pa;
        }
    }
    unsigned o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ix = t0[i + 2] - t0[i];
        const int iy = 3 * (t1[i] + t1[i + 2]) + 10 * t1[i + 1];
        o[i] = ((unsigned)ix & 0xffffu) | ((unsigned)iy << 16);
    }
    int* d = L.D + so + (long long)y * L.pitch + x0;
    // the last group of a row may run past w: those columns lie in the zero border, keep them 0
    if (x0 + 8 <= L.w) {
        *reinterpret_cast<uint4*>(d) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(d + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
        for (int i = 0; i < 8 && x0 + i < L.w; ++i) d[i] = (int)o[i];
    }
}

inline int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }   // floor division

}  // namespace

int vofe_launch_pyramid(vofe_ctx* ctx, vofe_pyr* pyr, int slot0, int count, const uint8_t* d_img,
                        int pitch, size_t img_stride) {
    hipStream_t st = ctx->stream;
    const int ww = pyr->win_w, wh = pyr->win_h;
    {
        const VofeLevel& L = pyr->lv[0];
        const int gx0 = fdiv(-ww, 16), gx1 = fdiv(L.w + ww - 1, 16);
        const int ngx = gx1 - gx0 + 1;
        dim3 grid((ngx + 255) / 256, L.h + 2 * wh, count);
        hipLaunchKernelGGL(pad_copy_level0, grid, dim3(256), 0, st, d_img, pitch, img_stride, L, slot0, ww, wh, gx0, ngx);
    }
    for (int l = 1; l < pyr->levels; ++l) {
        const VofeLevel& S = pyr->lv[l - 1];
        const VofeLevel& D = pyr->lv[l];
        const int gx0 = fdiv(-ww, 8), gx1 = fdiv(D.w + ww - 1, 8);
        const int ngx = gx1 - gx0 + 1;
        const int bs = ngx >= 256 ? 256 : (ngx >= 128 ? 128 : 64);
        dim3 grid((ngx + bs - 1) / bs, D.h + 2 * wh, count);
        hipLaunchKernelGGL(pyr_down_u8, grid, dim3(bs), 0, st, S, D, slot0, ww, wh, gx0, ngx);
    }
    if (pyr->with_deriv) {
        ScharrPlan plan;
        plan.levels = pyr->levels;
        int total = 0;
        for (int l = 0; l < pyr->levels; ++l) {
            plan.blk_start[l] = total;
            plan.blk_x[l] = (pyr->lv[l].w + 511) / 512;
            total += plan.blk_x[l] * ((pyr->lv[l].h + 3) / 4);
        }
        plan.blk_start[pyr->levels] = total;
        hipLaunchKernelGGL(scharr_deriv_s16x2, dim3(total, count), dim3(256), 0, st, pyr->d_lv, plan, slot0);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "pyramid launch");
    return VOFE_OK;
}
