/*
 * vofe_oracle.cpp -- CPU restatement of the feature front end.  TEST INFRASTRUCTURE ONLY.
 *
 * See vofe_oracle.h for the scope statement.  PARITY UNPINNED against cv2 (absent from the image);
 * the algorithm is the one written down in SURVEY.md Appendix A, which restates OpenCV 4.10
 * modules/imgproc/src/{featureselect,corner,deriv,box_filter,thresh,morph,pyramids}.cpp and
 * modules/video/src/lkpyramid.cpp as reached from the reference call sites
 * continuous.py:145-151, continuous.py:380-385, initialization.py:41-52.
 *
 * Canonical arithmetic (shared with the HIP kernels so that both sides agree bit for bit):
 *  - integer stages exact; window sums of integer products accumulated exactly (int64) and
 *    converted to f32 once;
 *  - float stages in the operation order stated at each function, built with -ffp-contract=off;
 *  - floor_i() maps non-finite / huge coordinates to INT_MIN so that they fail the bounds tests.
 */
#include "vofe_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- helpers
inline int reflect101(int p, int n) {            // SURVEY Appendix A conventions
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}
inline int floor_i(float v) {                     // cvFloor with a defined result for wild inputs
    if (!(v > -1073741824.f && v < 1073741824.f)) return INT_MIN;
    int i = (int)v;
    return i - ((float)i > v);
}
inline int round_i(float v) { return (int)lrintf(v); }   // cvRound: half to even

template <class F>
void parallel_for(int n, int nthreads, F&& body) {
    if (nthreads <= 1 || n <= 1) { body(0, n); return; }
    nthreads = std::min(nthreads, n);
    std::vector<std::thread> th;
    int chunk = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        int lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([=, &body] { body(lo, hi); });
    }
    for (auto& t : th) t.join();
}

// ---------------------------------------------------------------- planes
struct PlaneU8 {                                  // u8 level padded by (px,py), REFLECT_101 filled
    int w = 0, h = 0, px = 0, py = 0, pitch = 0;
    std::vector<uint8_t> buf;
    void alloc(int w_, int h_, int px_, int py_) {
        w = w_; h = h_; px = px_; py = py_; pitch = w + 2 * px;
        buf.assign((size_t)pitch * (h + 2 * py), 0);
    }
    uint8_t* row(int y) { return buf.data() + (size_t)(y + py) * pitch + px; }
    const uint8_t* row(int y) const { return buf.data() + (size_t)(y + py) * pitch + px; }
    void fill_border() {
        for (int y = -py; y < h + py; ++y) {
            int sy = reflect101(y, h);
            uint8_t* d = row(y);
            const uint8_t* s = row(sy);
            if (y < 0 || y >= h)
                for (int x = 0; x < w; ++x) d[x] = s[x];
            for (int x = -px; x < 0; ++x) d[x] = s[reflect101(x, w)];
            for (int x = w; x < w + px; ++x) d[x] = s[reflect101(x, w)];
        }
    }
};
struct PlaneS16x2 {                               // (Ix,Iy) interleaved, zero padded by (px,py)
    int w = 0, h = 0, px = 0, py = 0, pitch = 0;  // pitch in int16 units
    std::vector<int16_t> buf;
    void alloc(int w_, int h_, int px_, int py_) {
        w = w_; h = h_; px = px_; py = py_; pitch = 2 * (w + 2 * px);
        buf.assign((size_t)pitch * (h + 2 * py), 0);
    }
    int16_t* row(int y) { return buf.data() + (size_t)(y + py) * pitch + 2 * px; }
    const int16_t* row(int y) const { return buf.data() + (size_t)(y + py) * pitch + 2 * px; }
};

// SURVEY A.3 pyrDown 8U: 5x5 [1 4 6 4 1] separable, stride 2, (sum+128)>>8, REFLECT_101.
void pyr_down(const uint8_t* src, int h, int w, int spitch, uint8_t* dst, int dpitch) {
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    std::vector<int> rows((size_t)5 * dw);
    for (int Y = 0; Y < dh; ++Y) {
        for (int k = 0; k < 5; ++k) {
            const uint8_t* s = src + (size_t)reflect101(2 * Y - 2 + k, h) * spitch;
            int* r = rows.data() + (size_t)k * dw;
            for (int X = 0; X < dw; ++X) {
                int x0 = reflect101(2 * X - 2, w), x1 = reflect101(2 * X - 1, w), x2 = 2 * X,
                    x3 = reflect101(2 * X + 1, w), x4 = reflect101(2 * X + 2, w);
                r[X] = s[x0] + 4 * s[x1] + 6 * s[x2] + 4 * s[x3] + s[x4];
            }
        }
        const int *r0 = rows.data(), *r1 = r0 + dw, *r2 = r1 + dw, *r3 = r2 + dw, *r4 = r3 + dw;
        uint8_t* d = dst + (size_t)Y * dpitch;
        for (int X = 0; X < dw; ++X)
            d[X] = (uint8_t)((r0[X] + 4 * r1[X] + 6 * r2[X] + 4 * r3[X] + r4[X] + 128) >> 8);
    }
}

// SURVEY A.4 calcScharrDeriv: vertical pass t0 = 3(a+c)+10b, t1 = c-a; horizontal pass
// Ix = t0[x+1]-t0[x-1], Iy = 3(t1[x-1]+t1[x+1])+10 t1[x]; REFLECT_101 both ways.
void scharr(const uint8_t* src, int h, int w, int spitch, int16_t* dst, int dpitch) {
    std::vector<int> t0(w + 2), t1(w + 2);
    for (int y = 0; y < h; ++y) {
        const uint8_t* a = src + (size_t)reflect101(y - 1, h) * spitch;
        const uint8_t* b = src + (size_t)y * spitch;
        const uint8_t* c = src + (size_t)reflect101(y + 1, h) * spitch;
        for (int x = -1; x <= w; ++x) {
            int sx = reflect101(x, w);
            t0[x + 1] = 3 * (a[sx] + c[sx]) + 10 * b[sx];
            t1[x + 1] = c[sx] - a[sx];
        }
        int16_t* d = dst + (size_t)y * dpitch;
        for (int x = 0; x < w; ++x) {
            d[2 * x] = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)(3 * (t1[x] + t1[x + 2]) + 10 * t1[x + 1]);
        }
    }
}

// SURVEY A.3 buildOpticalFlowPyramid
struct Pyramid {
    std::vector<PlaneU8> I;
    std::vector<PlaneS16x2> D;
    int max_level = 0;
};
void build_pyramid(Pyramid& P, const uint8_t* img, int h, int w, int pitch, int win_w, int win_h,
                   int max_level, bool with_deriv) {
    int lw = w, lh = h;
    for (int level = 0; level <= max_level; ++level) {
        P.I.emplace_back();
        PlaneU8& L = P.I.back();
        L.alloc(lw, lh, win_w, win_h);
        if (level == 0) {
            for (int y = 0; y < lh; ++y) memcpy(L.row(y), img + (size_t)y * pitch, lw);
        } else {
            const PlaneU8& S = P.I[level - 1];
            pyr_down(S.row(0), S.h, S.w, S.pitch, L.row(0), L.pitch);
        }
        L.fill_border();
        if (with_deriv) {
            P.D.emplace_back();
            PlaneS16x2& Dl = P.D.back();
            Dl.alloc(lw, lh, win_w, win_h);
            scharr(L.row(0), lh, lw, L.pitch, Dl.row(0), Dl.pitch);
        }
        P.max_level = level;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= win_w || lh <= win_h) break;
    }
}

// ---------------------------------------------------------------- LK
struct LKArgs {
    int win_w, win_h, max_count, flags;
    double eps2;
    float min_eig_thr;
};
constexpr int W_BITS = 14;
constexpr float FLT_SCALE = 1.f / (1 << 20);
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

struct Weights { int w00, w01, w10, w11; };
inline Weights bilinear_weights(float fx, float fy, int ix, int iy) {
    float a = fx - (float)ix, b = fy - (float)iy;
    Weights q;
    q.w00 = round_i((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    q.w01 = round_i(a * (1.f - b) * (float)(1 << W_BITS));
    q.w10 = round_i((1.f - a) * b * (float)(1 << W_BITS));
    q.w11 = (1 << W_BITS) - q.w00 - q.w01 - q.w10;
    return q;
}

void lk_point(const Pyramid& prev, const Pyramid& next, const LKArgs& A, float px, float py,
              float& nx, float& ny, uint8_t& status, float& err, std::vector<int16_t>& buf) {
    const int w = A.win_w, h = A.win_h;
    const float hx = (w - 1) * 0.5f, hy = (h - 1) * 0.5f;
    const int maxLevel = prev.max_level;
    buf.resize((size_t)3 * w * h);
    int16_t* Ip = buf.data();
    int16_t* dIp = Ip + (size_t)w * h;                // interleaved (ix,iy)
    float outx = nx, outy = ny;                        // nextPts[i]
    for (int level = maxLevel; level >= 0; --level) {
        const PlaneU8& I = prev.I[level];
        const PlaneS16x2& D = prev.D[level];
        const PlaneU8& J = next.I[level];
        const float sc = (float)(1. / (1 << level));
        float ppx = px * sc, ppy = py * sc;
        float npx, npy;
        if (level == maxLevel) {
            if (A.flags & 4) { npx = outx * sc; npy = outy * sc; }
            else { npx = ppx; npy = ppy; }
        } else { npx = outx * 2.f; npy = outy * 2.f; }
        outx = npx; outy = npy;

        ppx -= hx; ppy -= hy;
        int ipx = floor_i(ppx), ipy = floor_i(ppy);
        if (ipx < -w || ipx >= I.w || ipy < -h || ipy >= I.h) {
            if (level == 0) { status = 0; err = 0.f; }
            continue;
        }
        Weights q = bilinear_weights(ppx, ppy, ipx, ipy);
        int64_t sA11 = 0, sA12 = 0, sA22 = 0;
        for (int y = 0; y < h; ++y) {
            const uint8_t* s0 = I.row(ipy + y) + ipx;
            const uint8_t* s1 = I.row(ipy + y + 1) + ipx;
            const int16_t* d0 = D.row(ipy + y) + 2 * ipx;
            const int16_t* d1 = D.row(ipy + y + 1) + 2 * ipx;
            for (int x = 0; x < w; ++x) {
                int iv = descale(s0[x] * q.w00 + s0[x + 1] * q.w01 + s1[x] * q.w10 + s1[x + 1] * q.w11, W_BITS - 5);
                int ix = descale(d0[2 * x] * q.w00 + d0[2 * x + 2] * q.w01 + d1[2 * x] * q.w10 + d1[2 * x + 2] * q.w11, W_BITS);
                int iy = descale(d0[2 * x + 1] * q.w00 + d0[2 * x + 3] * q.w01 + d1[2 * x + 1] * q.w10 + d1[2 * x + 3] * q.w11, W_BITS);
                Ip[y * w + x] = (int16_t)iv;
                dIp[2 * (y * w + x)] = (int16_t)ix;
                dIp[2 * (y * w + x) + 1] = (int16_t)iy;
                sA11 += (int64_t)ix * ix; sA12 += (int64_t)ix * iy; sA22 += (int64_t)iy * iy;
            }
        }
        float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float Dd = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * w * h);
        if (A.flags & 8) err = minEig;
        if (minEig < A.min_eig_thr || Dd < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        Dd = 1.f / Dd;
        npx -= hx; npy -= hy;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < A.max_count; ++j) {
            int inx = floor_i(npx), iny = floor_i(npy);
            if (inx < -w || inx >= J.w || iny < -h || iny >= J.h) {
                if (level == 0) status = 0;
                break;
            }
            Weights r = bilinear_weights(npx, npy, inx, iny);
            int64_t sb1 = 0, sb2 = 0;
            for (int y = 0; y < h; ++y) {
                const uint8_t* j0 = J.row(iny + y) + inx;
                const uint8_t* j1 = J.row(iny + y + 1) + inx;
                for (int x = 0; x < w; ++x) {
                    int diff = descale(j0[x] * r.w00 + j0[x + 1] * r.w01 + j1[x] * r.w10 + j1[x + 1] * r.w11, W_BITS - 5) - Ip[y * w + x];
                    sb1 += (int64_t)diff * dIp[2 * (y * w + x)];
                    sb2 += (int64_t)diff * dIp[2 * (y * w + x) + 1];
                }
            }
            float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            float dx = (A12 * b2 - A22 * b1) * Dd;
            float dy = (A12 * b1 - A11 * b2) * Dd;
            npx += dx; npy += dy;
            outx = npx + hx; outy = npy + hy;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= A.eps2) break;
            if (j > 0 && std::abs((double)(dx + pdx)) < 0.01 && std::abs((double)(dy + pdy)) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0 && !(A.flags & 8)) {
            float ex = outx - hx, ey = outy - hy;
            int inx = floor_i(ex), iny = floor_i(ey);
            if (inx < -w || inx >= J.w || iny < -h || iny >= J.h) { status = 0; continue; }
            Weights r = bilinear_weights(ex, ey, inx, iny);
            int64_t se = 0;
            for (int y = 0; y < h; ++y) {
                const uint8_t* j0 = J.row(iny + y) + inx;
                const uint8_t* j1 = J.row(iny + y + 1) + inx;
                for (int x = 0; x < w; ++x) {
                    int diff = descale(j0[x] * r.w00 + j0[x + 1] * r.w01 + j1[x] * r.w10 + j1[x + 1] * r.w11, W_BITS - 5) - Ip[y * w + x];
                    se += std::abs(diff);
                }
            }
            err = (float)se / (float)(32 * w * h);
        }
    }
    nx = outx; ny = outy;
}

int lk_check(int h, int w, int n, int win_w, int win_h, int max_level) {
    if (h <= 0 || w <= 0 || n < 0) return -1;
    if (max_level < 0 || win_w <= 2 || win_h <= 2) return -2;
    return 0;
}
void lk_criteria(int type, int& max_count, double& eps) {
    max_count = (type & 1) ? std::min(std::max(max_count, 0), 100) : 30;
    eps = (type & 2) ? std::min(std::max(eps, 0.), 10.) : 0.01;
    eps *= eps;
}

// ---------------------------------------------------------------- corner score
void corner_score(const uint8_t* img, int h, int w, int pitch, int harris, double k, float* score, int nthreads) {
    const float s1 = (float)(1.0 / 3060.0), s2 = (float)(2.0 / 3060.0);
    std::vector<float> dx((size_t)h * w), dy((size_t)h * w);
    parallel_for(h, nthreads, [&](int y0, int y1) {
        std::vector<float> ra(w), rb(w), rc(w);       // row-pass outputs of rows y-1,y,y+1
        for (int y = y0; y < y1; ++y) {
            const uint8_t* a = img + (size_t)reflect101(y - 1, h) * pitch;
            const uint8_t* b = img + (size_t)y * pitch;
            const uint8_t* c = img + (size_t)reflect101(y + 1, h) * pitch;
            for (int x = 0; x < w; ++x) {
                int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
                // Dx: row kernel [-1 0 1] (exact), column kernel [s 2s s]
                float r0 = (float)a[xp] - (float)a[xm], r1 = (float)b[xp] - (float)b[xm], r2 = (float)c[xp] - (float)c[xm];
                dx[(size_t)y * w + x] = s2 * r1 + s1 * (r0 + r2);
                // Dy: row kernel [s 2s s] on u8, column kernel [-1 0 1]
                float q0 = s2 * (float)a[x] + s1 * (float)(a[xm] + a[xp]);
                float q2 = s2 * (float)c[x] + s1 * (float)(c[xm] + c[xp]);
                dy[(size_t)y * w + x] = q2 - q0;
            }
        }
    });
    parallel_for(h, nthreads, [&](int y0, int y1) {
        for (int y = y0; y < y1; ++y) {
            for (int x = 0; x < w; ++x) {
                double rs0[3], rs1[3], rs2[3];
                for (int j = 0; j < 3; ++j) {
                    int sy = reflect101(y - 1 + j, h);
                    double a0 = 0, a1 = 0, a2 = 0;
                    for (int i = 0; i < 3; ++i) {
                        int sx = reflect101(x - 1 + i, w);
                        float gx = dx[(size_t)sy * w + sx], gy = dy[(size_t)sy * w + sx];
                        float c0 = gx * gx, c1 = gx * gy, c2 = gy * gy;
                        if (i == 0) { a0 = c0; a1 = c1; a2 = c2; }
                        else { a0 += (double)c0; a1 += (double)c1; a2 += (double)c2; }
                    }
                    rs0[j] = a0; rs1[j] = a1; rs2[j] = a2;
                }
                float c0 = (float)((rs0[0] + rs0[1]) + rs0[2]);
                float c1 = (float)((rs1[0] + rs1[1]) + rs1[2]);
                float c2 = (float)((rs2[0] + rs2[1]) + rs2[2]);
                float out;
                if (!harris) {
                    float a = c0 * 0.5f, b = c1, c = c2 * 0.5f;
                    out = (a + c) - std::sqrt((a - c) * (a - c) + b * b);
                } else {
                    float kf = (float)k;
                    out = (c0 * c2 - c1 * c1) - kf * ((c0 + c2) * (c0 + c2));
                }
                score[(size_t)y * w + x] = out;
            }
        }
    });
}

}  // namespace

// ================================================================= C ABI
extern "C" {

int vofe_cpu_pyr_down_u8(const uint8_t* src, int h, int w, int spitch, uint8_t* dst, int dpitch) {
    if (!src || !dst || h <= 0 || w <= 0) return -1;
    pyr_down(src, h, w, spitch, dst, dpitch);
    return 0;
}
int vofe_cpu_scharr_s16x2(const uint8_t* src, int h, int w, int spitch, int16_t* dst, int dpitch) {
    if (!src || !dst || h <= 0 || w <= 0) return -1;
    scharr(src, h, w, spitch, dst, dpitch);
    return 0;
}
int vofe_cpu_pyramid_levels(int h, int w, int win_w, int win_h, int max_level, int* level_w, int* level_h) {
    if (h <= 0 || w <= 0 || max_level < 0) return -1;
    int n = 0;
    for (int level = 0; level <= max_level; ++level) {
        if (level_w) level_w[n] = w;
        if (level_h) level_h[n] = h;
        ++n;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win_w || h <= win_h) break;
    }
    return n;
}
int vofe_cpu_pyramid_level(const uint8_t* img, int h, int w, int pitch, int level, uint8_t* dst) {
    if (!img || !dst || h <= 0 || w <= 0 || level < 0) return -1;
    std::vector<uint8_t> cur((size_t)h * w), nxt;
    for (int y = 0; y < h; ++y) memcpy(cur.data() + (size_t)y * w, img + (size_t)y * pitch, w);
    for (int l = 0; l < level; ++l) {
        int dw = (w + 1) / 2, dh = (h + 1) / 2;
        nxt.assign((size_t)dw * dh, 0);
        pyr_down(cur.data(), h, w, w, nxt.data(), dw);
        cur.swap(nxt); w = dw; h = dh;
    }
    memcpy(dst, cur.data(), (size_t)w * h);
    return 0;
}

int vofe_cpu_lk_track(const uint8_t* prev_img, const uint8_t* next_img, int h, int w, int pitch,
                      const float* prev_pts, float* next_pts, uint8_t* status, float* err, int n,
                      int win_w, int win_h, int max_level, int crit_type, int max_count, double eps,
                      int flags, double min_eig_thr, int nthreads) {
    int rc = lk_check(h, w, n, win_w, win_h, max_level);
    if (rc) return rc;
    if (n == 0) return 0;
    if (!prev_img || !next_img || !prev_pts || !next_pts || !status || !err) return -1;
    LKArgs A;
    A.win_w = win_w; A.win_h = win_h; A.flags = flags; A.min_eig_thr = (float)min_eig_thr;
    lk_criteria(crit_type, max_count, eps);
    A.max_count = max_count; A.eps2 = eps;
    Pyramid P, N;
    build_pyramid(P, prev_img, h, w, pitch, win_w, win_h, max_level, true);
    build_pyramid(N, next_img, h, w, pitch, win_w, win_h, max_level, false);
    parallel_for(n, nthreads, [&](int lo, int hi) {
        std::vector<int16_t> buf;
        for (int i = lo; i < hi; ++i) {
            status[i] = 1;
            err[i] = 0.f;
            float nx = next_pts[2 * i], ny = next_pts[2 * i + 1];
            lk_point(P, N, A, prev_pts[2 * i], prev_pts[2 * i + 1], nx, ny, status[i], err[i], buf);
            next_pts[2 * i] = nx; next_pts[2 * i + 1] = ny;
        }
    });
    return 0;
}

int vofe_cpu_track_fb(const uint8_t* prev_img, const uint8_t* next_img, int h, int w, int pitch,
                      const float* prev_pts, float* next_pts, uint8_t* status, float* err, int n,
                      int win_w, int win_h, int max_level, int crit_type, int max_count, double eps,
                      double min_eig_thr, float fb_thresh, int nthreads) {
    int rc = vofe_cpu_lk_track(prev_img, next_img, h, w, pitch, prev_pts, next_pts, status, err, n,
                               win_w, win_h, max_level, crit_type, max_count, eps, 0, min_eig_thr, nthreads);
    if (rc || n == 0) return rc;
    std::vector<float> back((size_t)2 * n), berr(n);
    std::vector<uint8_t> bst(n);
    rc = vofe_cpu_lk_track(next_img, prev_img, h, w, pitch, next_pts, back.data(), bst.data(), berr.data(), n,
                           win_w, win_h, max_level, crit_type, max_count, eps, 0, min_eig_thr, nthreads);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        float dx = std::fabs(prev_pts[2 * i] - back[2 * i]);
        float dy = std::fabs(prev_pts[2 * i + 1] - back[2 * i + 1]);
        float d = dx > dy ? dx : dy;
        if (dx != dx || dy != dy) d = NAN;           // numpy max propagates NaN; NaN < thr is False
        status[i] = (uint8_t)(status[i] & (d < fb_thresh ? 1 : 0));
    }
    return 0;
}

int vofe_cpu_corner_score(const uint8_t* img, int h, int w, int pitch, int harris, double k,
                          float* score, int nthreads) {
    if (!img || !score || h <= 0 || w <= 0) return -1;
    corner_score(img, h, w, pitch, harris, k, score, nthreads);
    return 0;
}

int vofe_cpu_gftt(const uint8_t* img, int h, int w, int pitch, int max_corners, double quality,
                  double min_distance, const uint8_t* mask, int harris, double k,
                  float* corners_xy, int capacity, int* n_out, int nthreads) {
    if (!n_out) return -1;
    *n_out = 0;
    if (!(quality > 0) || !(min_distance >= 0) || max_corners < 0) return -2;     // SURVEY A.2 asserts
    if (!img || h <= 0 || w <= 0) return 0;                                       // empty image -> None
    std::vector<float> eig((size_t)h * w);
    corner_score(img, h, w, pitch, harris, k, eig.data(), nthreads);
    // 1. global max (within mask)
    bool any = false;
    float maxv = 0.f;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            if (mask && !mask[(size_t)y * w + x]) continue;
            float v = eig[(size_t)y * w + x];
            if (!any || v > maxv) { maxv = v; any = true; }
        }
    if (!any) return 0;
    // 2. threshold to zero
    const float thr = (float)((double)maxv * quality);
    for (auto& v : eig) v = v > thr ? v : 0.f;
    // 3./4. 3x3 dilate + collect interior local maxima
    struct Cand { float v; int idx; };
    std::vector<Cand> cand;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            float v = eig[(size_t)y * w + x];
            if (v == 0.f) continue;
            if (mask && !mask[(size_t)y * w + x]) continue;
            float m = v;
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) m = std::max(m, eig[(size_t)(y + j) * w + x + i]);
            if (v == m) cand.push_back({v, y * w + x});
        }
    if (cand.empty()) return 0;
    // 5. descending by score, ties: larger address first
    std::sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) {
        return a.v > b.v || (a.v == b.v && a.idx > b.idx);
    });
    // 6. grid greedy
    int ncorners = 0;
    auto emit = [&](int x, int y) {
        if (ncorners < capacity && corners_xy) { corners_xy[2 * ncorners] = (float)x; corners_xy[2 * ncorners + 1] = (float)y; }
        ++ncorners;
    };
    if (min_distance >= 1) {
        const int cell = (int)lrint(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        std::vector<std::vector<std::pair<float, float>>> grid((size_t)gw * gh);
        const double md2 = min_distance * min_distance;
        for (const Cand& c : cand) {
            int y = c.idx / w, x = c.idx - y * w;
            int xc = x / cell, yc = y / cell;
            int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
            int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            bool good = true;
            for (int yy = y1; yy <= y2 && good; ++yy)
                for (int xx = x1; xx <= x2 && good; ++xx)
                    for (auto& p : grid[(size_t)yy * gw + xx]) {
                        float ddx = (float)x - p.first, ddy = (float)y - p.second;
                        if ((double)(ddx * ddx + ddy * ddy) < md2) { good = false; break; }
                    }
            if (good) {
                grid[(size_t)yc * gw + xc].push_back({(float)x, (float)y});
                emit(x, y);
                if (max_corners > 0 && ncorners == max_corners) break;
            }
        }
    } else {
        for (const Cand& c : cand) {
            int y = c.idx / w, x = c.idx - y * w;
            emit(x, y);
            if (max_corners > 0 && ncorners == max_corners) break;
        }
    }
    *n_out = ncorners;
    return 0;
}

}  // extern "C"
