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
 *
 * VARIANTS (vofe_cpu_set_variant): every point SURVEY Appendix A marks as depending on the cv2
 * binary's SIMD dispatch / IPP use has switchable alternatives here, so that tests can MEASURE how
 * far the plausible cv2 arithmetics lie from the canonical one (tests/test_variants_cpu.py,
 * DESIGN.md section 3).  Variant 0 is the canonical arithmetic and the only one the kernels follow.
 */
#include "vofe_oracle.h"

#include <immintrin.h>

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

// ---------------------------------------------------------------- variants (SURVEY A.1, A.2, A.5 "⚠" points)
struct Variant {
    int lk_sum = 0;       // A.5 steps 4, 6.  0 exact int64 (canonical) | 1 f32 raster (scalar C++ build)
                          // 2/3 four f32 lanes, 4 px per step, without/with FMA (the NEON loop's shape)
                          // 4/5 universal-intrinsics 128-bit loop: 8 px per step into 4 lanes, A by muladd
                          //     without/with FMA, b by int32 pair dot products converted to f32 (x86 wheel)
                          // 6/7 eight f32 lanes, 8 px per step, without/with FMA (a 256-bit build)
    int sobel = 0;        // A.1 step 2.  0 canonical | 1 FMA in the smoothing pass | 2 exact integer Sobel, scaled after (IPP-like)
                          //   | 3 the smoothing ROW pass accumulated tap by tap, left to right (OpenCV's generic RowFilter
                          //   for 8u -> 32f) instead of the symmetric form
    int mineig_fma = 0;   // A.1 step 5.  sqrt argument as fma(b, b, (a-c)*(a-c))
    int box_running = 0;  // A.1 step 4.  f64 box sums as OpenCV's running row / column sums instead of direct 3+3 sums
    int harris = 0;       // A.1 step 5.  0 canonical f32 k*((a+c)^2) | 1 scalar path with k double | 2 f32 (k*(a+c))*(a+c)
    int thresh_ipp = 0;   // A.2 step 2.  keep v unless v < (float)(thr + FLT_EPSILON)
    bool any_lk() const { return lk_sum != 0; }
};
Variant g_variant;        // set between calls only (vofe_cpu_set_variant); worker threads read it

// One f32 running sum of integer products in one of the orders a cv2 build may use.
struct F32Acc {
    int lanes = 1, step = 1;
    bool fma = false;
    float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float tail = 0.f;
    void init(int mode) {
        switch (mode) {
            case 2: case 3: lanes = 4; step = 4; break;
            case 4: case 5: lanes = 4; step = 8; break;
            case 6: case 7: lanes = 8; step = 8; break;
            default: lanes = 1; step = 1; break;
        }
        fma = (mode == 3 || mode == 5 || mode == 7);
    }
    void add_row(const int* a, const int* b, int w) {
        int x = 0;
        if (lanes > 1)
            for (; x <= w - step; x += step)
                for (int s = 0; s < step; ++s) {
                    float fa = (float)a[x + s], fb = (float)b[x + s];
                    float& l = lane[s % lanes];
                    l = fma ? fmaf(fa, fb, l) : l + fa * fb;
                }
        for (; x < w; ++x) tail += (float)(a[x] * b[x]);          // scalar loop: (itemtype)(int product), raster order
    }
    float total() const {
        if (lanes == 1) return tail;
        if (step == 4) return tail + (((lane[0] + lane[1]) + lane[2]) + lane[3]);   // stored to an array, summed left to right
        float r = (lane[0] + lane[1]) + (lane[2] + lane[3]);                        // v_reduce_sum: pairwise
        if (lanes == 8) r = r + ((lane[4] + lane[5]) + (lane[6] + lane[7]));
        return tail + r;
    }
};
// b1, b2 of the 128-bit universal-intrinsics loop: int32 dot products of pixel pairs (x, x+4), converted, f32 lanes.
struct PairDotB {
    float q0[4] = {0, 0, 0, 0}, q1[4] = {0, 0, 0, 0}, t1 = 0.f, t2 = 0.f;
    void add_row(const int* d, const int* ix, const int* iy, int w) {
        int x = 0;
        for (; x <= w - 8; x += 8) {
            q0[0] += (float)(d[x] * ix[x] + d[x + 4] * ix[x + 4]);
            q0[1] += (float)(d[x] * iy[x] + d[x + 4] * iy[x + 4]);
            q0[2] += (float)(d[x + 1] * ix[x + 1] + d[x + 5] * ix[x + 5]);
            q0[3] += (float)(d[x + 1] * iy[x + 1] + d[x + 5] * iy[x + 5]);
            q1[0] += (float)(d[x + 2] * ix[x + 2] + d[x + 6] * ix[x + 6]);
            q1[1] += (float)(d[x + 2] * iy[x + 2] + d[x + 6] * iy[x + 6]);
            q1[2] += (float)(d[x + 3] * ix[x + 3] + d[x + 7] * ix[x + 7]);
            q1[3] += (float)(d[x + 3] * iy[x + 3] + d[x + 7] * iy[x + 7]);
        }
        for (; x < w; ++x) { t1 += (float)(d[x] * ix[x]); t2 += (float)(d[x] * iy[x]); }
    }
    void totals(float& b1, float& b2) const {
        float s0 = q0[0] + q1[0], s1 = q0[1] + q1[1], s2 = q0[2] + q1[2], s3 = q0[3] + q1[3];
        b1 = t1 + ((s0 + s2) + (0.f + 0.f));
        b2 = t2 + ((s1 + s3) + (0.f + 0.f));
    }
};

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
        buf.assign((size_t)pitch * (h + 2 * py) + 64, 0);      // (+ slack: the vector loads of lk_point_simd read past a row's end)
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
        buf.assign((size_t)pitch * (h + 2 * py) + 64, 0);
    }
    int16_t* row(int y) { return buf.data() + (size_t)(y + py) * pitch + 2 * px; }
    const int16_t* row(int y) const { return buf.data() + (size_t)(y + py) * pitch + 2 * px; }
};

// SURVEY A.3 pyrDown 8U: 5x5 [1 4 6 4 1] separable, stride 2, (sum+128)>>8, REFLECT_101.  Output rows [Y0, Y1).
void pyr_down_rows(const uint8_t* src, int h, int w, int spitch, uint8_t* dst, int dpitch, int Y0, int Y1) {
    int dw = (w + 1) / 2;
    std::vector<int> rows((size_t)5 * dw);
    const int Xa = std::min(dw, 1), Xb = std::max(Xa, (w - 3) / 2 + 1);      // X in [Xa, Xb): 2X-2 >= 0 and 2X+2 <= w-1, no reflection
    for (int Y = Y0; Y < Y1; ++Y) {
        for (int k = 0; k < 5; ++k) {
            const uint8_t* s = src + (size_t)reflect101(2 * Y - 2 + k, h) * spitch;
            int* r = rows.data() + (size_t)k * dw;
            auto edge = [&](int X) {
                int x0 = reflect101(2 * X - 2, w), x1 = reflect101(2 * X - 1, w), x2 = 2 * X,
                    x3 = reflect101(2 * X + 1, w), x4 = reflect101(2 * X + 2, w);
                r[X] = s[x0] + 4 * s[x1] + 6 * s[x2] + 4 * s[x3] + s[x4];
            };
            for (int X = 0; X < Xa; ++X) edge(X);
            for (int X = Xa; X < Xb; ++X) {
                const uint8_t* q = s + 2 * X;
                r[X] = q[-2] + 4 * q[-1] + 6 * q[0] + 4 * q[1] + q[2];
            }
            for (int X = Xb; X < dw; ++X) edge(X);
        }
        const int *r0 = rows.data(), *r1 = r0 + dw, *r2 = r1 + dw, *r3 = r2 + dw, *r4 = r3 + dw;
        uint8_t* d = dst + (size_t)Y * dpitch;
        for (int X = 0; X < dw; ++X)
            d[X] = (uint8_t)((r0[X] + 4 * r1[X] + 6 * r2[X] + 4 * r3[X] + r4[X] + 128) >> 8);
    }
}
void pyr_down(const uint8_t* src, int h, int w, int spitch, uint8_t* dst, int dpitch, int nthreads = 1) {
    parallel_for((h + 1) / 2, nthreads, [&](int Y0, int Y1) { pyr_down_rows(src, h, w, spitch, dst, dpitch, Y0, Y1); });
}

// SURVEY A.4 calcScharrDeriv: vertical pass t0 = 3(a+c)+10b, t1 = c-a; horizontal pass
// Ix = t0[x+1]-t0[x-1], Iy = 3(t1[x-1]+t1[x+1])+10 t1[x]; REFLECT_101 both ways.  Rows [y0, y1).
void scharr_rows(const uint8_t* src, int h, int w, int spitch, int16_t* dst, int dpitch, int y0, int y1) {
    std::vector<int> t0(w + 2), t1(w + 2);
    for (int y = y0; y < y1; ++y) {
        const uint8_t* a = src + (size_t)reflect101(y - 1, h) * spitch;
        const uint8_t* b = src + (size_t)y * spitch;
        const uint8_t* c = src + (size_t)reflect101(y + 1, h) * spitch;
        for (int x = 0; x < w; ++x) {
            t0[x + 1] = 3 * (a[x] + c[x]) + 10 * b[x];
            t1[x + 1] = c[x] - a[x];
        }
        for (int x : {-1, w}) {
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
void scharr(const uint8_t* src, int h, int w, int spitch, int16_t* dst, int dpitch, int nthreads = 1) {
    parallel_for(h, nthreads, [&](int y0, int y1) { scharr_rows(src, h, w, spitch, dst, dpitch, y0, y1); });
}

// SURVEY A.3 buildOpticalFlowPyramid
struct Pyramid {
    std::vector<PlaneU8> I;
    std::vector<PlaneS16x2> D;
    int max_level = 0;
};
void build_pyramid(Pyramid& P, const uint8_t* img, int h, int w, int pitch, int win_w, int win_h,
                   int max_level, bool with_deriv, int nthreads = 1) {
    int lw = w, lh = h;
    for (int level = 0; level <= max_level; ++level) {
        P.I.emplace_back();
        PlaneU8& L = P.I.back();
        L.alloc(lw, lh, win_w, win_h);
        if (level == 0) {
            for (int y = 0; y < lh; ++y) memcpy(L.row(y), img + (size_t)y * pitch, lw);
        } else {
            const PlaneU8& S = P.I[level - 1];
            pyr_down(S.row(0), S.h, S.w, S.pitch, L.row(0), L.pitch, (size_t)lw * lh >= 32768 ? nthreads : 1);
        }
        L.fill_border();
        if (with_deriv) {
            P.D.emplace_back();
            PlaneS16x2& Dl = P.D.back();
            Dl.alloc(lw, lh, win_w, win_h);
            scharr(L.row(0), lh, lw, L.pitch, Dl.row(0), Dl.pitch, (size_t)lw * lh >= 32768 ? nthreads : 1);
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

template <bool VAR>
void lk_point(const Pyramid& prev, const Pyramid& next, const LKArgs& A, float px, float py,
              float& nx, float& ny, uint8_t& status, float& err, std::vector<int16_t>& buf) {
    const int w = A.win_w, h = A.win_h;
    const int vmode = VAR ? g_variant.lk_sum : 0;
    std::vector<int> ra, rb, rc;                       // one window row of integer factors (variants only)
    if (VAR) { ra.resize(w); rb.resize(w); rc.resize(w); }
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
        F32Acc fA11, fA12, fA22;
        if (VAR) { fA11.init(vmode); fA12.init(vmode); fA22.init(vmode); }
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
                if (VAR) { ra[x] = ix; rb[x] = iy; }
            }
            if (VAR) { fA11.add_row(ra.data(), ra.data(), w); fA12.add_row(ra.data(), rb.data(), w); fA22.add_row(rb.data(), rb.data(), w); }
        }
        float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        if (VAR) { A11 = fA11.total() * FLT_SCALE; A12 = fA12.total() * FLT_SCALE; A22 = fA22.total() * FLT_SCALE; }
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
            F32Acc fb1, fb2;
            PairDotB pb;
            if (VAR) { fb1.init(vmode); fb2.init(vmode); }
            for (int y = 0; y < h; ++y) {
                const uint8_t* j0 = J.row(iny + y) + inx;
                const uint8_t* j1 = J.row(iny + y + 1) + inx;
                for (int x = 0; x < w; ++x) {
                    int diff = descale(j0[x] * r.w00 + j0[x + 1] * r.w01 + j1[x] * r.w10 + j1[x + 1] * r.w11, W_BITS - 5) - Ip[y * w + x];
                    sb1 += (int64_t)diff * dIp[2 * (y * w + x)];
                    sb2 += (int64_t)diff * dIp[2 * (y * w + x) + 1];
                    if (VAR) { ra[x] = diff; rb[x] = dIp[2 * (y * w + x)]; rc[x] = dIp[2 * (y * w + x) + 1]; }
                }
                if (VAR) {
                    if (vmode == 4 || vmode == 5) pb.add_row(ra.data(), rb.data(), rc.data(), w);
                    else { fb1.add_row(ra.data(), rb.data(), w); fb2.add_row(ra.data(), rc.data(), w); }
                }
            }
            float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            if (VAR) {
                if (vmode == 4 || vmode == 5) pb.totals(b1, b2); else { b1 = fb1.total(); b2 = fb2.total(); }
                b1 *= FLT_SCALE; b2 *= FLT_SCALE;
            }
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


// ---------------------------------------------------------------- LK, AVX2 (the timed CPU baseline)
// lk_point<false> eight window pixels at a time.  The same canonical arithmetic -- integer bilinear samples, exact integer
// window sums converted to float once -- so the results are bit-identical to the scalar restatement (tests compare them); it
// exists because the CPU baseline that bench.py reports beside the GPU number should not be a scalar loop when the reference's
// cv2 runs SIMD code.  Layout trick as in OpenCV's own loop: pixel pairs (p[x], p[x+1]) as 16-bit halves of 32-bit lanes, one
// _mm256_madd_epi16 against (w00, w01) / (w10, w11) per source row.
bool g_simd = false;

struct Acc64 {                                     // exact sum of 8 int32 lanes per call, kept in 4 int64 lanes
    __m256i s = _mm256_setzero_si256();
    inline void add(__m256i v32) {
        s = _mm256_add_epi64(s, _mm256_cvtepi32_epi64(_mm256_castsi256_si128(v32)));
        s = _mm256_add_epi64(s, _mm256_cvtepi32_epi64(_mm256_extracti128_si256(v32, 1)));
    }
    inline int64_t total() const {
        alignas(32) int64_t t[4];
        _mm256_store_si256((__m256i*)t, s);
        return t[0] + t[1] + t[2] + t[3];
    }
};
inline __m256i pairs_u8(const uint8_t* p) {        // lanes i = 0..7: p[i] | p[i+1] << 16
    const __m128i a = _mm_loadl_epi64((const __m128i*)p), b = _mm_loadl_epi64((const __m128i*)(p + 1));
    const __m128i lo = _mm_cvtepu8_epi16(a), hi = _mm_cvtepu8_epi16(b);
    return _mm256_set_m128i(_mm_unpackhi_epi16(lo, hi), _mm_unpacklo_epi16(lo, hi));
}
inline __m256i lane_mask(int n) {                  // lanes < n all ones
    alignas(32) static const int32_t idx[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    return _mm256_cmpgt_epi32(_mm256_set1_epi32(n), _mm256_load_si256((const __m256i*)idx));
}

void lk_point_simd(const Pyramid& prev, const Pyramid& next, const LKArgs& A, float px, float py,
                   float& nx, float& ny, uint8_t& status, float& err, std::vector<int16_t>& buf) {
    const int w = A.win_w, h = A.win_h;
    const int wp = (w + 7) & ~7;                       // window row stride in the buffers (padded to whole vectors)
    const float hx = (w - 1) * 0.5f, hy = (h - 1) * 0.5f;
    const int maxLevel = prev.max_level;
    buf.assign((size_t)3 * wp * h + 32, 0);
    int16_t* Ip = buf.data();
    int16_t* dIp = Ip + (size_t)wp * h;               // interleaved (ix, iy), 2 * wp per row
    float outx = nx, outy = ny;
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
        const Weights q = bilinear_weights(ppx, ppy, ipx, ipy);
        const __m256i w0 = _mm256_set1_epi32((q.w00 & 0xffff) | (q.w01 << 16)), w1 = _mm256_set1_epi32((q.w10 & 0xffff) | (q.w11 << 16));
        const __m256i r9 = _mm256_set1_epi32(1 << 8), r14 = _mm256_set1_epi32(1 << 13);
        Acc64 a11, a12, a22;
        for (int y = 0; y < h; ++y) {
            const uint8_t* s0 = I.row(ipy + y) + ipx;
            const uint8_t* s1 = I.row(ipy + y + 1) + ipx;
            const int16_t* d0 = D.row(ipy + y) + 2 * ipx;
            const int16_t* d1 = D.row(ipy + y + 1) + 2 * ipx;
            for (int x = 0; x < w; x += 8) {
                const __m256i m = lane_mask(w - x);
                __m256i iv = _mm256_add_epi32(_mm256_add_epi32(_mm256_madd_epi16(pairs_u8(s0 + x), w0), _mm256_madd_epi16(pairs_u8(s1 + x), w1)), r9);
                iv = _mm256_srai_epi32(iv, 9);
                // derivative pairs: lane i of A_ = (Ix_i, Iy_i), of B_ = (Ix_{i+1}, Iy_{i+1})
                const __m256i A0 = _mm256_loadu_si256((const __m256i*)(d0 + 2 * x)), B0 = _mm256_loadu_si256((const __m256i*)(d0 + 2 * x + 2));
                const __m256i A1 = _mm256_loadu_si256((const __m256i*)(d1 + 2 * x)), B1 = _mm256_loadu_si256((const __m256i*)(d1 + 2 * x + 2));
                const __m256i X0 = _mm256_blend_epi16(A0, _mm256_slli_epi32(B0, 16), 0xAA), Y0 = _mm256_blend_epi16(_mm256_srli_epi32(A0, 16), B0, 0xAA);
                const __m256i X1 = _mm256_blend_epi16(A1, _mm256_slli_epi32(B1, 16), 0xAA), Y1 = _mm256_blend_epi16(_mm256_srli_epi32(A1, 16), B1, 0xAA);
                __m256i ix = _mm256_srai_epi32(_mm256_add_epi32(_mm256_add_epi32(_mm256_madd_epi16(X0, w0), _mm256_madd_epi16(X1, w1)), r14), 14);
                __m256i iy = _mm256_srai_epi32(_mm256_add_epi32(_mm256_add_epi32(_mm256_madd_epi16(Y0, w0), _mm256_madd_epi16(Y1, w1)), r14), 14);
                iv = _mm256_and_si256(iv, m); ix = _mm256_and_si256(ix, m); iy = _mm256_and_si256(iy, m);
                // store: Ip as int16, dIp as (ix, iy) int16 pairs
                const __m128i iv16 = _mm_packs_epi32(_mm256_castsi256_si128(iv), _mm256_extracti128_si256(iv, 1));
                _mm_storeu_si128((__m128i*)(Ip + (size_t)y * wp + x), iv16);
                const __m256i dd = _mm256_or_si256(_mm256_and_si256(ix, _mm256_set1_epi32(0xffff)), _mm256_slli_epi32(iy, 16));
                _mm256_storeu_si256((__m256i*)(dIp + 2 * ((size_t)y * wp + x)), dd);
                a11.add(_mm256_mullo_epi32(ix, ix)); a12.add(_mm256_mullo_epi32(ix, iy)); a22.add(_mm256_mullo_epi32(iy, iy));
            }
        }
        float A11 = (float)a11.total() * FLT_SCALE, A12 = (float)a12.total() * FLT_SCALE, A22 = (float)a22.total() * FLT_SCALE;
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
            const Weights r = bilinear_weights(npx, npy, inx, iny);
            const __m256i v0 = _mm256_set1_epi32((r.w00 & 0xffff) | (r.w01 << 16)), v1 = _mm256_set1_epi32((r.w10 & 0xffff) | (r.w11 << 16));
            Acc64 b1a, b2a;
            for (int y = 0; y < h; ++y) {
                const uint8_t* j0 = J.row(iny + y) + inx;
                const uint8_t* j1 = J.row(iny + y + 1) + inx;
                __m256i r1 = _mm256_setzero_si256(), r2 = _mm256_setzero_si256();       // a row's partial sums fit int32
                for (int x = 0; x < w; x += 8) {
                    __m256i jv = _mm256_srai_epi32(_mm256_add_epi32(_mm256_add_epi32(_mm256_madd_epi16(pairs_u8(j0 + x), v0), _mm256_madd_epi16(pairs_u8(j1 + x), v1)), r9), 9);
                    const __m256i ip = _mm256_cvtepi16_epi32(_mm_loadu_si128((const __m128i*)(Ip + (size_t)y * wp + x)));
                    const __m256i diff = _mm256_sub_epi32(jv, ip);          // lanes past the window: garbage times (ix, iy) = (0, 0)
                    const __m256i dd = _mm256_loadu_si256((const __m256i*)(dIp + 2 * ((size_t)y * wp + x)));
                    r1 = _mm256_add_epi32(r1, _mm256_madd_epi16(_mm256_and_si256(diff, _mm256_set1_epi32(0xffff)), dd));
                    r2 = _mm256_add_epi32(r2, _mm256_madd_epi16(_mm256_slli_epi32(diff, 16), dd));
                }
                b1a.add(r1); b2a.add(r2);
            }
            float b1 = (float)b1a.total() * FLT_SCALE, b2 = (float)b2a.total() * FLT_SCALE;
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
            const Weights r = bilinear_weights(ex, ey, inx, iny);
            const __m256i v0 = _mm256_set1_epi32((r.w00 & 0xffff) | (r.w01 << 16)), v1 = _mm256_set1_epi32((r.w10 & 0xffff) | (r.w11 << 16));
            Acc64 se;
            for (int y = 0; y < h; ++y) {
                const uint8_t* j0 = J.row(iny + y) + inx;
                const uint8_t* j1 = J.row(iny + y + 1) + inx;
                for (int x = 0; x < w; x += 8) {
                    __m256i jv = _mm256_srai_epi32(_mm256_add_epi32(_mm256_add_epi32(_mm256_madd_epi16(pairs_u8(j0 + x), v0), _mm256_madd_epi16(pairs_u8(j1 + x), v1)), r9), 9);
                    const __m256i ip = _mm256_cvtepi16_epi32(_mm_loadu_si128((const __m128i*)(Ip + (size_t)y * wp + x)));
                    se.add(_mm256_and_si256(_mm256_abs_epi32(_mm256_sub_epi32(jv, ip)), lane_mask(w - x)));
                }
            }
            err = (float)se.total() / (float)(32 * w * h);
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
inline float score_from_cov(float c0, float c1, float c2, int harris, double k, const Variant& V) {
    if (!harris) {
        float a = c0 * 0.5f, b = c1, c = c2 * 0.5f;
        float arg = V.mineig_fma ? fmaf(b, b, (a - c) * (a - c)) : (a - c) * (a - c) + b * b;
        return (a + c) - std::sqrt(arg);
    }
    if (V.harris == 1)           // OpenCV's scalar loop: float products, k double
        return (float)((double)(c0 * c2 - c1 * c1) - (k * (double)(c0 + c2)) * (double)(c0 + c2));
    float kf = (float)k;
    if (V.harris == 2)           // OpenCV's vector loop: (k*(a+c))*(a+c) in f32
        return (c0 * c2 - c1 * c1) - (kf * (c0 + c2)) * (c0 + c2);
    return (c0 * c2 - c1 * c1) - kf * ((c0 + c2) * (c0 + c2));
}

void corner_score(const uint8_t* img, int h, int w, int pitch, int harris, double k, float* score, int nthreads) {
    const Variant V = g_variant;
    const float s1 = (float)(1.0 / 3060.0), s2 = (float)(2.0 / 3060.0);
    std::vector<float> dx((size_t)h * w), dy((size_t)h * w);
    parallel_for(h, nthreads, [&](int y0, int y1) {
        for (int y = y0; y < y1; ++y) {
            const uint8_t* a = img + (size_t)reflect101(y - 1, h) * pitch;
            const uint8_t* b = img + (size_t)y * pitch;
            const uint8_t* c = img + (size_t)reflect101(y + 1, h) * pitch;
            for (int x = 0; x < w; ++x) {
                int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
                // Dx: row kernel [-1 0 1] (exact), column kernel [s 2s s]
                float r0 = (float)a[xp] - (float)a[xm], r1 = (float)b[xp] - (float)b[xm], r2 = (float)c[xp] - (float)c[xm];
                // Dy: row kernel [s 2s s] on u8, column kernel [-1 0 1]
                float q0, q2;
                if (V.sobel == 0) {
                    dx[(size_t)y * w + x] = s2 * r1 + s1 * (r0 + r2);
                    q0 = s2 * (float)a[x] + s1 * (float)(a[xm] + a[xp]);
                    q2 = s2 * (float)c[x] + s1 * (float)(c[xm] + c[xp]);
                    dy[(size_t)y * w + x] = q2 - q0;
                } else if (V.sobel == 3) {       // row smoothing tap by tap: (s*I[x-1] + 2s*I[x]) + s*I[x+1]
                    dx[(size_t)y * w + x] = s2 * r1 + s1 * (r0 + r2);
                    q0 = (s1 * (float)a[xm] + s2 * (float)a[x]) + s1 * (float)a[xp];
                    q2 = (s1 * (float)c[xm] + s2 * (float)c[x]) + s1 * (float)c[xp];
                    dy[(size_t)y * w + x] = q2 - q0;
                } else if (V.sobel == 1) {       // muladd in the smoothing passes: f0*S0 first, then fma of the outer taps
                    dx[(size_t)y * w + x] = fmaf(r0 + r2, s1, s2 * r1);
                    q0 = fmaf((float)(a[xm] + a[xp]), s1, s2 * (float)a[x]);
                    q2 = fmaf((float)(c[xm] + c[xp]), s1, s2 * (float)c[x]);
                    dy[(size_t)y * w + x] = q2 - q0;
                } else {                         // exact integer Sobel, one multiplication by the scale afterwards
                    dx[(size_t)y * w + x] = (2.f * r1 + (r0 + r2)) * s1;
                    int g = (2 * c[x] + c[xm] + c[xp]) - (2 * a[x] + a[xm] + a[xp]);
                    dy[(size_t)y * w + x] = (float)g * s1;
                }
            }
        }
    });
    if (V.box_running) {
        // OpenCV's boxFilter on the 3-channel product image: RowSum<float,double> (first 3 summed from 0, then
        // s += S[i+3] - S[i] along the bordered row), ColumnSum<double,float> (SUM += row for the first two
        // rows, then per output row s0 = SUM + new row; out = (float)s0; SUM = s0 - oldest row).  One stripe.
        std::vector<double> rs((size_t)3 * h * w);
        std::vector<float> br((size_t)3 * (w + 2));
        for (int y = 0; y < h; ++y) {
            for (int i = -1; i <= w; ++i) {
                int sx = reflect101(i, w);
                float gx = dx[(size_t)y * w + sx], gy = dy[(size_t)y * w + sx];
                br[3 * (i + 1)] = gx * gx; br[3 * (i + 1) + 1] = gx * gy; br[3 * (i + 1) + 2] = gy * gy;
            }
            for (int ch = 0; ch < 3; ++ch) {
                double sacc = 0;
                for (int i = 0; i < 3; ++i) sacc += (double)br[3 * i + ch];
                rs[((size_t)y * w) * 3 + ch] = sacc;
                for (int x = 0; x < w - 1; ++x) {
                    sacc += (double)br[3 * (x + 3) + ch] - (double)br[3 * x + ch];
                    rs[((size_t)y * w + x + 1) * 3 + ch] = sacc;
                }
            }
        }
        std::vector<double> SUM((size_t)3 * w, 0.0);
        auto rowp = [&](int y) { return rs.data() + (size_t)reflect101(y, h) * w * 3; };
        for (int y = -1; y <= 0; ++y) { const double* Sp = rowp(y); for (int i = 0; i < 3 * w; ++i) SUM[i] += Sp[i]; }
        for (int y = 0; y < h; ++y) {
            const double* Sp = rowp(y + 1);
            const double* Sm = rowp(y - 1);
            for (int x = 0; x < w; ++x) {
                float cc[3];
                for (int ch = 0; ch < 3; ++ch) {
                    double s0 = SUM[3 * x + ch] + Sp[3 * x + ch];
                    cc[ch] = (float)s0;
                    SUM[3 * x + ch] = s0 - Sm[3 * x + ch];
                }
                score[(size_t)y * w + x] = score_from_cov(cc[0], cc[1], cc[2], harris, k, V);
            }
        }
        return;
    }
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
                score[(size_t)y * w + x] = score_from_cov(c0, c1, c2, harris, k, V);
            }
        }
    });
}

// ---------------------------------------------------------------- score map, any blockSize / Sobel aperture
// cv::getSobelKernels (imgproc/src/deriv.cpp) for one axis: integer taps of the binomial smoothing kernel (order 0) or of the
// first-derivative kernel (order 1).  ksize 1 with order 1 becomes the 3-tap [-1 0 1] as in OpenCV.
std::vector<int> sobel_taps(int ksize, int order) {
    if (ksize == 1 && order > 0) ksize = 3;
    std::vector<int> k(ksize + 1, 0);
    if (ksize == 1) { k[0] = 1; }
    else if (ksize == 3) {
        if (order == 0) { k[0] = 1; k[1] = 2; k[2] = 1; }
        else { k[0] = -1; k[1] = 0; k[2] = 1; }
    } else {
        k[0] = 1;
        for (int i = 0; i < ksize - order - 1; ++i) {
            int oldval = k[0];
            for (int j = 1; j <= ksize; ++j) { int nv = k[j] + k[j - 1]; k[j - 1] = oldval; oldval = nv; }
        }
        for (int i = 0; i < order; ++i) {
            int oldval = -k[0];
            for (int j = 1; j <= ksize; ++j) { int nv = k[j - 1] - k[j]; k[j - 1] = oldval; oldval = nv; }
        }
    }
    k.resize(ksize);
    return k;
}

// cornerEigenValsVecs with any block_size >= 1 and aperture in {1, 3, 5, 7} (OpenCV corner.cpp; SURVEY A.1 generalised).
// Operation order (what the HIP kernel corner_score_generic follows bit for bit):
//  - scale = 1 / (2^(aperture-1) * block_size * 255) in double; the SMOOTHING kernel carries it: f[j] = (float)(tap[j] * scale)
//  - row pass u8 -> f32 (cv::RowFilter): s = k[0]*S[0]; s += k[j]*S[j], tap by tap, left to right; the 3-tap smoothing row
//    keeps the symmetric form of SURVEY A.1 (f[1]*I[x] + f[0]*(float)(I[x-1] + I[x+1])) so that aperture 3 is the same
//    arithmetic whatever the block size
//  - column pass f32 -> f32 (cv::SymmColumnFilter): symmetric s = f[c]*S[0]; s += f[c+k]*(S[k] + S[-k]);
//    antisymmetric s = 0; s += d[c+k]*(S[k] - S[-k])
//  - products in f32; box sum of block_size x block_size products in f64: each row left to right, then the rows top to
//    bottom, anchor block_size / 2, REFLECT_101 on the product maps; one rounding to f32
void corner_score_generic(const uint8_t* img, int h, int w, int pitch, int block, int aperture, int harris, double k,
                          float* score, int nthreads) {
    const Variant V = g_variant;
    const double scale = 1.0 / ((double)(1 << (aperture - 1)) * block * 255.0);
    const std::vector<int> dt = sobel_taps(aperture, 1), st = sobel_taps(aperture, 0);
    const int Ld = (int)dt.size(), Ls = (int)st.size(), hd = Ld / 2, hs = Ls / 2;
    std::vector<float> fs(Ls), fd(Ld);
    for (int j = 0; j < Ls; ++j) fs[j] = (float)((double)st[j] * scale);
    for (int j = 0; j < Ld; ++j) fd[j] = (float)dt[j];
    std::vector<float> R((size_t)h * w), Q((size_t)h * w), dx((size_t)h * w), dy((size_t)h * w);
    parallel_for(h, nthreads, [&](int y0, int y1) {
        for (int y = y0; y < y1; ++y) {
            const uint8_t* row = img + (size_t)y * pitch;
            for (int x = 0; x < w; ++x) {
                float s = fd[0] * (float)row[reflect101(x - hd, w)];
                for (int j = 1; j < Ld; ++j) s += fd[j] * (float)row[reflect101(x - hd + j, w)];
                R[(size_t)y * w + x] = s;
                float q;
                if (Ls == 3)
                    q = fs[1] * (float)row[x] + fs[0] * (float)((int)row[reflect101(x - 1, w)] + (int)row[reflect101(x + 1, w)]);
                else {
                    q = fs[0] * (float)row[reflect101(x - hs, w)];
                    for (int j = 1; j < Ls; ++j) q += fs[j] * (float)row[reflect101(x - hs + j, w)];
                }
                Q[(size_t)y * w + x] = q;
            }
        }
    });
    parallel_for(h, nthreads, [&](int y0, int y1) {
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < w; ++x) {
                float s = fs[hs] * R[(size_t)y * w + x];
                for (int t = 1; t <= hs; ++t)
                    s += fs[hs + t] * (R[(size_t)reflect101(y + t, h) * w + x] + R[(size_t)reflect101(y - t, h) * w + x]);
                dx[(size_t)y * w + x] = s;
                float d = 0.f;
                for (int t = 1; t <= hd; ++t)
                    d += fd[hd + t] * (Q[(size_t)reflect101(y + t, h) * w + x] - Q[(size_t)reflect101(y - t, h) * w + x]);
                dy[(size_t)y * w + x] = d;
            }
    });
    const int an = block / 2;
    parallel_for(h, nthreads, [&](int y0, int y1) {
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < w; ++x) {
                double t0 = 0, t1 = 0, t2 = 0;
                for (int j = 0; j < block; ++j) {
                    const int sy = reflect101(y - an + j, h);
                    double a0 = 0, a1 = 0, a2 = 0;
                    for (int i = 0; i < block; ++i) {
                        const int sx = reflect101(x - an + i, w);
                        const float gx = dx[(size_t)sy * w + sx], gy = dy[(size_t)sy * w + sx];
                        const float c0 = gx * gx, c1 = gx * gy, c2 = gy * gy;
                        if (i == 0) { a0 = c0; a1 = c1; a2 = c2; }
                        else { a0 += (double)c0; a1 += (double)c1; a2 += (double)c2; }
                    }
                    if (j == 0) { t0 = a0; t1 = a1; t2 = a2; }
                    else { t0 += a0; t1 += a1; t2 += a2; }
                }
                score[(size_t)y * w + x] = score_from_cov((float)t0, (float)t1, (float)t2, harris, k, V);
            }
    });
}

bool score_geometry_ok(int block, int aperture) {
    return block >= 1 && block <= 31 && (aperture == 1 || aperture == 3 || aperture == 5 || aperture == 7);
}

}  // namespace

// ================================================================= C ABI
extern "C" {

int vofe_cpu_set_variant(int lk_sum, int sobel, int mineig_fma, int box_running, int harris, int thresh_ipp) {
    if (lk_sum < 0 || lk_sum > 7 || sobel < 0 || sobel > 3 || harris < 0 || harris > 2) return -1;
    g_variant.lk_sum = lk_sum; g_variant.sobel = sobel; g_variant.mineig_fma = !!mineig_fma;
    g_variant.box_running = !!box_running; g_variant.harris = harris; g_variant.thresh_ipp = !!thresh_ipp;
    return 0;
}

int vofe_cpu_set_simd(int on) { g_simd = on != 0; return 0; }

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
    if (nthreads > 1) {                     // the two pyramids side by side, rows of the large levels over the threads
        std::thread other([&] { build_pyramid(N, next_img, h, w, pitch, win_w, win_h, max_level, false, std::max(1, nthreads / 4)); });
        build_pyramid(P, prev_img, h, w, pitch, win_w, win_h, max_level, true, std::max(1, nthreads - nthreads / 4));
        other.join();
    } else {
        build_pyramid(P, prev_img, h, w, pitch, win_w, win_h, max_level, true);
        build_pyramid(N, next_img, h, w, pitch, win_w, win_h, max_level, false);
    }
    parallel_for(n, nthreads, [&](int lo, int hi) {
        std::vector<int16_t> buf;
        for (int i = lo; i < hi; ++i) {
            status[i] = 1;
            err[i] = 0.f;
            float nx = next_pts[2 * i], ny = next_pts[2 * i + 1];
            if (g_simd && !g_variant.any_lk()) lk_point_simd(P, N, A, prev_pts[2 * i], prev_pts[2 * i + 1], nx, ny, status[i], err[i], buf);
            else if (g_variant.any_lk()) lk_point<true>(P, N, A, prev_pts[2 * i], prev_pts[2 * i + 1], nx, ny, status[i], err[i], buf);
            else lk_point<false>(P, N, A, prev_pts[2 * i], prev_pts[2 * i + 1], nx, ny, status[i], err[i], buf);
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

int vofe_cpu_corner_score_ex(const uint8_t* img, int h, int w, int pitch, int block_size, int aperture, int harris,
                             double k, float* score, int nthreads) {
    if (!img || !score || h <= 0 || w <= 0 || !score_geometry_ok(block_size, aperture)) return -1;
    if (block_size == 3 && aperture == 3) corner_score(img, h, w, pitch, harris, k, score, nthreads);
    else corner_score_generic(img, h, w, pitch, block_size, aperture, harris, k, score, nthreads);
    return 0;
}
int vofe_cpu_corner_score(const uint8_t* img, int h, int w, int pitch, int harris, double k,
                          float* score, int nthreads) {
    return vofe_cpu_corner_score_ex(img, h, w, pitch, 3, 3, harris, k, score, nthreads);
}
/* test aid: the generic path forced for block 3 / aperture 3 too (must reproduce the dedicated path bit for bit) */
int vofe_cpu_corner_score_generic(const uint8_t* img, int h, int w, int pitch, int block_size, int aperture, int harris,
                                  double k, float* score, int nthreads) {
    if (!img || !score || h <= 0 || w <= 0 || !score_geometry_ok(block_size, aperture)) return -1;
    corner_score_generic(img, h, w, pitch, block_size, aperture, harris, k, score, nthreads);
    return 0;
}

int vofe_cpu_gftt(const uint8_t* img, int h, int w, int pitch, int max_corners, double quality,
                  double min_distance, const uint8_t* mask, int harris, double k,
                  float* corners_xy, int capacity, int* n_out, int nthreads) {
    return vofe_cpu_gftt_ex(img, h, w, pitch, max_corners, quality, min_distance, mask, 3, 3, harris, k, corners_xy, capacity,
                            n_out, nthreads);
}

int vofe_cpu_gftt_ex(const uint8_t* img, int h, int w, int pitch, int max_corners, double quality,
                     double min_distance, const uint8_t* mask, int block_size, int aperture, int harris, double k,
                     float* corners_xy, int capacity, int* n_out, int nthreads) {
    if (!n_out) return -1;
    *n_out = 0;
    if (!(quality > 0) || !(min_distance >= 0) || max_corners < 0) return -2;     // SURVEY A.2 asserts
    if (!score_geometry_ok(block_size, aperture)) return -2;
    if (!img || h <= 0 || w <= 0) return 0;                                       // empty image -> None
    std::vector<float> eig((size_t)h * w);
    if (block_size == 3 && aperture == 3) corner_score(img, h, w, pitch, harris, k, eig.data(), nthreads);
    else corner_score_generic(img, h, w, pitch, block_size, aperture, harris, k, eig.data(), nthreads);
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
    if (g_variant.thresh_ipp) {                      // IPP-style TOZERO: v < thr + FLT_EPSILON -> 0
        const float thr2 = thr + 1.1920928955078125e-07f;
        for (auto& v : eig) v = v < thr2 ? 0.f : v;
    } else
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
