// Corner detection for gfx950: fused Sobel -> structure tensor -> 3x3 box -> min-eigenvalue /
// Harris score map, global max, threshold + 3x3 local-max compaction, deterministic parallel
// min-distance selection and a descending key sort.
//
// Replaces cv2.goodFeaturesToTrack (reference continuous.py:380-385, initialization.py:41-42):
// cv::cornerMinEigenVal / cornerHarris, minMaxLoc, threshold(TOZERO), dilate 3x3, the
// greaterThanPtr sort and the grid-based greedy minDistance loop.  Arithmetic and ordering per
// SURVEY.md A.1 / A.2; the float operation order is the canonical one of oracle/vofe_oracle.cpp
// (built with -ffp-contract=off), so score maps and corner lists are bit-identical to the oracle.
//
// The greedy minDistance pass is sequential in OpenCV.  Here it is a fixed-point iteration that
// provably yields the same set: a candidate is accepted once every
// higher-ranked candidate closer than minDistance is rejected, and rejected once one of them is
// accepted; ranks are (score, linear index) compared directly, so no global sort is needed before
// selection and only the accepted corners are sorted.
#include <cmath>
#include <cstring>
#include <vector>

#include <cstdio>

#include "vofe_internal.h"

namespace {

constexpr int TX = 64, TY = 16;                 // score tile (outputs) per 256-thread block
constexpr int SORT_CH = 8192;                   // keys sorted in LDS per block (64 KiB)
constexpr int SORT_THREADS = 1024;
constexpr int SELECT_ROUNDS = 8;                // rounds launched between host checks

constexpr int N_SEG = 8;                        // growing rank prefixes tried by the selection
constexpr int BUCKET_BITS = 14;                 // (bucket << 2 | state) must fit the u16 state map
static_assert(BUCKET_BITS <= 14, "bucket index is stored in 14 bits");
constexpr int N_BUCKET = 1 << BUCKET_BITS;      // score buckets of the coarse rank order (fine: few atomics collide)

constexpr size_t GFTT_BYTES_PER_PX = 8 + 4 + 4 + 4 + 4 + 2 + 1;   // workspace maps per pixel (vofe_gftt_impl)
constexpr int MAX_SLOTS = 32;
struct GfttHdr {
    unsigned max_key[MAX_SLOTS];   // order-preserving key of the global score maximum (0 = none), spread over slots:
                                   // same-address atomics serialise (~10 ns each), the readers take the max of all
    unsigned n_cand;
    unsigned n_acc;
    unsigned scat_lim;          // buckets below this one hold the first rank prefix: the first scatter launch takes only those
    unsigned undecided[SELECT_ROUNDS];
    unsigned seg_end[N_SEG];    // prefix lengths of the bucket-ordered candidate list (bucket boundaries)
    unsigned fin_state;         // finish_small: 1 = corners emitted, 2 = prefix too long for one block (host runs the split chain)
    unsigned bm_mb, bm_shift;   // the call's BucketMap (bucket_scan keeps it: max_key is zeroed for the next call)
};

// Coarse rank order: bucket 0 holds the highest scores.  Buckets partition the positive-float bit
// range (threshold, max] monotonically, so every candidate of a lower bucket outranks every
// candidate of a higher one; the order inside a bucket is arbitrary (ranks are always compared
// exactly on (score bits, index), the list order only steers the schedule and the prefix cut).
struct BucketMap { unsigned mb, shift; };
__device__ __forceinline__ BucketMap bucket_map(unsigned max_key, float thr) {
    BucketMap m;
    m.mb = max_key & 0x7fffffffu;                               // bits of the (positive) maximum
    const unsigned tb = __float_as_uint(thr);
    const unsigned range = m.mb > tb ? m.mb - tb : 0u;
    const int bits = 32 - __clz((int)range | 1);
    m.shift = bits > BUCKET_BITS ? (unsigned)(bits - BUCKET_BITS) : 0u;
    return m;
}
__device__ __forceinline__ unsigned bucket_of(const BucketMap& m, unsigned kb) { return (m.mb - kb) >> m.shift; }
__device__ __forceinline__ float gftt_threshold(unsigned mk, double quality);

__device__ __forceinline__ unsigned float_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ float gftt_threshold(unsigned mk, double quality) {
    return mk ? (float)((double)key_float(mk) * quality) : 0.f;
}

// ---------------------------------------------------------------------------------------------
// score map + global max
__global__ __launch_bounds__(256) void corner_score_f32(const uint8_t* __restrict__ img, int h, int w, int pitch,
                                                        size_t img_stride, int harris, float kf,
                                                        float* __restrict__ score, GfttHdr* __restrict__ hdr,
                                                        const uint8_t* __restrict__ mask) {
    __shared__ uint8_t sI[(TY + 4) * (TX + 4)];
    __shared__ float sC[3][(TY + 2) * (TX + 2)];
    __shared__ unsigned sMax;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const uint8_t* im = img + (size_t)blockIdx.z * img_stride;
    const int tid = threadIdx.x;
    if (tid == 0) sMax = 0;
    for (int e = tid; e < (TY + 4) * (TX + 4); e += 256) {
        const int j = e / (TX + 4), i = e - j * (TX + 4);
        const int sx = vofe_reflect101(x0 - 2 + i, w), sy = vofe_reflect101(y0 - 2 + j, h);
        sI[e] = im[(size_t)sy * pitch + sx];
    }
    __syncthreads();
    const float s1 = (float)(1.0 / 3060.0), s2 = (float)(2.0 / 3060.0);
    for (int e = tid; e < (TY + 2) * (TX + 2); e += 256) {
        const int j = e / (TX + 2), i = e - j * (TX + 2);
        const int cx = x0 - 1 + i, cy = y0 - 1 + j;
        if (cx < 0 || cx >= w || cy < 0 || cy >= h) continue;
        const uint8_t* p = sI + (j + 1) * (TX + 4) + (i + 1);      // centre pixel in the I tile
        const int a0 = p[-(TX + 4) - 1], a1 = p[-(TX + 4)], a2 = p[-(TX + 4) + 1];
        const int b0 = p[-1], b2 = p[1];
        const int c0 = p[(TX + 4) - 1], c1 = p[(TX + 4)], c2 = p[(TX + 4) + 1];
        const float r0 = (float)a2 - (float)a0, r1 = (float)b2 - (float)b0, r2 = (float)c2 - (float)c0;
        const float dx = s2 * r1 + s1 * (r0 + r2);
        const float q0 = s2 * (float)a1 + s1 * (float)(a0 + a2);
        const float q2 = s2 * (float)c1 + s1 * (float)(c0 + c2);
        const float dy = q2 - q0;
        sC[0][e] = dx * dx; sC[1][e] = dx * dy; sC[2][e] = dy * dy;
    }
    __syncthreads();
    unsigned best = 0;
    for (int e = tid; e < TY * TX; e += 256) {
        const int j = e / TX, i = e - j * TX;
        const int x = x0 + i, y = y0 + j;
        if (x >= w || y >= h) continue;
        int ci[3], cj[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            ci[t] = vofe_reflect101(x - 1 + t, w) - (x0 - 1);
            cj[t] = (vofe_reflect101(y - 1 + t, h) - (y0 - 1)) * (TX + 2);
        }
        float c[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            double rs[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float* row = sC[ch] + cj[t];
                rs[t] = ((double)row[ci[0]] + (double)row[ci[1]]) + (double)row[ci[2]];
            }
            c[ch] = (float)((rs[0] + rs[1]) + rs[2]);
        }
        float out;
        if (!harris) {
            const float a = c[0] * 0.5f, b = c[1], cc = c[2] * 0.5f;
            out = (a + cc) - sqrtf((a - cc) * (a - cc) + b * b);
        } else {
            out = (c[0] * c[2] - c[1] * c[1]) - kf * ((c[0] + c[2]) * (c[0] + c[2]));
        }
        score[((size_t)blockIdx.z * h + y) * w + x] = out;
        if (hdr && (!mask || mask[(size_t)y * w + x])) best = max(best, float_key(out));
    }
    if (hdr) {
        for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
        if ((tid & 63) == 0 && best) atomicMax(&sMax, best);
        __syncthreads();
        // same-address atomics serialise (~10 ns each): only blocks that can still raise the maximum issue one
        unsigned* slot = &hdr[blockIdx.z].max_key[(blockIdx.x + 5 * blockIdx.y) & (MAX_SLOTS - 1)];
        if (tid == 0 && sMax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, sMax);
    }
}

// ---------------------------------------------------------------------------------------------
// score map for any blockSize (1..31) / Sobel aperture (1, 3, 5, 7): cv2.goodFeaturesToTrack(..., blockSize, gradientSize)
// and cornerMinEigenVal / cornerHarris with those arguments (the reference passes neither, continuous.py:380-385, so block 3 /
// aperture 3 keeps its dedicated kernels).  Same structure as corner_score_f32, sizes at run time: image tile with the Sobel
// halo (REFLECT_101 at load), separable Sobel as OpenCV's filter engine runs it (row pass u8 -> f32 tap by tap, then the
// symmetric / antisymmetric column pass), products, f64 row sums of blockSize products, f64 column sums of blockSize row sums
// (anchor blockSize / 2, REFLECT_101 on the product maps: indices are reflected when the sums read, products exist only at
// image positions).  Operation order = oracle/vofe_oracle.cpp corner_score_generic, bit for bit.
// REFLECT_101 with the in-range case first (all but a frame's outermost pixels): one unsigned compare instead of the loop
__device__ __forceinline__ int refl101(int p, int n) { return (unsigned)p < (unsigned)n ? p : vofe_reflect101(p, n); }
struct ScoreGeom {
    int bs, an;                  // box size and anchor
    int ld, hd, ls, hs, hm;      // derivative / smoothing tap counts, their half widths, image halo = max(hd, hs)
    int tx, ty;                  // outputs per block
    float fd[8], fs[8];          // derivative taps (small integers), smoothing taps times the scale
};
__host__ __device__ inline size_t score_generic_lds(const ScoreGeom& g) {
    const size_t PW = g.tx + g.bs - 1, PH = g.ty + g.bs - 1, IW = PW + 2 * g.hm, IH = PH + 2 * g.hm;
    const size_t rq = 8 * IH * PW, hsum = 24 * PH * (size_t)g.tx;
    return (rq > hsum ? rq : hsum) + 12 * PH * PW + ((IH * IW + 15) & ~(size_t)15);
}
// BS / KS > 0: that blockSize / aperture at compile time (the loops over taps and box entries unroll; same operations in the
// same order), 0: run-time geometry.
template <int BS, int KS>
__global__ __launch_bounds__(256) void corner_score_generic(const uint8_t* __restrict__ img, int h, int w, int pitch,
                                                            size_t img_stride, ScoreGeom gg, int harris, float kf,
                                                            float* __restrict__ score, GfttHdr* __restrict__ hdr,
                                                            const uint8_t* __restrict__ mask) {
    struct G {
        int tx, ty;
        int bs, an, ld, hd, ls, hs, hm;
    } g;
    g.tx = gg.tx; g.ty = gg.ty;
    if (BS) { g.bs = BS; g.an = BS / 2; } else { g.bs = gg.bs; g.an = gg.an; }
    if (KS) {
        g.ld = KS <= 3 ? 3 : KS; g.ls = KS == 1 ? 1 : KS;
        g.hd = g.ld / 2; g.hs = g.ls / 2; g.hm = g.hd > g.hs ? g.hd : g.hs;
    } else { g.ld = gg.ld; g.hd = gg.hd; g.ls = gg.ls; g.hs = gg.hs; g.hm = gg.hm; }
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float sfd[8], sfs[8];
    __shared__ unsigned sMax;
    const int PW = g.tx + g.bs - 1, PH = g.ty + g.bs - 1, IW = PW + 2 * g.hm, IH = PH + 2 * g.hm;
    const size_t rq = 8 * (size_t)IH * PW, hsum = 24 * (size_t)PH * g.tx;
    // [ R | Q ] (row-pass outputs) share their space with the f64 row sums H[3], which are written after the column pass
    float* sR = reinterpret_cast<float*>(smem);
    float* sQ = sR + IH * PW;
    double* sH = reinterpret_cast<double*>(smem);
    float* sC = reinterpret_cast<float*>(smem + (rq > hsum ? rq : hsum));
    uint8_t* sI = reinterpret_cast<uint8_t*>(sC + 3 * PH * PW);
    const int x0 = blockIdx.x * g.tx, y0 = blockIdx.y * g.ty;
    const int px0 = x0 - g.an, py0 = y0 - g.an;
    const uint8_t* im = img + (size_t)blockIdx.z * img_stride;
    const int tid = threadIdx.x;
    if (tid == 0) sMax = 0;
    if (tid < 8) { sfd[tid] = gg.fd[tid]; sfs[tid] = gg.fs[tid]; }
    // (row, column) of an element index advance with it: one division per loop instead of one per element
#define SCORE_WALK(W) int j = tid / (W), i = tid - j * (W); const int dj_ = 256 / (W), di_ = 256 - dj_ * (W)
#define SCORE_STEP(W) do { i += di_; j += dj_; if (i >= (W)) { i -= (W); ++j; } } while (0)
    { SCORE_WALK(IW);
    for (int e = tid; e < IH * IW; e += 256) {
        const int sx = refl101(px0 - g.hm + i, w), sy = refl101(py0 - g.hm + j, h);
        sI[e] = im[(size_t)sy * pitch + sx];
        SCORE_STEP(IW);
    } }
    __syncthreads();
    // row pass at every tile row (rows beyond the image are reflected copies, so their row-pass values are the reflected
    // rows' values), at image columns only
    { SCORE_WALK(PW);
    for (int e = tid; e < IH * PW; e += 256) {
        const int cx = px0 + i;
        const uint8_t* p = sI + j * IW + i + g.hm;
        SCORE_STEP(PW);
        if (cx < 0 || cx >= w) continue;
        float r = sfd[0] * (float)p[-g.hd];
        for (int t = 1; t < g.ld; ++t) r += sfd[t] * (float)p[t - g.hd];
        float q;
        if (g.ls == 3) q = sfs[1] * (float)p[0] + sfs[0] * (float)((int)p[-1] + (int)p[1]);
        else {
            q = sfs[0] * (float)p[-g.hs];
            for (int t = 1; t < g.ls; ++t) q += sfs[t] * (float)p[t - g.hs];
        }
        sR[e] = r; sQ[e] = q;
    } }
    __syncthreads();
    // column pass and products at image positions of the product region
    float pc0 = 0.f, pc1 = 0.f, pc2 = 0.f;
    { SCORE_WALK(PW);
    for (int e0 = 0; e0 < PH * PW; e0 += 256) {
        const int e = e0 + tid;
        bool have = false;
        if (e < PH * PW) {
            const int cx = px0 + i, cy = py0 + j;
            if (cx >= 0 && cx < w && cy >= 0 && cy < h) {
                const float* r = sR + (j + g.hm) * PW + i;
                const float* q = sQ + (j + g.hm) * PW + i;
                float dx = sfs[g.hs] * r[0];
                for (int t = 1; t <= g.hs; ++t) dx += sfs[g.hs + t] * (r[t * PW] + r[-t * PW]);
                float dy = 0.f;
                for (int t = 1; t <= g.hd; ++t) dy += sfd[g.hd + t] * (q[t * PW] - q[-t * PW]);
                pc0 = dx * dx; pc1 = dx * dy; pc2 = dy * dy;
                have = true;
            }
        }
        if (have) { sC[e] = pc0; sC[PH * PW + e] = pc1; sC[2 * PH * PW + e] = pc2; }
        SCORE_STEP(PW);
    } }
    __syncthreads();                     // R / Q are dead from here: the row sums take their place
    { SCORE_WALK(g.tx);
    for (int e = tid; e < PH * g.tx; e += 256) {
        const int x = x0 + i, cy = py0 + j;
        const float* c = sC + j * PW;
        SCORE_STEP(g.tx);
        if (x >= w || cy < 0 || cy >= h) continue;
        const int i0 = refl101(x - g.an, w) - px0;
        double a0 = c[i0], a1 = c[PH * PW + i0], a2 = c[2 * PH * PW + i0];
        for (int t = 1; t < g.bs; ++t) {
            const int it = refl101(x - g.an + t, w) - px0;
            a0 += (double)c[it]; a1 += (double)c[PH * PW + it]; a2 += (double)c[2 * PH * PW + it];
        }
        sH[e] = a0; sH[PH * g.tx + e] = a1; sH[2 * PH * g.tx + e] = a2;
    } }
    __syncthreads();
    unsigned best = 0;
    { SCORE_WALK(g.tx);
    for (int e = tid; e < g.ty * g.tx; e += 256) {
        const int x = x0 + i, y = y0 + j;
        const int ii = i;
        SCORE_STEP(g.tx);
        if (x >= w || y >= h) continue;
        const int j0 = (refl101(y - g.an, h) - py0) * g.tx + ii;
        double t0 = sH[j0], t1 = sH[PH * g.tx + j0], t2 = sH[2 * PH * g.tx + j0];
        for (int t = 1; t < g.bs; ++t) {
            const int jt = (refl101(y - g.an + t, h) - py0) * g.tx + ii;
            t0 += sH[jt]; t1 += sH[PH * g.tx + jt]; t2 += sH[2 * PH * g.tx + jt];
        }
        const float c0 = (float)t0, c1 = (float)t1, c2 = (float)t2;
        float out;
        if (!harris) {
            const float a = c0 * 0.5f, b = c1, cc = c2 * 0.5f;
            out = (a + cc) - sqrtf((a - cc) * (a - cc) + b * b);
        } else {
            out = (c0 * c2 - c1 * c1) - kf * ((c0 + c2) * (c0 + c2));
        }
        score[((size_t)blockIdx.z * h + y) * w + x] = out;
        if (hdr && (!mask || mask[(size_t)y * w + x])) best = max(best, float_key(out));
    } }
#undef SCORE_WALK
#undef SCORE_STEP
    if (hdr) {
        for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
        if ((tid & 63) == 0 && best) atomicMax(&sMax, best);
        __syncthreads();
        unsigned* slot = &hdr[blockIdx.z].max_key[(blockIdx.x + 5 * blockIdx.y) & (MAX_SLOTS - 1)];
        if (tid == 0 && sMax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, sMax);
    }
}

// ---------------------------------------------------------------------------------------------
// score map, register-strip version (frames with w >= 8, h >= 4): one thread owns 4 adjacent columns
// and walks RS output rows; no LDS.  Per image row it loads the 8 bytes covering columns x-2..x+5 with
// one unaligned load, keeps a 3-row ring of the horizontal Sobel terms, a 3-row ring of the f64
// horizontal box sums of the three products, and emits one row of 4 scores per step.  All float/double
// operations are the ones of corner_score_f32 above, in the same order.
//   Borders (REFLECT_101 on the image for the Sobel taps, REFLECT_101 on the product maps for the box
// sums) cost no per-pixel select: pixels are fetched through the reflection, so the product row "-1"
// is computed from image rows 2, 1, 0 - its dx equals row 1's (f32 addition commutes) and its dy is
// row 1's negated (exactly): it IS product row 1 with dx*dy negated; likewise row h against row h-2
// and, with dx and dy trading places, columns -1 and w.  Rows: the image row index is reflected and
// the xy row sums of a reflected row change sign (a wave-uniform branch taken on at most two steps per
// frame).  Columns: the 8 loaded bytes go through two v_perm_b32 whose per-thread selectors do the
// alignment shift and the reflection at once, and five per-thread sign masks flip dx*dy of a
// reflected product column (all zero except in the frame's first and last thread of a row).
typedef unsigned long long u64_unaligned __attribute__((aligned(1)));

// Correctly rounded sqrtf for an argument that is +0 or >= 2^-96: the compiler's own expansion (v_sqrt_f32, then the
// neighbours one ulp below and above tested by the sign of their fused residuals) minus its rescaling of tiny
// arguments and its zero/infinity select (a zero argument passes through the tests unchanged: both residuals compare
// false).  The caller checks the range, wave-uniformly, and uses sqrtf otherwise.
__device__ __forceinline__ float sqrt_rn_normal(float s) {
    const float r = __builtin_amdgcn_sqrtf(s);
    const float rm = __uint_as_float(__float_as_uint(r) - 1u), rp = __uint_as_float(__float_as_uint(r) + 1u);
    const float em = __builtin_fmaf(-rm, r, s), ep = __builtin_fmaf(-rp, r, s);
    float q = (0.f >= em) ? rm : r;
    q = (0.f < ep) ? rp : q;
    return q;
}

template <int RS>
__global__ __launch_bounds__(256) void corner_score_strip(const uint8_t* __restrict__ img, int h, int w, int pitch,
                                                          size_t img_stride, int harris, float kf,
                                                          float* __restrict__ score, GfttHdr* __restrict__ hdr,
                                                          const uint8_t* __restrict__ mask) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = ((int)blockIdx.x * 64 + lane) * 4;
    const int y0 = ((int)blockIdx.y * 4 + wv) * RS;
    if (y0 >= h) return;                                  // whole wave (no block barrier below)
    const uint8_t* im = img + (size_t)blockIdx.z * img_stride;
    const bool live = x < w;
    const int xa = min(max(x - 2, 0), w - 8);             // first loaded column
    const float s1 = (float)(1.0 / 3060.0), s2 = (float)(2.0 / 3060.0);

    // f[i] (i = 0..7) is image column x-2+i through REFLECT_101, as a byte index into the 8 loaded bytes
    unsigned sel_lo = 0, sel_hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = x - 2 + i;
        c = c < 0 ? -c : c;
        c = c >= w ? 2 * w - 2 - c : c;
        const unsigned b = (unsigned)min(max(c - xa, 0), 7);      // clamped: bytes that only feed unused columns
        if (i < 4) sel_lo |= b << (8 * i); else sel_hi |= b << (8 * (i - 4));
    }
    // product column c (c = 0..5) is image column x-1+c; a reflected one (-1 or w) carries -dx*dy
    unsigned sgn[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) sgn[c] = (x - 1 + c < 0 || x - 1 + c == w) ? 0x80000000u : 0u;

    unsigned long long bits[RS + 4];
#pragma unroll
    for (int t = 0; t < RS + 4; ++t) {
        int a = y0 - 2 + t;
        a = a < 0 ? -a : a;
        a = a >= h ? 2 * h - 2 - a : a;
        a = min(max(a, 0), h - 1);                        // rows that only feed unused outputs
        bits[t] = *reinterpret_cast<const __attribute__((address_space(1))) u64_unaligned*>(
            (const __attribute__((address_space(1))) uint8_t*)(im + (size_t)a * pitch + xa));
    }

    float hd[3][6], hs[3][6];
    double rsum[3][3][4];
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < RS + 4; ++t) {
        // ---- horizontal Sobel terms of image row y0-2+t
        {
            const unsigned rl = (unsigned)bits[t], rh = (unsigned)(bits[t] >> 32);
            const unsigned lo = __builtin_amdgcn_perm(rh, rl, sel_lo), hi = __builtin_amdgcn_perm(rh, rl, sel_hi);
            float f[8];
            f[0] = (float)(lo & 0xffu); f[1] = (float)((lo >> 8) & 0xffu); f[2] = (float)((lo >> 16) & 0xffu); f[3] = (float)(lo >> 24);
            f[4] = (float)(hi & 0xffu); f[5] = (float)((hi >> 8) & 0xffu); f[6] = (float)((hi >> 16) & 0xffu); f[7] = (float)(hi >> 24);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                hd[t % 3][c] = f[c + 2] - f[c];
                hs[t % 3][c] = s2 * f[c + 1] + s1 * (f[c] + f[c + 2]);
            }
        }
        if (t < 2) continue;
        // ---- products of row v = y0-3+t (image rows t-2, t-1, t of the ring) and their f64 row sums
        {
            const int top = (t - 2) % 3, mid = (t - 1) % 3, bot = t % 3;
            float pxx[6], pxy[6], pyy[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const float dx = s2 * hd[mid][c] + s1 * (hd[top][c] + hd[bot][c]);
                const float dy = hs[bot][c] - hs[top][c];
                pxx[c] = dx * dx; pyy[c] = dy * dy;
                pxy[c] = __uint_as_float(__float_as_uint(dx * dy) ^ sgn[c]);
            }
            const int slot = t % 3;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                rsum[slot][0][o] = ((double)pxx[o] + (double)pxx[o + 1]) + (double)pxx[o + 2];
                rsum[slot][1][o] = ((double)pxy[o] + (double)pxy[o + 1]) + (double)pxy[o + 2];
                rsum[slot][2][o] = ((double)pyy[o] + (double)pyy[o + 1]) + (double)pyy[o + 2];
            }
            const int v = y0 - 3 + t;
            if (v < 0 || v >= h) {                        // uniform: a reflected row
#pragma unroll
                for (int o = 0; o < 4; ++o) rsum[slot][1][o] = -rsum[slot][1][o];
            }
        }
        if (t < 4) continue;
        // ---- output row y = y0-4+t from the row sums of rows y-1, y, y+1 (slots t-2, t-1, t)
        const int y = y0 - 4 + t;
        if (y >= h) break;                                // uniform
        float out[4];
        float c[4][3];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                c[o][ch] = (float)((rsum[(t - 2) % 3][ch][o] + rsum[(t - 1) % 3][ch][o]) + rsum[t % 3][ch][o]);
        if (!harris) {
            float tr[4], ss[4];
            unsigned lowest = 0xffffffffu;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float a = c[o][0] * 0.5f, b = c[o][1], cc = c[o][2] * 0.5f;
                tr[o] = a + cc;
                ss[o] = (a - cc) * (a - cc) + b * b;                         // >= +0
                lowest = min(lowest, __float_as_uint(ss[o]) - 1u);          // +0 wraps to the top
            }
            if (!__any(lowest < 0x0f800000u - 1u)) {                        // nothing in (0, 2^-96): the usual case
#pragma unroll
                for (int o = 0; o < 4; ++o) out[o] = tr[o] - sqrt_rn_normal(ss[o]);
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o) out[o] = tr[o] - sqrtf(ss[o]);
            }
        } else {
#pragma unroll
            for (int o = 0; o < 4; ++o)
                out[o] = (c[o][0] * c[o][2] - c[o][1] * c[o][1]) - kf * ((c[o][0] + c[o][2]) * (c[o][0] + c[o][2]));
        }
        if (live) {
            float* dst = score + ((size_t)blockIdx.z * h + y) * w + x;
            if (x + 3 < w) {
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                f4u q; q.x = out[0]; q.y = out[1]; q.z = out[2]; q.w = out[3];
                *reinterpret_cast<f4u*>(dst) = q;
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o) if (x + o < w) dst[o] = out[o];
            }
        }
        if (hdr) {                                        // uniform
            if (mask) {                                   // uniform
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (x + o < w && mask[(size_t)y * w + x + o]) best[o] = fmaxf(best[o], out[o]);
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o) best[o] = fmaxf(best[o], out[o]);   // columns >= w are dropped below
            }
        }
    }
    if (hdr) {
        // scores are never -inf or NaN, so the float maximum orders as the keys do (the sign of a zero maximum is
        // immaterial: the threshold it yields compares the same)
        float bf = -INFINITY;
#pragma unroll
        for (int o = 0; o < 4; ++o) if (x + o < w) bf = fmaxf(bf, best[o]);
        unsigned bk = bf == -INFINITY ? 0u : float_key(bf);
        for (int o = 32; o > 0; o >>= 1) bk = max(bk, (unsigned)__shfl_xor((int)bk, o));
        unsigned* slot = &hdr[blockIdx.z].max_key[(blockIdx.x * 4 + wv + 7 * blockIdx.y) & (MAX_SLOTS - 1)];
        if (lane == 0 && bk) atomicMax(slot, bk);               // no pre-check load: it would only add an L2 round trip
    }
}

// ---------------------------------------------------------------------------------------------
// threshold + 3x3 local maximum -> candidate maps + bucket histogram.
// state map (u16 per pixel): 0 = no candidate, else (score bucket << 2) | state, state 1 = undecided;
// ckey map: score bits of candidates, 0 elsewhere.  No list is built here: the bucket scatter below
// reads the state map.  Histogram and scatter cursors exist in N_COPY copies chosen by pixel tile,
// which divides the same-address atomic traffic of dense buckets.
// rows per thread (template parameter of nms_map): a 256-thread block covers 64 x 4*ROWS pixels.  Every row costs a
// thread two dependent memory round trips (the score, then -- for pixels above the threshold -- the eight neighbours),
// so a small frame takes fewer rows per thread and more waves, a large one amortises the per-wave set-up over 8 rows
constexpr int NMS_ROWS_MAX = 8;
constexpr int N_COPY = 1;        // measured: more copies do not speed the atomics up, they only lengthen the scan
__device__ __forceinline__ unsigned tile_copy(int x, int y) { return (unsigned)((x >> 6) + (y >> 5)) & (N_COPY - 1); }   // (N_COPY == 1: always 0)

// ---------------------------------------------------------------------------------------------
// exclusive scan of the bucket histogram in (bucket, copy) order: one 1024-thread block per image,
// N_BUCKET/1024 consecutive buckets per thread -> scatter cursors, the candidate count, and the
// prefix lengths seg_end[k] = first bucket boundary >= min(n, base << k) (base = 0: everything).
static_assert(N_COPY == 1, "the scan below handles one histogram copy");
template <int NT>
__device__ __forceinline__ void bucket_scan_body(unsigned* __restrict__ hist_img, unsigned* __restrict__ cursor_img,
                                                 GfttHdr* __restrict__ hd, unsigned base, unsigned full_ratio, double quality,
                                                 unsigned* s_w, unsigned* s_n) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (wv == 0) {                                       // the call's bucket map, for the one-block last stage
        unsigned mk = hd->max_key[lane & (MAX_SLOTS - 1)];
        for (int o = MAX_SLOTS / 2; o > 0; o >>= 1) mk = max(mk, (unsigned)__shfl_xor((int)mk, o));
        const BucketMap bm = bucket_map(mk, gftt_threshold(mk, quality));
        if (lane == 0) { hd->bm_mb = bm.mb; hd->bm_shift = bm.shift; }
    }
    constexpr int BPT = N_BUCKET / NT;                   // consecutive buckets per thread
    static_assert(BPT % 4 == 0, "uint4 access");
    unsigned cnt[BPT];
    unsigned mine = 0;
#pragma unroll
    for (int i = 0; i < BPT / 4; ++i) {
        const uint4 v = reinterpret_cast<const uint4*>(hist_img)[t * (BPT / 4) + i];
        cnt[4 * i] = v.x; cnt[4 * i + 1] = v.y; cnt[4 * i + 2] = v.z; cnt[4 * i + 3] = v.w;
        mine += v.x + v.y + v.z + v.w;
        // the histogram has been consumed: leave it zero for the next call (no memset command per call)
        reinterpret_cast<uint4*>(hist_img)[t * (BPT / 4) + i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // likewise the header fields the next call expects to be zero; max_key was last read by the NMS kernel, the accepted
    // count and the round flags are first touched by the selection that follows this kernel
    if (t < SELECT_ROUNDS) hd->undecided[t] = 0u;
    if (t == SELECT_ROUNDS) hd->n_acc = 0u;
    unsigned inc = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    if (t >= 64 && t < 64 + MAX_SLOTS) hd->max_key[t - 64] = 0u;       // (after the first wave has read the maximum)
    unsigned off = 0;
    for (int q = 0; q < wv; ++q) off += s_w[q];
    if (t == NT - 1) { *s_n = off + inc; hd->n_cand = off + inc; }
    __syncthreads();
    const unsigned n = *s_n;
    unsigned e0 = off + inc - mine;
    unsigned first = e0;
#pragma unroll
    for (int i = 0; i < BPT; ++i) { const unsigned k = cnt[i]; cnt[i] = e0; e0 += k; }     // cnt[i] := start of bucket i
#pragma unroll
    for (int i = 0; i < BPT / 4; ++i)
        reinterpret_cast<uint4*>(cursor_img)[t * (BPT / 4) + i] = make_uint4(cnt[4 * i], cnt[4 * i + 1], cnt[4 * i + 2], cnt[4 * i + 3]);
    for (int k = 0; k < N_SEG; ++k) {
        unsigned long long want = base ? ((unsigned long long)base << k) : (unsigned long long)n;
        // a prefix that is already a sizeable part of the list is not worth a possible second pass
        if (k == N_SEG - 1 || want > n || (unsigned long long)n <= (unsigned long long)full_ratio * base) want = n;
        const unsigned tk = (unsigned)want;
        if (tk == 0) { if (t == 0) { hd->seg_end[k] = 0; if (k == 0) hd->scat_lim = 0; } continue; }
        if (!(first < tk && e0 >= tk)) continue;              // the boundary bucket is not among this thread's
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const unsigned end_i = (i + 1 < BPT) ? cnt[i + 1] : e0;
            if (cnt[i] < tk && end_i >= tk) {
                hd->seg_end[k] = end_i;
                if (k == 0) hd->scat_lim = (unsigned)(t * BPT + i + 1);
            }
        }
    }
}

__global__ __launch_bounds__(1024) void bucket_scan(unsigned* __restrict__ hist, unsigned* __restrict__ cursor,
                                                    GfttHdr* __restrict__ hdr, unsigned base, unsigned full_ratio, double quality) {
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_n;
    bucket_scan_body<1024>(hist + (size_t)blockIdx.x * N_BUCKET, cursor + (size_t)blockIdx.x * N_BUCKET, hdr + blockIdx.x,
                           base, full_ratio, quality, s_w, &s_n);
}

template <int NMS_ROWS>
__global__ __launch_bounds__(256) void nms_map(const float* __restrict__ score, int h, int w, double quality,
                                               GfttHdr* __restrict__ hdr, uint16_t* __restrict__ state,
                                               unsigned* __restrict__ ckey, unsigned* __restrict__ hist,
                                               const uint8_t* __restrict__ mask) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = blockIdx.y * (4 * NMS_ROWS) + wv;
    const size_t ib = (size_t)blockIdx.z * h * w;
    const float* sc = score + ib;
    unsigned mk = hdr[blockIdx.z].max_key[lane & (MAX_SLOTS - 1)];       // wave-wide maximum of the slots
    for (int o = MAX_SLOTS / 2; o > 0; o >>= 1) mk = max(mk, (unsigned)__shfl_xor((int)mk, o));
    const float thr = gftt_threshold(mk, quality);
    const BucketMap bm = bucket_map(mk, thr);
    unsigned* hs = hist + ((size_t)blockIdx.z * N_COPY + tile_copy(blockIdx.x * 64, blockIdx.y * (4 * NMS_ROWS))) * N_BUCKET;
    if (x < w) {
#pragma unroll
        for (int k = 0; k < NMS_ROWS; ++k) {
            const int y = y0 + 4 * k;
            if (y >= h) break;
            bool is = false;
            float v = 0.f;
            if (mk && x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
                const float* p = sc + (size_t)y * w + x;
                v = p[0];
                if (v > thr && v != 0.f && (!mask || mask[(size_t)y * w + x])) {
                    float m = fmaxf(fmaxf(p[-w - 1], p[-w]), p[-w + 1]);
                    m = fmaxf(m, fmaxf(p[-1], p[1]));
                    m = fmaxf(m, fmaxf(fmaxf(p[w - 1], p[w]), p[w + 1]));
                    is = v >= m;
                }
            }
            const unsigned d = is ? bucket_of(bm, __float_as_uint(v)) : 0u;
            state[ib + (size_t)y * w + x] = is ? (uint16_t)((d << 2) | 1u) : (uint16_t)0;
            ckey[ib + (size_t)y * w + x] = is ? __float_as_uint(v) : 0u;
            if (is) atomicAdd(&hs[d], 1u);
        }
    }
}

// bucket-ordered candidate list, straight from the state map (8 pixels per thread)
// Beside the pixel index the list carries, by list position, the candidate's score bits (lkey) and an "accepted" flag
// (lflag, cleared here, set by the selection), so that the last stage reads three contiguous arrays instead of chasing
// list -> state map -> key map.
__global__ __launch_bounds__(256) void bucket_scatter(const uint16_t* __restrict__ state, const unsigned* __restrict__ ckey, int h, int w,
                                                      unsigned* __restrict__ cursor, unsigned* __restrict__ sorted,
                                                      unsigned* __restrict__ lkey, uint8_t* __restrict__ lflag,
                                                      const GfttHdr* __restrict__ hdr, int rest) {
    // rest == 0: only the buckets of the first rank prefix (what the selection reads; on a 4K frame that is a tenth of the
    // candidates, and every scattered candidate is a returning memory-side atomic); rest == 1: the others, launched only
    // when a pass has to widen the prefix.  The cursors of the two sets of buckets are disjoint.
    const unsigned lim = hdr[blockIdx.y].scat_lim;
    const size_t px = (size_t)h * w;
    const uint16_t* stt = state + (size_t)blockIdx.y * px;
    unsigned* cs = cursor + (size_t)blockIdx.y * N_COPY * N_BUCKET;
    // the state base is 16-byte aligned; image b starts b*px elements in, so groups of 8 are shifted
    // to stay aligned in memory (the first group of an image may start before element 0)
    const long long mis = (long long)(((size_t)blockIdx.y * px) & 7);
    for (long long e0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8 - mis; e0 < (long long)px;
         e0 += (long long)gridDim.x * 256 * 8) {
        unsigned v[8];
        if (e0 >= 0 && e0 + 8 <= (long long)px) {
            const uint4 q = *reinterpret_cast<const uint4*>(stt + e0);
            v[0] = q.x & 0xffffu; v[1] = q.x >> 16; v[2] = q.y & 0xffffu; v[3] = q.y >> 16;
            v[4] = q.z & 0xffffu; v[5] = q.z >> 16; v[6] = q.w & 0xffffu; v[7] = q.w >> 16;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (e0 + i >= 0 && e0 + i < (long long)px) ? stt[e0 + i] : 0u;
        }
        if (!(v[0] | v[1] | v[2] | v[3] | v[4] | v[5] | v[6] | v[7])) continue;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (!v[i] || ((v[i] >> 2) < lim) == (rest != 0)) continue;
            const unsigned p = (unsigned)(e0 + i);
            const int y = (int)(p / (unsigned)w), x = (int)(p - (unsigned)y * w);
            const unsigned pos = atomicAdd(&cs[(size_t)tile_copy(x, y) * N_BUCKET + (v[i] >> 2)], 1u);
            sorted[(size_t)blockIdx.y * px + pos] = p;
            lkey[(size_t)blockIdx.y * px + pos] = ckey[(size_t)blockIdx.y * px + p];
            lflag[(size_t)blockIdx.y * px + pos] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// min-distance fixed point: one wavefront per candidate, grid-stride over the bucket-ordered list.
// State map entry: (bucket << 2) | state, state 1 undecided, 2 accepted, 3 rejected, 0 = no
// candidate.  The bucket is a coarse rank, so one 2-byte load per window pixel tells whether a
// neighbour exists, whether it outranks the centre (exact (score bits, index) comparison through
// ckey only when the buckets tie) and what its state is.  Any schedule gives the same fixed point
// because a decision is only taken from final (2/3) states of higher-ranked neighbours.  One launch
// makes `passes` sweeps over its candidates and re-reads the blocking neighbours up to `spins`
// times per visit; state traffic goes through L2 (agent-scope relaxed atomics = sc1 loads/stores)
// so sweeps see other waves' decisions.
__device__ __forceinline__ unsigned ld_state(const uint16_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(uint16_t* p, unsigned v) {
    __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int SEL_CH = 6;        // window entries per lane held in registers (windows up to 19x19)

__device__ __forceinline__ bool outranks(const unsigned* __restrict__ ck, unsigned vq, unsigned q, unsigned dp, unsigned p) {
    const unsigned dq = vq >> 2;
    if (dq != dp) return dq < dp;
    const unsigned kq = ck[q], kp = ck[p];
    return kq > kp || (kq == kp && q > p);
}

struct SelGeom { int h, w, r, side, area; float md2, inv_side; };
struct SelWin { unsigned p, myv, q[SEL_CH], v[SEL_CH]; };

// issue the loads of one candidate's window (register-resident windows only)
__device__ __forceinline__ void sel_fetch(const SelGeom& g, const uint16_t* stt, unsigned p, int lane, SelWin& W) {
    W.p = p;
    W.myv = ld_state(stt + p);
    const int py = (int)(p / (unsigned)g.w), px = (int)(p - (unsigned)py * g.w);
#pragma unroll
    for (int i = 0; i < SEL_CH; ++i) {
        const int e = lane + 64 * i;
        const int ey = (int)(((float)e + 0.5f) * g.inv_side);                 // e / side, exact for e < 4096
        const int dy = ey - g.r, dx = e - ey * g.side - g.r;
        const int qx = px + dx, qy = py + dy;
        const bool ok = e < g.area && qx >= 0 && qx < g.w && qy >= 0 && qy < g.h && (float)(dx * dx + dy * dy) < g.md2;
        W.q[i] = ok ? (unsigned)(qy * g.w + qx) : p;
        W.v[i] = ld_state(stt + W.q[i]);
    }
}

// decide one candidate from its (possibly slightly old) window; blocking neighbours are re-read.
// Returns true when the candidate is still undecided.
__device__ __forceinline__ bool sel_eval(const unsigned* __restrict__ ck, uint16_t* stt, const SelWin& W, int lane, int spins, uint8_t* lf) {
    if ((W.myv & 3u) != 1u) return false;
    const unsigned dp = W.myv >> 2, p = W.p;
    bool rel[SEL_CH];
    bool rej = false, blk = false;
#pragma unroll
    for (int i = 0; i < SEL_CH; ++i) {
        const unsigned st = W.v[i] & 3u;
        rel[i] = false;
        if (W.q[i] == p || st == 0u || st == 3u) continue;
        if (!outranks(ck, W.v[i], W.q[i], dp, p)) continue;
        if (st == 2u) rej = true; else { blk = true; rel[i] = true; }
    }
    bool any_rej = __any(rej), any_blk = __any(blk);
    for (int spin = 0; !any_rej && any_blk && spin < spins; ++spin) {
        if (spin) __builtin_amdgcn_s_sleep(2);
        rej = false; blk = false;
#pragma unroll
        for (int i = 0; i < SEL_CH; ++i) {
            if (!rel[i]) continue;
            const unsigned st = ld_state(stt + W.q[i]) & 3u;
            if (st == 2u) rej = true; else if (st == 1u) blk = true; else rel[i] = false;
        }
        any_rej = __any(rej); any_blk = __any(blk);
    }
    if (any_rej) { if (lane == 0) st_state(stt + p, (dp << 2) | 3u); return false; }
    if (!any_blk) { if (lane == 0) { st_state(stt + p, (dp << 2) | 2u); *lf = 1; } return false; }
    return true;
}

// windows too large for registers: plain scan
__device__ __forceinline__ bool sel_visit_large(const SelGeom& g, const unsigned* __restrict__ ck, uint16_t* stt, unsigned p, int lane, uint8_t* lf) {
    const unsigned myv = ld_state(stt + p);
    if ((myv & 3u) != 1u) return false;
    const unsigned dp = myv >> 2;
    const int py = (int)(p / (unsigned)g.w), px = (int)(p - (unsigned)py * g.w);
    bool rej = false, blk = false;
    for (int e = lane; e < g.area; e += 64) {
        const int dy = e / g.side - g.r, dx = e - (dy + g.r) * g.side - g.r;
        const int qx = px + dx, qy = py + dy;
        if (qx < 0 || qx >= g.w || qy < 0 || qy >= g.h) continue;
        if ((float)(dx * dx + dy * dy) >= g.md2) continue;
        const unsigned q = (unsigned)(qy * g.w + qx);
        const unsigned vq = ld_state(stt + q), st = vq & 3u;
        if (q == p || st == 0u || st == 3u || !outranks(ck, vq, q, dp, p)) continue;
        if (st == 2u) rej = true; else blk = true;
    }
    if (__any(rej)) { if (lane == 0) st_state(stt + p, (dp << 2) | 3u); return false; }
    if (!__any(blk)) { if (lane == 0) { st_state(stt + p, (dp << 2) | 2u); *lf = 1; } return false; }
    return true;
}

// A wave owns candidates wave, wave + nwaves, ... of the bucket-ordered list, in chunks of 64: a first
// sweep with the next candidate's window loads in flight while the current one is decided, then up
// to passes-1 sweeps over the ones still blocked (bit mask).
__global__ __launch_bounds__(256) void select_round(const unsigned* __restrict__ ckey, int h, int w, int r, float md2,
                                                    GfttHdr* __restrict__ hdr, uint16_t* __restrict__ state,
                                                    const unsigned* __restrict__ cand, uint8_t* __restrict__ lflag, int seg, int passes,
                                                    int spins, int slot) {
    if (slot > 0 && !hdr[blockIdx.y].undecided[slot - 1]) return;     // the previous launch settled everything
    const size_t ib = (size_t)blockIdx.y * h * w;
    const unsigned* ck = ckey + ib;
    const unsigned* cl = cand + ib;
    uint8_t* lf = lflag + ib;
    uint16_t* stt = state + ib;
    const unsigned n = hdr[blockIdx.y].seg_end[seg];
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    SelGeom g;
    g.h = h; g.w = w; g.r = r; g.side = 2 * r + 1; g.area = g.side * g.side; g.md2 = md2; g.inv_side = 1.0f / (float)g.side;
    const bool small = g.area <= 64 * SEL_CH;
    bool left = false;
    for (unsigned c0 = wave; c0 < n; c0 += nwaves * 64u) {
        const unsigned kmax = min(64u, (n - c0 + nwaves - 1) / nwaves);
        unsigned long long pending = 0;
        if (small) {
            SelWin cur;
            sel_fetch(g, stt, cl[c0], lane, cur);
            unsigned p_next = kmax > 1 ? cl[c0 + nwaves] : 0u;
            for (unsigned k = 0; k < kmax; ++k) {
                SelWin nxt = cur;
                const unsigned p1 = p_next;
                if (k + 2 < kmax) p_next = cl[c0 + (k + 2) * nwaves];
                if (k + 1 < kmax) sel_fetch(g, stt, p1, lane, nxt);
                if (sel_eval(ck, stt, cur, lane, spins, lf + c0 + k * nwaves)) pending |= 1ull << k;
                cur = nxt;
            }
        } else {
            for (unsigned k = 0; k < kmax; ++k)
                if (sel_visit_large(g, ck, stt, cl[c0 + k * nwaves], lane, lf + c0 + k * nwaves)) pending |= 1ull << k;
        }
        for (int pass = 1; pass < passes && pending; ++pass) {
            unsigned long long m = pending;
            while (m) {
                const unsigned k = (unsigned)__builtin_ctzll(m);
                m &= m - 1;
                const unsigned p = cl[c0 + k * nwaves];
                bool l;
                if (small) { SelWin W; sel_fetch(g, stt, p, lane, W); l = sel_eval(ck, stt, W, lane, spins, lf + c0 + k * nwaves); }
                else l = sel_visit_large(g, ck, stt, p, lane, lf + c0 + k * nwaves);
                if (!l) pending &= ~(1ull << k);
            }
        }
        if (pending) left = true;
    }
    if (left && lane == 0) hdr[blockIdx.y].undecided[slot] = 1u;     // flag only: a counter would be a contended atomic
}

// gather accepted candidates of the rank prefix as 64-bit keys (score bits << 32 | linear index);
// also clears the rank counters the sort accumulates into (n_acc <= n)
__global__ __launch_bounds__(256) void gather_accepted(const unsigned* __restrict__ ckey, int h, int w,
                                                       GfttHdr* __restrict__ hdr, const uint16_t* __restrict__ state,
                                                       const unsigned* __restrict__ cand, unsigned long long* __restrict__ keys,
                                                       unsigned* __restrict__ rank, int take_all, int seg) {
    const size_t ib = (size_t)blockIdx.y * h * w;
    const unsigned n = hdr[blockIdx.y].seg_end[seg];
    const int lane = threadIdx.x & 63;
    for (unsigned c0 = (blockIdx.x * 256 + threadIdx.x) & ~63u; c0 < n; c0 += gridDim.x * 256) {
        const unsigned c = c0 + lane;
        bool ok = false;
        unsigned p = 0;
        if (c < n) { p = cand[ib + c]; ok = take_all || (state[ib + p] & 3u) == 2u; rank[ib + c] = 0u; }
        const unsigned long long bal = __ballot(ok);
        if (!bal) continue;
        unsigned base = 0;
        const int leader = __ffsll((long long)bal) - 1;
        if (lane == leader) base = atomicAdd(&hdr[blockIdx.y].n_acc, (unsigned)__popcll(bal));
        base = (unsigned)__shfl((int)base, leader);
        if (ok) keys[ib + base + __popcll(bal & ((1ull << lane) - 1))] = ((unsigned long long)ckey[ib + p] << 32) | p;
    }
}

// ---------------------------------------------------------------------------------------------
// descending sort of unique 64-bit keys: normalised bitonic network (every compare-exchange puts
// the larger key at the lower index), so indices >= n act as virtual minimum keys that never move.
struct SortJob { unsigned n; };

__device__ __forceinline__ void cmpx(unsigned long long* s, unsigned i, unsigned p) {   // i < p
    const unsigned long long a = s[i], b = s[p];
    if (a < b) { s[i] = b; s[p] = a; }
}

// mode 0: full sort of each chunk of `len` keys (power of two <= SORT_CH; SORT_CH unless there is a
// single chunk); mode 1: steps j = SORT_CH/2 .. 1 of a larger stage.
// emit != 0 (single chunk only): write the first min(n, max_out) corners as (x,y) floats.
__global__ __launch_bounds__(SORT_THREADS) void sort_local(unsigned long long* __restrict__ keys, const GfttHdr* __restrict__ hdr,
                                                           size_t key_stride, int mode, int emit, int w,
                                                           float* __restrict__ corners, int out_stride, int max_out,
                                                           unsigned len) {
    extern __shared__ unsigned long long sk[];
    const unsigned n = hdr[blockIdx.y].n_acc;
    const unsigned base = blockIdx.x * SORT_CH;
    if (base >= n) return;
    unsigned long long* g = keys + (size_t)blockIdx.y * key_stride;
    for (unsigned e = threadIdx.x; e < len; e += SORT_THREADS) sk[e] = (base + e < n) ? g[base + e] : 0ull;
    __syncthreads();
    if (mode == 0) {
        for (unsigned k = 2; k <= len; k <<= 1) {
            for (unsigned t = threadIdx.x; t < len / 2; t += SORT_THREADS) {       // mirror step
                const unsigned blk = t / (k / 2), o = t - blk * (k / 2);
                const unsigned i = blk * k + o, p = blk * k + (k - 1 - o);
                cmpx(sk, i, p);
            }
            __syncthreads();
            for (unsigned j = k >> 2; j > 0; j >>= 1) {
                for (unsigned t = threadIdx.x; t < len / 2; t += SORT_THREADS) {
                    const unsigned i = ((t / j) * 2 * j) + (t % j);
                    cmpx(sk, i, i + j);
                }
                __syncthreads();
            }
        }
    } else {
        for (unsigned j = len >> 1; j > 0; j >>= 1) {
            for (unsigned t = threadIdx.x; t < len / 2; t += SORT_THREADS) {
                const unsigned i = ((t / j) * 2 * j) + (t % j);
                cmpx(sk, i, i + j);
            }
            __syncthreads();
        }
    }
    for (unsigned e = threadIdx.x; e < len; e += SORT_THREADS)
        if (base + e < n) g[base + e] = sk[e];
    if (emit) {
        const unsigned m = min(n, (unsigned)max_out);
        float2* out = reinterpret_cast<float2*>(corners) + (size_t)blockIdx.y * out_stride;
        for (unsigned e = threadIdx.x; e < m; e += SORT_THREADS) {
            const unsigned p = (unsigned)(sk[e] & 0xffffffffull);
            const unsigned y = p / (unsigned)w;
            out[e] = make_float2((float)(p - y * w), (float)y);
        }
    }
}

// one global step of stage k: mirror (first) or distance j
__global__ __launch_bounds__(256) void sort_global_step(unsigned long long* __restrict__ keys, const GfttHdr* __restrict__ hdr,
                                                        size_t key_stride, unsigned k, unsigned j, int mirror) {
    const unsigned n = hdr[blockIdx.y].n_acc;
    unsigned long long* g = keys + (size_t)blockIdx.y * key_stride;
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    unsigned i, p;
    if (mirror) {
        const unsigned blk = t / (k / 2), o = t - blk * (k / 2);
        i = blk * k + o; p = blk * k + (k - 1 - o);
    } else {
        i = ((t / j) * 2 * j) + (t % j); p = i + j;
    }
    if (p >= n) return;
    const unsigned long long a = g[i], b = g[p];
    if (a < b) { g[i] = b; g[p] = a; }
}

__global__ __launch_bounds__(256) void emit_corners(const unsigned long long* __restrict__ keys, const GfttHdr* __restrict__ hdr,
                                                    size_t key_stride, int w, float* __restrict__ corners, int out_stride,
                                                    int max_out) {
    const unsigned n = hdr[blockIdx.y].n_acc;
    const unsigned m = min(n, (unsigned)max_out);
    const unsigned e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m) return;
    const unsigned p = (unsigned)(keys[(size_t)blockIdx.y * key_stride + e] & 0xffffffffull);
    const unsigned y = p / (unsigned)w;
    reinterpret_cast<float2*>(corners)[(size_t)blockIdx.y * out_stride + e] = make_float2((float)(p - y * w), (float)y);
}

// ---------------------------------------------------------------------------------------------
// rank sort for up to RANK_MAX keys: rank(i) = number of larger keys (keys are unique), counted in
// (256 keys) x (RANK_J keys) tiles over a 2-D grid; the inner operand is wave-uniform, so it comes
// through scalar loads and the loop is one 64-bit compare + add per key.
constexpr unsigned RANK_J = 128, RANK_MAX = 32768;
__global__ __launch_bounds__(256) void rank_count(const unsigned long long* __restrict__ keys, const GfttHdr* __restrict__ hdr,
                                                  size_t key_stride, unsigned* __restrict__ rank, size_t rank_stride) {
    const unsigned n = hdr[blockIdx.z].n_acc;
    if (n > RANK_MAX) return;                                     // the host runs the bitonic network instead
    const unsigned long long* g = keys + (size_t)blockIdx.z * key_stride;
    for (unsigned i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {
        const unsigned i = i0 + threadIdx.x;
        const unsigned long long mine = i < n ? g[i] : ~0ull;
        for (unsigned j0 = blockIdx.y * RANK_J; j0 < n; j0 += gridDim.y * RANK_J) {
            const unsigned jn = min(RANK_J, n - j0);
            const unsigned long long* gj = g + j0;
            unsigned cnt = 0, t = 0;
            for (; t + 16 <= jn; t += 16) {                       // 16 keys = two s_load_dwordx16 per trip
#pragma unroll
                for (unsigned u = 0; u < 16; ++u) cnt += gj[t + u] > mine ? 1u : 0u;
            }
            for (; t < jn; ++t) cnt += gj[t] > mine ? 1u : 0u;
            if (i < n && cnt) atomicAdd(&rank[(size_t)blockIdx.z * rank_stride + i], cnt);
        }
    }
}
__global__ __launch_bounds__(256) void emit_ranked(const unsigned long long* __restrict__ keys, const GfttHdr* __restrict__ hdr,
                                                   size_t key_stride, const unsigned* __restrict__ rank, size_t rank_stride,
                                                   int w, float* __restrict__ corners, int out_stride, int max_out,
                                                   GfttHdr* __restrict__ host_hdr, float* __restrict__ host_corners) {
    // host_hdr / host_corners: pinned host memory mapped into the device.  The last kernel of a call writes what the
    // host needs there itself, so the call ends with a stream synchronisation and no device-to-host copy commands
    // (each of those costs ~10 us of latency behind the kernels).
    if (host_hdr && blockIdx.x == 0 && threadIdx.x < sizeof(GfttHdr) / 4)
        reinterpret_cast<unsigned*>(host_hdr + blockIdx.y)[threadIdx.x] = reinterpret_cast<const unsigned*>(hdr + blockIdx.y)[threadIdx.x];
    const unsigned n = hdr[blockIdx.y].n_acc;
    if (n > RANK_MAX) return;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned rk = rank[(size_t)blockIdx.y * rank_stride + i];
        if (rk >= (unsigned)max_out) continue;
        const unsigned p = (unsigned)(keys[(size_t)blockIdx.y * key_stride + i] & 0xffffffffull);
        const unsigned y = p / (unsigned)w;
        const float2 c = make_float2((float)(p - y * w), (float)y);
        reinterpret_cast<float2*>(corners)[(size_t)blockIdx.y * out_stride + rk] = c;
        if (host_corners) reinterpret_cast<float2*>(host_corners)[rk] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// Last stage of a detector call when the rank prefix is short (the reference's settings: a few thousand candidates):
// ONE block per image replaces gather_accepted + rank_count + emit_ranked.  The candidate list is in bucket order
// (coarse rank), so an order-preserving compaction of the accepted candidates is already sorted except inside runs of
// equal bucket; rank = run start + number of larger keys in the run (keys are unique), exact whatever the run length.
// Thread t owns the candidates [t*m, (t+1)*m) of the prefix; block-wide exclusive scan of the accepted counts; keys and
// buckets of the accepted candidates go to LDS in list order; then the run fix-up and the scatter of (x, y) to rank
// order, into the caller's buffer and (host_corners) into mapped pinned host memory, followed by the header.
constexpr unsigned FIN_MAX = 12288;              // prefix length one block handles (keys 96 KiB + buckets 24 KiB of LDS)
constexpr int FIN_PER = FIN_MAX / 1024;
__global__ __launch_bounds__(1024) void finish_small(int h, int w, GfttHdr* __restrict__ hdr, const unsigned* __restrict__ cand,
                                                     const unsigned* __restrict__ lkey, const uint8_t* __restrict__ lflag,
                                                     int take_all, int seg, float* __restrict__ corners, int out_stride,
                                                     int max_out, GfttHdr* __restrict__ host_hdr, float* __restrict__ host_corners) {
    extern __shared__ unsigned long long fin_keys[];
    uint16_t* fin_bkt = reinterpret_cast<uint16_t*>(fin_keys + FIN_MAX);
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_tot;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int img = blockIdx.x;
    const size_t ib = (size_t)img * h * w;
    GfttHdr* hd = hdr + img;
    const unsigned n = hd->seg_end[seg];
    if (n > FIN_MAX) {                                // uniform: too long for one block, the host runs the split chain
        if (t == 0) hd->fin_state = 2u;
        __syncthreads();
        if (host_hdr && t < sizeof(GfttHdr) / 4) reinterpret_cast<unsigned*>(host_hdr + img)[t] = reinterpret_cast<const unsigned*>(hd)[t];
        return;
    }
    const unsigned m = (n + 1023u) / 1024u;           // candidates per thread (<= FIN_PER)
    const unsigned c0 = (unsigned)t * m;
    BucketMap bm;
    bm.mb = hd->bm_mb; bm.shift = hd->bm_shift;
    unsigned p[FIN_PER], kb[FIN_PER];
    unsigned cnt = 0;
#pragma unroll
    for (int i = 0; i < FIN_PER; ++i) {                // three contiguous, independent reads per candidate
        const unsigned c = c0 + i;
        const bool in = (unsigned)i < m && c < n;
        p[i] = in ? cand[ib + c] : 0xffffffffu;
        kb[i] = in ? lkey[ib + c] : 0u;
        const unsigned fl = (in && !take_all) ? (unsigned)lflag[ib + c] : 1u;
        if (!(in && fl)) p[i] = 0xffffffffu;
        cnt += p[i] != 0xffffffffu ? 1u : 0u;
    }
    unsigned inc = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    unsigned off = 0;
    for (int q = 0; q < wv; ++q) off += s_w[q];
    if (t == 1023) s_tot = off + inc;
    unsigned pos = off + inc - cnt;
#pragma unroll
    for (int i = 0; i < FIN_PER; ++i)
        if (p[i] != 0xffffffffu) {
            fin_keys[pos] = ((unsigned long long)kb[i] << 32) | p[i];
            fin_bkt[pos] = (uint16_t)bucket_of(bm, kb[i]);
            ++pos;
        }
    __syncthreads();
    const unsigned na = s_tot;
    float2* out = reinterpret_cast<float2*>(corners) + (size_t)img * out_stride;
    // ranks first (registers), then the (x, y) pairs go through LDS in rank order, so that the stores to the caller's buffer
    // and to mapped host memory are contiguous (2 000 scattered 8-byte writes over the bus cost ~6 us)
    unsigned rks[FIN_PER];
    float2 cs[FIN_PER];
#pragma unroll
    for (int i = 0; i < FIN_PER; ++i) {
        const unsigned j = (unsigned)t + 1024u * i;
        rks[i] = 0xffffffffu;
        cs[i] = make_float2(0.f, 0.f);
        if (j >= na) continue;
        const unsigned d = fin_bkt[j];
        const unsigned long long key = fin_keys[j];
        unsigned lo = j, hi = j + 1;
        while (lo > 0 && fin_bkt[lo - 1] == d) --lo;
        while (hi < na && fin_bkt[hi] == d) ++hi;
        unsigned rk = lo;
        for (unsigned k = lo; k < hi; ++k) rk += fin_keys[k] > key ? 1u : 0u;
        const unsigned q = (unsigned)(key & 0xffffffffull);
        const unsigned y = q / (unsigned)w;
        rks[i] = rk;
        cs[i] = make_float2((float)(q - y * w), (float)y);
    }
    __syncthreads();                                  // every key has been read: the key array becomes the output stage
    float2* stage = reinterpret_cast<float2*>(fin_keys);
#pragma unroll
    for (int i = 0; i < FIN_PER; ++i) if (rks[i] != 0xffffffffu) stage[rks[i]] = cs[i];
    __syncthreads();
    const unsigned nout = min(na, (unsigned)max(max_out, 0));
    for (unsigned j = t; j < nout; j += 1024) {
        const float2 c = stage[j];
        out[j] = c;
        if (host_corners) reinterpret_cast<float2*>(host_corners)[j] = c;
    }
    if (t == 0) { hd->n_acc = na; hd->fin_state = 1u; }
    __syncthreads();
    if (host_hdr && t < sizeof(GfttHdr) / 4) reinterpret_cast<unsigned*>(host_hdr + img)[t] = reinterpret_cast<const unsigned*>(hd)[t];
}

// ---------------------------------------------------------------------------------------------
// Last stage for LONG rank prefixes (more candidates than finish_small's one block holds: whole lists, 4K frames): the
// same idea over many blocks.  The candidate list is in bucket order, so an order-preserving compaction of the accepted
// candidates is sorted except inside runs of equal bucket, and rank = run start + number of larger keys in the run.
//   fin_count:     accepted candidates per tile of FT list entries                          -> tile counts
//   fin_compact:   tile base = sum of the counts before it; keys (score bits << 32 | pixel) -> keys[], in list order
//   fin_rank_emit: run bounds by walking the neighbours' buckets, rank inside the run, (x, y) to rank order, header out
// O(n) instead of the O(n^2 / tile) key comparisons of rank_count (4K frame: 46 us -> one launch of a few us).
constexpr unsigned FT = 4096;                    // list entries per tile (4 per thread of a 1024-thread block)
__global__ __launch_bounds__(1024) void fin_count(const GfttHdr* __restrict__ hdr, size_t px, const uint8_t* __restrict__ lflag,
                                                  int take_all, int seg, unsigned* __restrict__ tile_cnt) {
    __shared__ unsigned s_w[16];
    const size_t ib = (size_t)blockIdx.y * px;
    const unsigned n = hdr[blockIdx.y].seg_end[seg];
    const unsigned ntile = (n + FT - 1) / FT;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (unsigned tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const unsigned c0 = tile * FT + 4u * t;
        unsigned cnt = 0;
        for (unsigned i = 0; i < 4; ++i) if (c0 + i < n) cnt += take_all ? 1u : (unsigned)(lflag[ib + c0 + i] & 1);
        for (int o = 32; o > 0; o >>= 1) cnt += (unsigned)__shfl_xor((int)cnt, o);
        if (lane == 0) s_w[wv] = cnt;
        __syncthreads();
        if (t == 0) {
            unsigned tot = 0;
            for (int q = 0; q < 16; ++q) tot += s_w[q];
            tile_cnt[ib + tile] = tot;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void fin_compact(GfttHdr* __restrict__ hdr, size_t px, const unsigned* __restrict__ cand,
                                                    const unsigned* __restrict__ lkey, const uint8_t* __restrict__ lflag, int take_all,
                                                    int seg, const unsigned* __restrict__ tile_cnt, unsigned long long* __restrict__ keys,
                                                    uint4* __restrict__ zero_me) {
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_base;
    const size_t ib = (size_t)blockIdx.y * px;
    const unsigned n = hdr[blockIdx.y].seg_end[seg];
    const unsigned ntile = (n + FT - 1) / FT;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    (void)zero_me;
    if (ntile == 0 && blockIdx.x == 0 && t == 0) hdr[blockIdx.y].n_acc = 0u;
    for (unsigned tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        // base of this tile: the counts of all tiles before it
        unsigned part = 0;
        for (unsigned j = t; j < tile; j += 1024) part += tile_cnt[ib + j];
        for (int o = 32; o > 0; o >>= 1) part += (unsigned)__shfl_xor((int)part, o);
        if (lane == 0) s_w[wv] = part;
        __syncthreads();
        if (t == 0) {
            unsigned tot = 0;
            for (int q = 0; q < 16; ++q) tot += s_w[q];
            s_base = tot;
        }
        __syncthreads();
        const unsigned base = s_base;
        __syncthreads();
        // this thread's four entries, in list order
        const unsigned c0 = tile * FT + 4u * t;
        unsigned fl[4], cnt = 0;
#pragma unroll
        for (unsigned i = 0; i < 4; ++i) {
            fl[i] = (c0 + i < n) ? (take_all ? 1u : (unsigned)(lflag[ib + c0 + i] & 1)) : 0u;
            cnt += fl[i];
        }
        unsigned inc = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = (unsigned)__shfl_up((int)inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        unsigned off = 0, tot = 0;
        for (int q = 0; q < 16; ++q) { off += q < wv ? s_w[q] : 0u; tot += s_w[q]; }
        unsigned pos = base + off + inc - cnt;
#pragma unroll
        for (unsigned i = 0; i < 4; ++i)
            if (fl[i]) keys[ib + pos++] = ((unsigned long long)lkey[ib + c0 + i] << 32) | cand[ib + c0 + i];
        if (tile == ntile - 1 && t == 0) hdr[blockIdx.y].n_acc = base + tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void fin_rank_emit(const GfttHdr* __restrict__ hdr, size_t px, const unsigned long long* __restrict__ keys,
                                                     int w, float* __restrict__ corners, int out_stride, int max_out,
                                                     GfttHdr* __restrict__ host_hdr, float* __restrict__ host_corners) {
    if (host_hdr && blockIdx.x == 0 && threadIdx.x < sizeof(GfttHdr) / 4)
        reinterpret_cast<unsigned*>(host_hdr + blockIdx.y)[threadIdx.x] = reinterpret_cast<const unsigned*>(hdr + blockIdx.y)[threadIdx.x];
    const GfttHdr* hd = hdr + blockIdx.y;
    const unsigned na = hd->n_acc;
    if (na > RANK_MAX) return;                                   // the host runs the bitonic network over keys[] instead
    BucketMap bm;
    bm.mb = hd->bm_mb; bm.shift = hd->bm_shift;
    const unsigned long long* g = keys + (size_t)blockIdx.y * px;
    for (unsigned j = blockIdx.x * 256 + threadIdx.x; j < na; j += gridDim.x * 256) {
        const unsigned long long key = g[j];
        const unsigned d = bucket_of(bm, (unsigned)(key >> 32));
        unsigned rk = j;                                         // = run start + larger keys in the run, counted outwards from j
        for (unsigned k = j; k > 0; --k) {                       // earlier members of the run: those with a smaller key move behind
            const unsigned long long o = g[k - 1];
            if (bucket_of(bm, (unsigned)(o >> 32)) != d) break;
            if (o < key) --rk;
        }
        for (unsigned k = j + 1; k < na; ++k) {                  // later members of the run: those with a larger key move in front
            const unsigned long long o = g[k];
            if (bucket_of(bm, (unsigned)(o >> 32)) != d) break;
            if (o > key) ++rk;
        }
        if (rk >= (unsigned)max_out) continue;
        const unsigned p = (unsigned)(key & 0xffffffffull);
        const unsigned y = p / (unsigned)w;
        const float2 c = make_float2((float)(p - y * w), (float)y);
        reinterpret_cast<float2*>(corners)[(size_t)blockIdx.y * out_stride + rk] = c;
        if (host_corners) reinterpret_cast<float2*>(host_corners)[rk] = c;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Static workspace of vofe_gftt_noalloc (include/vofe.h): the detector's whole workspace for frames of up to
// VOFE_NOALLOC_MAX_PIXELS pixels as ONE file-scope device array, so that entry point never allocates.  Every GPU the
// library runs on holds its own copy (at its own address) until the process ends.  `scratch` is only a host-side
// handle for the symbol: host code never passes it anywhere but to hipGetSymbolAddress; kernels, memsets and copies
// take the per-device address kept in g_scratch_ptr.
constexpr size_t NOALLOC_MAX_PX = VOFE_NOALLOC_MAX_PIXELS;
constexpr size_t NOALLOC_HDR_BYTES = 256;
static_assert(sizeof(GfttHdr) <= NOALLOC_HDR_BYTES, "one header");
constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_HDR_BYTES + 2 * (size_t)N_COPY * N_BUCKET * sizeof(unsigned) +
                                         NOALLOC_MAX_PX * GFTT_BYTES_PER_PX + 1024 + 32;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

constexpr int VOFE_MAX_DEVICES = 64;
static void* g_scratch_ptr[VOFE_MAX_DEVICES];        // per device: where THAT device's copy of `scratch` lives
static bool g_fin_lds_set[VOFE_MAX_DEVICES];         // per device: finish_small's dynamic-LDS attribute has been set

namespace {

int ensure_ws(vofe_ctx* ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return VOFE_OK;
    if (ctx->ws) { hipStreamSynchronize(ctx->stream); hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    ctx->gftt_clean_batch = 0;                     // a new allocation starts dirty
    hipError_t e = hipMalloc(&ctx->ws, bytes);
    if (e != hipSuccess) return vofe_hip_fail(e, "hipMalloc(gftt workspace)");
    ctx->ws_bytes = bytes;
    return VOFE_OK;
}
int ensure_pinned(vofe_ctx* ctx, size_t ints) {
    if (ctx->h_pinned_ints >= ints) return VOFE_OK;
    if (ctx->h_pinned) { hipStreamSynchronize(ctx->stream); hipHostFree(ctx->h_pinned); }
    ctx->h_pinned = nullptr; ctx->d_pinned = nullptr; ctx->h_pinned_ints = 0;
    hipError_t e = hipHostMalloc((void**)&ctx->h_pinned, ints * sizeof(int), hipHostMallocMapped);
    if (e != hipSuccess) return vofe_hip_fail(e, "hipHostMalloc");
    e = hipHostGetDevicePointer((void**)&ctx->d_pinned, ctx->h_pinned, 0);
    if (e != hipSuccess) return vofe_hip_fail(e, "hipHostGetDevicePointer");
    ctx->h_pinned_ints = ints;
    return VOFE_OK;
}

}  // namespace

int vofe_gftt_setup(vofe_ctx* ctx) {
    // once per device, with that device current (vofe_create has made it so): the device's own address of `scratch`
    if (ctx->device < 0 || ctx->device >= VOFE_MAX_DEVICES) { vofe_set_error("vofe_create: device index %d beyond %d", ctx->device, VOFE_MAX_DEVICES); return VOFE_E_UNSUPPORTED; }
    if (!g_scratch_ptr[ctx->device]) {
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return vofe_hip_fail(e, "hipGetSymbolAddress(scratch)");
        g_scratch_ptr[ctx->device] = ptr;
    }
    // finish_small launches with 120 KiB of dynamic LDS: the attribute that allows it belongs to the (current) device's copy
    // of the kernel, so it is set here, once per device, not behind a process-wide flag at the first detector call
    if (!g_fin_lds_set[ctx->device]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(finish_small), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((size_t)FIN_MAX * (8 + 2)));
        if (e != hipSuccess) return vofe_hip_fail(e, "hipFuncSetAttribute(finish_small, MaxDynamicSharedMemorySize): the one-block last stage needs 120 KiB of LDS");
        g_fin_lds_set[ctx->device] = true;
    }
    // the pinned, device-mapped header every detector call reads its counts from: one image's worth now, so that the
    // no-allocation entry point finds it in place (batched calls and host corner lists grow it on demand)
    return ensure_pinned(ctx, sizeof(GfttHdr) / 4);
}

namespace {
// cv::getSobelKernels for one axis (integer taps): binomial smoothing (order 0) or first derivative (order 1); a 1-tap
// derivative request becomes the 3-tap [-1 0 1] as in OpenCV
int sobel_taps(int ksize, int order, int* k) {
    if (ksize == 1 && order > 0) ksize = 3;
    int buf[9] = {0};
    if (ksize == 1) buf[0] = 1;
    else if (ksize == 3) {
        if (order == 0) { buf[0] = 1; buf[1] = 2; buf[2] = 1; }
        else { buf[0] = -1; buf[1] = 0; buf[2] = 1; }
    } else {
        buf[0] = 1;
        for (int i = 0; i < ksize - order - 1; ++i) {
            int oldval = buf[0];
            for (int j = 1; j <= ksize; ++j) { const int nv = buf[j] + buf[j - 1]; buf[j - 1] = oldval; oldval = nv; }
        }
        for (int i = 0; i < order; ++i) {
            int oldval = -buf[0];
            for (int j = 1; j <= ksize; ++j) { const int nv = buf[j - 1] - buf[j]; buf[j - 1] = oldval; oldval = nv; }
        }
    }
    for (int j = 0; j < ksize; ++j) k[j] = buf[j];
    return ksize;
}

int launch_corner_score_generic(vofe_ctx* ctx, const uint8_t* d_img, int batch, int h, int w, int pitch, size_t img_stride,
                                int block, int aperture, int harris, float k, float* d_score, unsigned* d_hdr,
                                const uint8_t* d_mask) {
    ScoreGeom g = {};
    g.bs = block; g.an = block / 2;
    int dt[8], st[8];
    g.ld = sobel_taps(aperture, 1, dt); g.ls = sobel_taps(aperture, 0, st);
    g.hd = g.ld / 2; g.hs = g.ls / 2; g.hm = g.hd > g.hs ? g.hd : g.hs;
    const double scale = 1.0 / ((double)(1 << (aperture - 1)) * block * 255.0);        // cornerEigenValsVecs, 8-bit input
    for (int j = 0; j < g.ld; ++j) g.fd[j] = (float)dt[j];
    for (int j = 0; j < g.ls; ++j) g.fs[j] = (float)((double)st[j] * scale);            // Mat * double, stored as f32
    // the largest tile whose LDS plan fits 64 KiB (less the kernel's few static words); a tile side stays above
    // blockSize / 2, so that an image longer than a tile is also longer than the box's overhang and reflected indices stay
    // inside the tile's own region
    static const int tiles[3][2] = {{64, 16}, {32, 16}, {16, 16}};
    const size_t lds_max = 64 * 1024 - 256;
    size_t lds = 0;
    for (int t = 0; t < 3; ++t) {
        g.tx = tiles[t][0]; g.ty = tiles[t][1];
        lds = score_generic_lds(g);
        if (lds <= lds_max) break;
    }
    if (lds > lds_max) { vofe_set_error("corner_score: blockSize %d / gradientSize %d needs %zu bytes of LDS", block, aperture, lds); return VOFE_E_UNSUPPORTED; }
    dim3 grid((w + g.tx - 1) / g.tx, (h + g.ty - 1) / g.ty, batch);
    if (grid.y > 65535) { vofe_set_error("corner_score: image too large"); return VOFE_E_UNSUPPORTED; }
    VOFE_TIMED(ctx, "corner_score_generic");
#define SCORE_GEOM_LAUNCH(B, K) hipLaunchKernelGGL((corner_score_generic<B, K>), grid, dim3(256), lds, ctx->stream, d_img, h, w, pitch, \
                                                   img_stride, g, harris, k, d_score, reinterpret_cast<GfttHdr*>(d_hdr), d_mask)
    // the geometries a cv2 user is most likely to pass get unrolled instances (OpenCV's own optical-flow example uses
    // blockSize 7); VOFE_CORNER_SCORE=generic keeps the run-time instance for all of them (tests compare the two)
    const bool rt = ctx->knobs.score_generic;
    if (!rt && block == 5 && aperture == 3) SCORE_GEOM_LAUNCH(5, 3);
    else if (!rt && block == 7 && aperture == 3) SCORE_GEOM_LAUNCH(7, 3);
    else if (!rt && block == 3 && aperture == 5) SCORE_GEOM_LAUNCH(3, 5);
    else if (!rt && block == 5 && aperture == 5) SCORE_GEOM_LAUNCH(5, 5);
    else SCORE_GEOM_LAUNCH(0, 0);
#undef SCORE_GEOM_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "corner_score_generic launch");
    return VOFE_OK;
}
}  // namespace

int vofe_launch_corner_score(vofe_ctx* ctx, const uint8_t* d_img, int batch, int h, int w, int pitch,
                             size_t img_stride, int harris, float k, float* d_score,
                             unsigned* d_hdr, const uint8_t* d_mask, int block_size, int aperture) {
    if (batch > 65535) { vofe_set_error("corner_score: batch too large"); return VOFE_E_UNSUPPORTED; }
    if (block_size != 3 || aperture != 3 || ctx->knobs.score_generic)       // VOFE_CORNER_SCORE=generic: also block 3 / aperture 3 (tests)
        return launch_corner_score_generic(ctx, d_img, batch, h, w, pitch, img_stride, block_size, aperture, harris, k, d_score,
                                           d_hdr, d_mask);
    if (w >= 8 && h >= 4 && !ctx->knobs.score_tile) {            // VOFE_CORNER_SCORE=tile forces the LDS-tile kernel (tests)
        // strips of 8 rows; 4 rows while that leaves SIMDs without a wave (small frames are latency bound)
        long long waves8 = (long long)((w + 255) / 256) * ((h + 7) / 8) * batch;
        if (const int rs = ctx->knobs.score_rs)                  // VOFE_SCORE_RS: 2, 4, 8 or 16 rows per strip (tests, tuning)
            waves8 = rs >= 16 ? 3072 : rs >= 8 ? 1024 : rs >= 4 ? 512 : 0;
        if (waves8 >= 3072) {
            // more 8-row waves than the chip holds at once (3 per SIMD at this register count): 16-row strips
            // halve the wave count and trim the 4 halo rows per strip
            dim3 grid((w + 255) / 256, (h + 63) / 64, batch);
            if (grid.y > 65535) { vofe_set_error("corner_score: image too large"); return VOFE_E_UNSUPPORTED; }
            VOFE_TIMED(ctx, "corner_score_strip<16>");
            hipLaunchKernelGGL(corner_score_strip<16>, grid, dim3(256), 0, ctx->stream, d_img, h, w, pitch, img_stride, harris, k,
                               d_score, reinterpret_cast<GfttHdr*>(d_hdr), d_mask);
        } else if (waves8 >= 1024) {
            dim3 grid((w + 255) / 256, (h + 31) / 32, batch);
            if (grid.y > 65535) { vofe_set_error("corner_score: image too large"); return VOFE_E_UNSUPPORTED; }
            VOFE_TIMED(ctx, "corner_score_strip<8>");
            hipLaunchKernelGGL(corner_score_strip<8>, grid, dim3(256), 0, ctx->stream, d_img, h, w, pitch, img_stride, harris, k,
                               d_score, reinterpret_cast<GfttHdr*>(d_hdr), d_mask);
        } else if (waves8 >= 512) {
            dim3 grid((w + 255) / 256, (h + 15) / 16, batch);
            VOFE_TIMED(ctx, "corner_score_strip<4>");
            hipLaunchKernelGGL(corner_score_strip<4>, grid, dim3(256), 0, ctx->stream, d_img, h, w, pitch, img_stride, harris, k,
                               d_score, reinterpret_cast<GfttHdr*>(d_hdr), d_mask);
        } else {
            // a single small frame: 2-row strips double the wave count again (the kernel is pure latency there)
            dim3 grid((w + 255) / 256, (h + 7) / 8, batch);
            VOFE_TIMED(ctx, "corner_score_strip<2>");
            hipLaunchKernelGGL(corner_score_strip<2>, grid, dim3(256), 0, ctx->stream, d_img, h, w, pitch, img_stride, harris, k,
                               d_score, reinterpret_cast<GfttHdr*>(d_hdr), d_mask);
        }
    } else {
        dim3 grid((w + TX - 1) / TX, (h + TY - 1) / TY, batch);
        if (grid.y > 65535) { vofe_set_error("corner_score: image too large"); return VOFE_E_UNSUPPORTED; }
        VOFE_TIMED(ctx, "corner_score_f32");
        hipLaunchKernelGGL(corner_score_f32, grid, dim3(256), 0, ctx->stream, d_img, h, w, pitch, img_stride, harris, k,
                           d_score, reinterpret_cast<GfttHdr*>(d_hdr), d_mask);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "corner_score launch");
    return VOFE_OK;
}

int vofe_gftt_impl(vofe_ctx* ctx, const uint8_t* d_img, int batch, int h, int w, int pitch,
                   size_t img_stride, int max_corners, double quality, double min_distance,
                   const uint8_t* d_mask, int harris, double k, float* d_corners_xy, int capacity,
                   int* h_n_out, float* h_corners_xy, bool noalloc, int block_size, int aperture) {
    hipStream_t st = ctx->stream;
    const size_t px = (size_t)h * w;
    if (px >= (1ull << 31)) { vofe_set_error("vofe_gftt: image too large"); return VOFE_E_UNSUPPORTED; }
    // workspace: hdr | bucket histograms | bucket cursors | keys u64 | score f32 (ranks after the gather) | bucket-ordered
    // candidates u32 | ckey u32 | lkey u32 (score bits by list position) | state u16 | lflag u8 (accepted, by list position)
    const size_t hdr_bytes = ((sizeof(GfttHdr) * batch + 255) / 256) * 256;
    const size_t hist_bytes = (size_t)batch * N_COPY * N_BUCKET * sizeof(unsigned);
    const size_t tot = hdr_bytes + 2 * hist_bytes + batch * px * GFTT_BYTES_PER_PX + 1024 + 16;
    int rc = VOFE_OK;
    uint8_t* base = nullptr;
    if (noalloc) {
        // the device's static scratch array instead of the context's workspace: nothing is allocated or freed
        if (batch != 1 || h_corners_xy || px > NOALLOC_MAX_PX || tot > NOALLOC_SCRATCH_BYTES) {
            vofe_set_error("vofe_gftt_noalloc: %dx%d = %zu pixels exceeds the static workspace (VOFE_NOALLOC_MAX_PIXELS = %zu)",
                           w, h, px, NOALLOC_MAX_PX);
            return VOFE_E_UNSUPPORTED;
        }
        base = static_cast<uint8_t*>(g_scratch_ptr[ctx->device]);
        if (!base || ctx->h_pinned_ints < sizeof(GfttHdr) / 4) { vofe_set_error("vofe_gftt_noalloc: context was not set up"); return VOFE_E_HIP; }
    } else {
        rc = ensure_ws(ctx, tot);
        if (rc) return rc;
        base = static_cast<uint8_t*>(ctx->ws);
    }
    // pinned staging: headers, then (single image, host output requested) the corner list itself, so that
    // corners and count arrive with the one synchronisation of the call
    const size_t hdr_ints = (sizeof(GfttHdr) / 4) * (size_t)batch;
    size_t stage_floats = 0;
    if (h_corners_xy && batch == 1) {
        long long m = capacity;
        if (max_corners > 0 && max_corners < m) m = max_corners;
        stage_floats = (size_t)std::max<long long>(m, 0) * 2;
        if (stage_floats > 2 * (size_t)RANK_MAX) stage_floats = 0;       // unbounded lists are fetched after the count is known
    }
    if (!noalloc) {
        rc = ensure_pinned(ctx, hdr_ints + stage_floats);
        if (rc) return rc;
    }
    GfttHdr* hdr = reinterpret_cast<GfttHdr*>(base);
    unsigned* hist = reinterpret_cast<unsigned*>(base + hdr_bytes);
    unsigned* cursor = reinterpret_cast<unsigned*>(base + hdr_bytes + hist_bytes);
    uint8_t* maps = base + hdr_bytes + 2 * hist_bytes;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(maps);
    float* score = reinterpret_cast<float*>(maps + batch * px * 8);
    unsigned* cand = reinterpret_cast<unsigned*>(maps + batch * px * 12);
    unsigned* ckey = reinterpret_cast<unsigned*>(maps + batch * px * 16);
    unsigned* lkey = reinterpret_cast<unsigned*>(maps + batch * px * 20);
    uint16_t* state = reinterpret_cast<uint16_t*>(maps + ((batch * px * 24 + 15) & ~(size_t)15));   // 16-byte aligned
    uint8_t* lflag = reinterpret_cast<uint8_t*>(state) + batch * px * 2;

    const bool suppress = min_distance >= 1.0;
    // schedule of the selection (measured, profiles/r01_host_paths.txt; any values give the same corners, see
    // tools/knob_matrix.sh): one launch of 4096 waves, two sweeps, a blocked candidate polls its blockers up to 64 times
    const VofeKnobs& kn = ctx->knobs;
    // (a batch shares the chip: with 1024 blocks per image most of a short prefix's waves find no candidate and the launch is
    // bound by wave start-up; 64 images: 196 -> 150 us with 64 blocks per image, tools/gftt_batch_bench.py)
    const int sel_blocks = std::max(kn.sel_blocks_set ? kn.sel_blocks : std::max(64, kn.sel_blocks / std::max(batch, 1)), 1);
    const int sel_launches = std::min(std::max(kn.sel_launches, 1), SELECT_ROUNDS);
    const int sel_passes = std::max(kn.sel_passes, 1);
    const int sel_spins = std::max(kn.sel_spins, 0);
    const int sel_prefix_env = kn.sel_prefix;                              // > 0: fixed first-prefix factor (tests, tuning)
    const int sel_full_env = kn.sel_full;                                  // > 0: fixed whole-list ratio
    // How many of the best-ranked candidates hold maxCorners accepted ones?  On densely textured frames the accepted
    // corners saturate at about one per 1.7 * minDistance^2 pixels (capacity C; measured 5366 on 1241x376 at 7 px, 2868
    // on 800x600 at 10 px, 1849 on 640x480 at 10 px) and the best K candidates hold about C * (1 - exp(-K / 1.5 C)) of
    // them (640x480 at 10 px: 915 in the best 2001, 1245 in 4003; 1241x376 at 10 px: 1064 in 2000).  The first prefix
    // is that K for 1.1 * maxCorners, plus 5 % (growing to 19 % at 0.75 C, where the model is optimistic), and at least 2 * maxCorners; a request beyond 0.75 C, or a list
    // shorter than twice the prefix, starts with the whole list: a prefix that turns out too short costs a second
    // selection pass (640x480, 1000 corners at 10 px: 113 us with the former fixed factor of 2, 102 us with the whole
    // list; 800x600, 3000 corners at 10 px needs the whole list: 122 us against 163 us via a prefix).
    double prefix_factor = sel_prefix_env > 0 ? (double)sel_prefix_env : 2.0;
    int sel_full = sel_full_env > 0 ? sel_full_env : 2;
    if (max_corners > 0 && suppress) {
        const double capacity = (double)h * (double)w / (1.7 * min_distance * min_distance);
        const double r = (double)max_corners / capacity;
        if (sel_full_env <= 0 && r >= 0.75) sel_full = 1 << 20;
        if (sel_prefix_env <= 0 && r < 0.75) prefix_factor = std::max(2.0, -1.5 / r * std::log(1.0 - 1.1 * r) * (1.05 + 0.4 * std::max(0.0, r - 0.4)));
    }
    // first rank prefix: without suppression the best maxCorners candidates are the answer; with it,
    // a few times maxCorners candidates usually hold maxCorners accepted ones (grown x2 otherwise)
    unsigned seg_base = 0;
    if (max_corners > 0) {
        const unsigned long long b0 = suppress ? (unsigned long long)std::ceil((double)max_corners * prefix_factor) : (unsigned long long)max_corners;
        seg_base = (unsigned)std::min<unsigned long long>(b0, 1u << 24);
    }

    // header + histogram start a call at zero: bucket_scan leaves them so, a memset is needed only for a fresh or
    // re-laid-out workspace (the layout depends on the batch) or after a call that did not run to its end
    // (the static scratch array is shared by every context on the device and keeps the last call's contents: always cleared)
    if (noalloc || ctx->gftt_clean_batch != batch || ctx->gftt_clean_ws != ctx->ws)
        VOFE_HIP(hipMemsetAsync(base, 0, hdr_bytes + hist_bytes, st));
    if (!noalloc) ctx->gftt_clean_batch = 0;
    rc = vofe_launch_corner_score(ctx, d_img, batch, h, w, pitch, img_stride, harris, (float)k, score,
                                  reinterpret_cast<unsigned*>(hdr), d_mask, block_size, aperture);
    if (rc) return rc;
    // VOFE_GFTT_CHAIN=split keeps the longer chain of round 1 (tests: both chains must give the same corners)
    const bool merged = kn.gftt_chain != 's';     // VOFE_GFTT_CHAIN=split: the last stage as three launches (tests: same corners)
    {
        // rows per thread by launch size: the chip holds ~8 k waves of this kernel at once
        const long long waves1 = (long long)((w + 63) / 64) * ((h + 3) / 4) * 4 * batch;          // at one row per thread
        int rows = kn.nms_rows > 0 ? kn.nms_rows : (waves1 <= 8192 ? 1 : waves1 <= 16384 ? 2 : waves1 <= 32768 ? 4 : 8);
        VOFE_TIMED(ctx, "nms_map");
#define VOFE_NMS(R) hipLaunchKernelGGL(nms_map<R>, dim3((w + 63) / 64, (h + 4 * R - 1) / (4 * R), batch), dim3(256), 0, st, \
                                       score, h, w, quality, hdr, state, ckey, hist, d_mask)
        if (rows >= 8) VOFE_NMS(8); else if (rows >= 4) VOFE_NMS(4); else if (rows >= 2) VOFE_NMS(2); else VOFE_NMS(1);
#undef VOFE_NMS
    }
    { VOFE_TIMED(ctx, "bucket_scan");
    hipLaunchKernelGGL(bucket_scan, dim3(batch), dim3(1024), 0, st, hist, cursor, hdr, seg_base, (unsigned)sel_full, quality); }
    { VOFE_TIMED(ctx, "bucket_scatter");
    hipLaunchKernelGGL(bucket_scatter, dim3((unsigned)std::min<size_t>((px + 2047) / 2048, 2048), batch), dim3(256), 0, st,
                       state, ckey, h, w, cursor, cand, lkey, lflag, hdr, 0); }
    bool scattered_rest = false;
    int r = 0;
    float md2 = 0.f;
    if (suppress) {
        r = (int)ceil(min_distance) - 1;
        md2 = (float)(min_distance * min_distance);
        // (float)(dx*dx+dy*dy) >= md2 must equal the double test d2 < minDistance^2: use the smallest
        // float >= minDistance^2 so that integer d2 compares identically
        if ((double)md2 < min_distance * min_distance) md2 = nextafterf(md2, INFINITY);
    }
    GfttHdr* hh = reinterpret_cast<GfttHdr*>(ctx->h_pinned);
    unsigned* rank = reinterpret_cast<unsigned*>(score);         // the score map is dead after nms_compact
    int max_out = capacity;
    if (max_corners > 0 && max_corners < max_out) max_out = max_corners;
    const bool use_rank = !kn.gftt_bitonic;                       // VOFE_GFTT_SORT=bitonic forces the network (tests)
    int seg = 0;
    for (int pass = 0;; ++pass) {
        if (seg > 0 && !scattered_rest) {
            // the prefix is being widened beyond the buckets the first scatter took: the other candidates join the list now
            VOFE_TIMED(ctx, "bucket_scatter");
            hipLaunchKernelGGL(bucket_scatter, dim3((unsigned)std::min<size_t>((px + 2047) / 2048, 2048), batch), dim3(256), 0, st,
                               state, ckey, h, w, cursor, cand, lkey, lflag, hdr, 1);
            scattered_rest = true;
        }
        if (pass > 0) {
            // clear the round flags and the accepted count, keep the rest
            for (int b = 0; b < batch; ++b) {
                VOFE_HIP(hipMemsetAsync(&hdr[b].n_acc, 0, sizeof(unsigned), st));
                VOFE_HIP(hipMemsetAsync(hdr[b].undecided, 0, sizeof(unsigned) * SELECT_ROUNDS, st));
            }
        }
        if (suppress)
            for (int round = 0; round < sel_launches; ++round) {
                VOFE_TIMED(ctx, "select_round");
                hipLaunchKernelGGL(select_round, dim3(sel_blocks, batch), dim3(256), 0, st, ckey, h, w, r, md2, hdr, state,
                                   cand, lflag, seg, sel_passes, sel_spins, round);
            }
        // Last stage.  A short rank prefix (the reference's settings) is finished by ONE block per image; its length is
        // known on the device only, so the host goes by the prefix it asked for (or, for the whole list, by the candidate
        // count of this context's previous call: consecutive frames of a camera are alike), and the kernel reports
        // back when the prefix was longer than one block handles (fin_state 2: the split chain runs after all).
        const unsigned long long last = ctx->gftt_last_ncand;
        // (bucket_scan takes the whole list when it is at most sel_full prefixes long, or when no prefix was asked for)
        const bool whole = !seg_base || seg >= N_SEG - 1 || (last && last <= (unsigned long long)sel_full * seg_base);
        unsigned long long ub_fin = whole ? (last ? last : px) : ((unsigned long long)seg_base << seg);
        if (!whole && last) ub_fin = std::min(ub_fin, last);
        const bool fin = merged && use_rank && max_out > 0 && ub_fin + ub_fin / 4 <= FIN_MAX;
        bool split = !fin;
        if (fin) {
            const size_t fin_lds = (size_t)FIN_MAX * (8 + 2);                    // (its launch attribute: vofe_gftt_setup, once per device)
            { VOFE_TIMED(ctx, "finish_small");
            hipLaunchKernelGGL(finish_small, dim3(batch), dim3(1024), fin_lds, st, h, w, hdr, cand, lkey, lflag, suppress ? 0 : 1, seg,
                               d_corners_xy, capacity, max_out, reinterpret_cast<GfttHdr*>(ctx->d_pinned),
                               stage_floats ? reinterpret_cast<float*>(ctx->d_pinned + hdr_ints) : nullptr); }
            VOFE_HIP(hipStreamSynchronize(st));
            for (int b = 0; b < batch; ++b) split = split || hh[b].fin_state == 2u;
        }
        // long prefixes: order-preserving multi-block compaction + in-run ranks (VOFE_GFTT_CHAIN=split / rank keep round 1's
        // gather / rank_count / emit chain for the tests)
        const bool ordered = split && use_rank && max_out > 0 && kn.gftt_chain == 0;
        if (ordered) {
            unsigned long long ub = whole ? (last ? std::max<unsigned long long>(last, 1) : px) : ((unsigned long long)seg_base << seg);
            ub = std::min<unsigned long long>(ub + ub / 4 + FT, px);
            const unsigned gt = (unsigned)std::min<unsigned long long>((ub + FT - 1) / FT, 512);
            { VOFE_TIMED(ctx, "fin_count");
            hipLaunchKernelGGL(fin_count, dim3(gt, batch), dim3(1024), 0, st, hdr, px, lflag, suppress ? 0 : 1, seg, rank); }
            { VOFE_TIMED(ctx, "fin_compact");
            hipLaunchKernelGGL(fin_compact, dim3(gt, batch), dim3(1024), 0, st, hdr, px, cand, lkey, lflag, suppress ? 0 : 1, seg, rank, keys,
                               (uint4*)nullptr); }
            const unsigned ge = (unsigned)std::min<unsigned long long>((std::min<unsigned long long>(ub, RANK_MAX) + 255) / 256, 1024);
            { VOFE_TIMED(ctx, "fin_rank_emit");
            hipLaunchKernelGGL(fin_rank_emit, dim3(std::max(ge, 1u), batch), dim3(256), 0, st, hdr, px, keys, w, d_corners_xy, capacity, max_out,
                               reinterpret_cast<GfttHdr*>(ctx->d_pinned), stage_floats ? reinterpret_cast<float*>(ctx->d_pinned + hdr_ints) : nullptr); }
            VOFE_HIP(hipStreamSynchronize(st));
        } else if (split) {
        { VOFE_TIMED(ctx, "gather_accepted");
        hipLaunchKernelGGL(gather_accepted, dim3(256, batch), dim3(256), 0, st, ckey, h, w, hdr, state, cand, keys,
                           rank, suppress ? 0 : 1, seg); }
        if (use_rank && max_out > 0) {
            // the sort runs speculatively behind the gather (sized by the device-side count), so the
            // common case needs one host round trip per call
            unsigned long long ub = seg_base ? ((unsigned long long)seg_base << seg) : px;
            ub = std::min<unsigned long long>(std::min<unsigned long long>(ub + ub / 4, px), RANK_MAX);
            const unsigned gi = (unsigned)((ub + 255) / 256), gj = (unsigned)((ub + RANK_J - 1) / RANK_J);
            { VOFE_TIMED(ctx, "rank_count");
            hipLaunchKernelGGL(rank_count, dim3(gi, gj, batch), dim3(256), 0, st, keys, hdr, px, rank, px); }
            VOFE_TIMED(ctx, "emit_ranked");
            // the emit kernel delivers header and (single image) corner list straight into pinned host memory
            hipLaunchKernelGGL(emit_ranked, dim3(gi, batch), dim3(256), 0, st, keys, hdr, px, rank, px, w, d_corners_xy,
                               capacity, max_out, reinterpret_cast<GfttHdr*>(ctx->d_pinned),
                               stage_floats ? reinterpret_cast<float*>(ctx->d_pinned + hdr_ints) : nullptr);
        } else {
            VOFE_HIP(hipMemcpyAsync(hh, hdr, sizeof(GfttHdr) * batch, hipMemcpyDeviceToHost, st));
        }
        VOFE_HIP(hipStreamSynchronize(st));
        }
        unsigned ncand_max = 0;
        for (int b = 0; b < batch; ++b) ncand_max = std::max(ncand_max, hh[b].n_cand);
        ctx->gftt_last_ncand = ncand_max;
        bool converged = true, enough = true;
        for (int b = 0; b < batch; ++b) {
            if (suppress && hh[b].undecided[sel_launches - 1]) converged = false;
            if (hh[b].seg_end[seg] < hh[b].n_cand && (max_corners <= 0 || hh[b].n_acc < (unsigned)max_corners)) enough = false;
        }
        if (kn.gftt_debug)
            fprintf(stderr, "gftt pass %d seg %d: n_cand %u seg_end %u n_acc %u undecided %u converged %d enough %d\n", pass, seg,
                    hh[0].n_cand, hh[0].seg_end[seg], hh[0].n_acc, hh[0].undecided[sel_launches - 1], (int)converged, (int)enough);
        if (converged && enough) break;
        if (converged && seg < N_SEG - 1) {
            // every image's prefix is final: widen it to where the acceptance rate seen so far (which
            // only falls with rank) would reach maxCorners, with 50 % head room
            int next = seg + 1;
            for (int b = 0; b < batch; ++b) {
                if (hh[b].seg_end[seg] >= hh[b].n_cand || hh[b].n_acc >= (unsigned)max_corners) continue;
                const double est = 1.5 * (double)hh[b].seg_end[seg] * (double)max_corners / (double)std::max(hh[b].n_acc, 1u);
                int kx = seg + 1;
                if (est > 0.5 * (double)hh[b].n_cand) kx = N_SEG - 1;         // most of the list anyway: take all
                while (kx < N_SEG - 1 && (double)hh[b].seg_end[kx] < est) ++kx;
                next = std::max(next, kx);
            }
            seg = next;
        }
        if (pass > 4096) { vofe_set_error("vofe_gftt: selection did not converge"); return VOFE_E_HIP; }
    }
    if (!noalloc) {
        ctx->gftt_clean_batch = batch;             // bucket_scan left max_key and the histograms of this layout at zero
        ctx->gftt_clean_ws = ctx->ws;
    }
    unsigned nmax = 0;
    for (int b = 0; b < batch; ++b) {
        const unsigned n = hh[b].n_acc;
        nmax = n > nmax ? n : nmax;
        h_n_out[b] = (int)((max_corners > 0 && n > (unsigned)max_corners) ? (unsigned)max_corners : n);
    }
    if (nmax == 0 || max_out <= 0) return VOFE_OK;
    const unsigned chunks = (nmax + SORT_CH - 1) / SORT_CH;
    const size_t lds = (size_t)SORT_CH * 8;
    if (use_rank && nmax <= RANK_MAX) {
        // already sorted and emitted by the speculative rank sort
        if (h_corners_xy && stage_floats) {
            memcpy(h_corners_xy, reinterpret_cast<int*>(ctx->h_pinned) + hdr_ints,
                   (size_t)std::min(h_n_out[0], max_out) * 2 * sizeof(float));     // only min(count, capacity) pairs were staged
            return VOFE_OK;
        }
    } else if (chunks == 1) {
        unsigned len = 64;
        while (len < nmax) len <<= 1;
        hipLaunchKernelGGL(sort_local, dim3(1, batch), dim3(SORT_THREADS), (size_t)len * 8, st, keys, hdr, px, 0, 1, w,
                           d_corners_xy, capacity, max_out, len);
    } else {
        hipLaunchKernelGGL(sort_local, dim3(chunks, batch), dim3(SORT_THREADS), lds, st, keys, hdr, px, 0, 0, w,
                           d_corners_xy, capacity, max_out, (unsigned)SORT_CH);
        unsigned npow = SORT_CH;
        while (npow < nmax) npow <<= 1;
        const unsigned pair_blocks = (npow / 2 + 255) / 256;
        for (unsigned kk = 2 * SORT_CH; kk <= npow; kk <<= 1) {
            hipLaunchKernelGGL(sort_global_step, dim3(pair_blocks, batch), dim3(256), 0, st, keys, hdr, px, kk, 0u, 1);
            for (unsigned j = kk >> 2; j >= (unsigned)SORT_CH; j >>= 1)
                hipLaunchKernelGGL(sort_global_step, dim3(pair_blocks, batch), dim3(256), 0, st, keys, hdr, px, kk, j, 0);
            hipLaunchKernelGGL(sort_local, dim3(chunks, batch), dim3(SORT_THREADS), lds, st, keys, hdr, px, 1, 0, w,
                               d_corners_xy, capacity, max_out, (unsigned)SORT_CH);
        }
        if (max_out > 0)
            hipLaunchKernelGGL(emit_corners, dim3((max_out + 255) / 256, batch), dim3(256), 0, st, keys, hdr, px, w,
                               d_corners_xy, capacity, max_out);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "gftt launch");
    if (h_corners_xy && batch == 1) {
        const size_t nb = (size_t)std::min(h_n_out[0], max_out) * 2 * sizeof(float);
        VOFE_HIP(hipMemcpyAsync(h_corners_xy, d_corners_xy, nb, hipMemcpyDeviceToHost, st));
        VOFE_HIP(hipStreamSynchronize(st));
    }
    return VOFE_OK;
}
