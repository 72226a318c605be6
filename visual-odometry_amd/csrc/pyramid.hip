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
#include <cstdlib>
#include <cstring>

#include "vofe_internal.h"

namespace {

struct U4 { unsigned a, b, c, d; } __attribute__((packed, aligned(4)));
struct U2 { unsigned a, b; } __attribute__((packed, aligned(4)));

// 16 pixels x0 .. x0+15 of an image row under REFLECT_101 (row = pixel 0 of the row, w = image width).  Groups
// inside the image are one (unaligned) 16-byte load; groups entirely left or right of it are the byte-reversed
// 16 bytes around the mirror column; only groups that straddle an image edge (or images narrower than the band)
// gather byte by byte.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ uint4 load16(const uint8_t* p) {
    const u32x4 q = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_unaligned*>(
        (const __attribute__((address_space(1))) uint8_t*)p);
    return make_uint4(q.x, q.y, q.z, q.w);
}
__device__ __forceinline__ uint4 reversed16(uint4 a) {
    return make_uint4(__builtin_bswap32(a.w), __builtin_bswap32(a.z), __builtin_bswap32(a.y), __builtin_bswap32(a.x));
}
__device__ __forceinline__ uint4 mirror16(const uint8_t* row, int x0, int w) {
    if (x0 >= 0 && x0 + 16 <= w) return load16(row + x0);
    if (x0 + 15 < 0 && -x0 <= w - 1) return reversed16(load16(row - x0 - 15));
    if (x0 >= w && 2 * (w - 1) - x0 - 15 >= 0) return reversed16(load16(row + 2 * (w - 1) - x0 - 15));
    uint8_t b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = row[vofe_reflect101(x0 + i, w)];
    uint4 v;
    __builtin_memcpy(&v, b, 16);
    return v;
}

// ---------------------------------------------------------------------------------------------
// level 0: copy the caller's image into the padded plane and fill the REFLECT_101 border.
// One thread = 16 consecutive bytes of one padded row; threads are numbered over (row, group) so that
// narrow frames (84 groups per KITTI row) still fill every wavefront.
__global__ __launch_bounds__(256) void pad_copy_level0(const uint8_t* __restrict__ src, int spitch,
                                                       size_t img_stride, VofeLevel L, int slot0,
                                                       int win_h, int gx0, int ngx, int items) {
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= items) return;
    const int row = it / ngx, g = it - row * ngx;
    const int y = row - win_h;
    const int x0 = (g + gx0) * 16;
    const uint8_t* s = src + (size_t)blockIdx.y * img_stride;
    uint8_t* d = L.I + (long long)(slot0 + (int)blockIdx.y) * L.slot_stride + (long long)y * L.pitch + x0;
    const int sy = vofe_reflect101(y, L.h);
    *reinterpret_cast<uint4*>(d) = mirror16(s + (size_t)sy * spitch, x0, L.w);
}

// The same for frames at least 16 * max(nl, nr) + 17 pixels wide (every frame but toy sizes), without divergence: a
// padded row is nl mirrored groups left of the image, ni image groups of which the last is shifted left to end
// at the last pixel (it overlaps its neighbour with equal bytes), and nr mirrored groups starting right after
// the last pixel.  Every thread is then one 16-byte load, an optional byte reversal and one 16-byte store
// (unaligned for the shifted and the right-hand groups); no group straddles an image edge.
__device__ __forceinline__ void store16(uint8_t* p, uint4 v) {
    u32x4 q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
    *reinterpret_cast<__attribute__((address_space(1))) u32x4_unaligned*>((__attribute__((address_space(1))) uint8_t*)p) = q;
}
__global__ __launch_bounds__(256) void pad_copy_level0_wide(const uint8_t* __restrict__ src, int spitch,
                                                            size_t img_stride, VofeLevel L, int slot0,
                                                            int win_h, int nl, int ni, int nr, int items) {
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= items) return;
    const int ngx = nl + ni + nr;
    const int row = it / ngx, g = it - row * ngx;
    const int y = row - win_h;
    const int sy = vofe_reflect101(y, L.h);
    const uint8_t* srow = src + (size_t)blockIdx.y * img_stride + (size_t)sy * spitch;
    uint8_t* drow = L.I + (long long)(slot0 + (int)blockIdx.y) * L.slot_stride + (long long)y * L.pitch;
    const int jr = g - nl - ni;                                    // >= 0: right-hand group
    const bool rev = g < nl || jr >= 0;
    const int dx = g < nl ? -16 * (nl - g) : (jr >= 0 ? L.w + 16 * jr : min(16 * (g - nl), L.w - 16));
    const int sx = g < nl ? -dx - 15 : (jr >= 0 ? L.w - 17 - 16 * jr : dx);   // mirrored: out[i] = src[sx + 15 - i]
    uint4 v = load16(srow + sx);
    const uint4 r = reversed16(v);
    if (rev) v = r;
    store16(drow + dx, v);       // (two rows per thread for more loads in flight measured the same; a non-temporal store gains 5 % here and loses it in the pyrDown that reads the plane next)
}

// ---------------------------------------------------------------------------------------------
// pyrDown: interior of the next level plus its two REFLECT_101 border columns on either side.
// One thread = 8 consecutive output pixels of TWO output rows (7 source rows instead of 10).  A source row
// arrives as six dwords covering columns 2*X0-4 .. 2*X0+19; the ten 4-byte windows starting at every second
// column come from v_alignbyte, and one output's row term  w_k * (p[-2] + 4 p[-1] + 6 p[0] + 4 p[1] + p[2])
// is two v_dot4_u32_u8 accumulations: window j against w_k*(1,4,6,4) and window j+2 against w_k*(1,0,0,0)
// (w_k*6 <= 36 fits a byte), so the whole 5x5 kernel is accumulated in integers exactly as SURVEY A.3 states it.
// Source rows are addressed through REFLECT_101 on the row index, so the source needs valid border COLUMNS only
// (two per side, on its interior rows): level 0 has its full border from pad_copy_level0, and every level
// written here gets those columns from the threads that own columns 1, 2, w-3 and w-2.  The full border
// bands of the new levels (LK and Scharr need them) are filled afterwards by one pyramid_borders launch.
// BANDS: the thread also writes the REFLECT_101 border bands (win_w columns left and right, win_h rows above and below,
// corners included) whose SOURCE pixels it owns: every border pixel mirrors exactly one interior pixel (column x in 1..win_w
// goes to -x, column x in w-1-win_w..w-2 to 2(w-1)-x, rows likewise), so a full group is stored once more byte-reversed at
// the mirrored position and a row in a mirror range once more to the mirrored row -- no pyramid_borders launch behind the
// chain.  Needs win_w <= w - 1 and win_h <= h - 1 on the written level (one reflection); the host checks.  A reversed
// group may reach up to 7 bytes past the band: the plane's padding (>= win_w + 20 columns) absorbs that.
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef u32x2v u32x2v_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ void store8u(uint8_t* p, unsigned a, unsigned b) {
    u32x2v q; q.x = a; q.y = b;
    *reinterpret_cast<__attribute__((address_space(1))) u32x2v_unaligned*>((__attribute__((address_space(1))) uint8_t*)p) = q;
}

// 5x5 kernel of two output rows x eight outputs from seven source row segments (24 bytes each, starting 2 bytes left of the
// first output's first tap): acc[o][j] += sum_k w_k * (row term of output j in source row k), exact integers.
__device__ __forceinline__ void pyr_down_acc(const U4 (&ra)[7], const U2 (&rb)[7], unsigned (&acc)[2][8]) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const U4 a = ra[k];
        const U2 b = rb[k];
        const unsigned wv[6] = {a.a, a.b, a.c, a.d, b.a, b.b};
        unsigned win[10];                          // win[m] = bytes 2+2m .. 5+2m of the row segment
#pragma unroll
        for (int m = 0; m < 10; ++m)
            win[m] = (m & 1) ? wv[(m + 1) / 2] : __builtin_amdgcn_alignbyte(wv[m / 2 + 1], wv[m / 2], 2);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int t = k - 2 * o;               // tap index of this source row in output row o
            if (t < 0 || t > 4) continue;
            const unsigned wk = (t == 0 || t == 4) ? 1u : (t == 2 ? 6u : 4u);
            const unsigned W = wk * 0x04060401u, E = wk;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[o][j] = __builtin_amdgcn_udot4(win[j], W, acc[o][j], false);
                acc[o][j] = __builtin_amdgcn_udot4(win[j + 2], E, acc[o][j], false);
            }
        }
    }
}

// Eight pixels X0 .. X0+7 (X0 a multiple of 8) of row y of a level -> its plane: the row itself and, with BANDS, every border
// position that mirrors one of these pixels (see above); without BANDS the two border columns either side that the next
// pyrDown reads.
template <bool BANDS>
__device__ __forceinline__ void pyr_store_row(uint8_t* plane, const VofeLevel& Dl, int X0, int y, const unsigned (&out)[8],
                                              int win_w, int win_h) {
    const int w = Dl.w, h = Dl.h;
    const bool full = X0 + 8 <= w;
    const bool edge = X0 == 0 || X0 + 8 > w - 3;
    const bool src_l = BANDS && X0 <= win_w;                   // holds a column of 1..win_w
    const bool src_r = BANDS && X0 + 7 >= w - 1 - win_w;       // holds a column of w-1-win_w..w-2
    const unsigned qx = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24);
    const unsigned qy = out[4] | (out[5] << 8) | (out[6] << 16) | (out[7] << 24);
    // destination rows: y itself and, with BANDS, its mirror images above the first / below the last image row
    const int ny = 1 + ((BANDS && y >= 1 && y <= win_h) ? 1 : 0) + ((BANDS && y <= h - 2 && y >= h - 1 - win_h) ? 1 : 0);
    for (int c = 0; c < ny; ++c) {
        int yd = y;
        if (c > 0) yd = (c == 1 && y >= 1 && y <= win_h) ? -y : 2 * (h - 1) - y;
        uint8_t* drow = plane + (long long)yd * Dl.pitch;
        if (full) *reinterpret_cast<uint2*>(drow + X0) = make_uint2(qx, qy);
        if (BANDS) {
            if (full) {
                const unsigned rx = __builtin_bswap32(qy), rw = __builtin_bswap32(qx);      // the group byte-reversed
                if (src_l) store8u(drow - X0 - 7, rx, rw);
                if (src_r) store8u(drow + 2 * (w - 1) - X0 - 7, rx, rw);
            } else if (X0 < w) {
                // the ragged last group of a row (one thread per row): byte by byte
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int x = X0 + j;
                    if (x < w) {
                        drow[x] = (uint8_t)out[j];
                        if (x >= 1 && x <= win_w) drow[-x] = (uint8_t)out[j];
                        if (x <= w - 2 && x >= w - 1 - win_w) drow[2 * (w - 1) - x] = (uint8_t)out[j];
                    }
                }
            }
            continue;
        }
        if (edge) {
            // Nearly every wavefront holds a thread at a row end (a 621-pixel row is 78 groups), so this path is as hot as
            // the body: keep it to a handful of instructions.  Left end: columns 1 and 2 go to -1 and -2.  Right end:
            // columns w-2 and w-3 go to w and w+1 (each is picked out of the owning group's eight outputs by a select
            // chain).  A ragged last group stores its w - X0 valid bytes as 4 + 2 + 1.
            if (X0 == 0) {
                if (1 < w) drow[-1] = (uint8_t)out[1];
                if (2 < w) drow[-2] = (uint8_t)out[2];
            }
            const int j2 = w - 2 - X0, j3 = w - 3 - X0;
            unsigned v2 = out[0], v3 = out[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) { v2 = j2 == j ? out[j] : v2; v3 = j3 == j ? out[j] : v3; }
            if ((unsigned)j2 < 8u) drow[w] = (uint8_t)v2;
            if ((unsigned)j3 < 8u) drow[w + 1] = (uint8_t)v3;
            if (!full && X0 < w) {
                const int nv = w - X0;                                                // 1..7
                uint8_t* dp = drow + X0;
                int o0 = 0;
                if (nv & 4) { *reinterpret_cast<unsigned*>(dp) = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24); o0 = 4; }
                if (nv & 2) {
                    const unsigned lo = o0 ? out[4] : out[0], hi = o0 ? out[5] : out[1];
                    *reinterpret_cast<unsigned short*>(dp + o0) = (unsigned short)(lo | (hi << 8));
                }
                if (nv & 1) {
                    const int jb = nv - 1;
                    unsigned vb = out[0];
#pragma unroll
                    for (int j = 1; j < 7; ++j) vb = jb == j ? out[j] : vb;
                    dp[jb] = (uint8_t)vb;
                }
            }
        }
    }
}

template <bool BANDS>
__device__ __forceinline__ void pyr_down_body(const VofeLevel& S, const VofeLevel& Dl, int slot0, int gpr, int items, int win_w, int win_h,
                                              int it, int img) {
    if (it >= items) return;
    const int yp = it / gpr, g = it - yp * gpr;
    const int X0 = g * 8, Y0 = yp * 2;
    const uint8_t* sp = S.I + (long long)(slot0 + img) * S.slot_stride + 2 * X0 - 4;
    unsigned acc[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 128;
    // all fourteen loads first: with the REFLECT_101 row index (a loop with branches) between them the loads of the seven
    // rows went out one row per memory round trip and the kernel was bound by that latency (22.5 us per 128 frames)
    int ry[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) ry[k] = vofe_reflect101(2 * Y0 - 2 + k, S.h);
    U4 ra[7];
    U2 rb[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const uint8_t* r = sp + (long long)ry[k] * S.pitch;
        ra[k] = *reinterpret_cast<const U4*>(r);
        rb[k] = *reinterpret_cast<const U2*>(r + 16);
    }
    pyr_down_acc(ra, rb, acc);
    uint8_t* plane = Dl.I + (long long)(slot0 + img) * Dl.slot_stride;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int y = Y0 + o;
        if (y >= Dl.h) break;
        unsigned out[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = acc[o][j] >> 8;
        pyr_store_row<BANDS>(plane, Dl, X0, y, out, win_w, win_h);
    }
}

template <bool BANDS>
__global__ __launch_bounds__(256) void pyr_down_u8(VofeLevel S, VofeLevel Dl, int slot0, int gpr, int items, int win_w, int win_h) {
    pyr_down_body<BANDS>(S, Dl, slot0, gpr, items, win_w, win_h, (int)blockIdx.x * 256 + (int)threadIdx.x, (int)blockIdx.y);
}

__device__ __forceinline__ unsigned byte_of(const unsigned* w, int i) { return (w[i >> 2] >> ((i & 3) * 8)) & 0xffu; }

// ---------------------------------------------------------------------------------------------
// Scharr derivative of every level in one launch.  One thread = 8 pixels of one row; a block is
// 64 groups x 4 rows.  The source border is valid (REFLECT_101), the destination border stays 0.
struct ScharrPlan {
    int blk_start[VOFE_MAX_LEVELS + 1];
    int blk_x[VOFE_MAX_LEVELS];
    int levels;
};

// Eight derivative pixels x0 .. x0+7 of a row from the sixteen bytes x0-4 .. x0+11 of the rows above (a), at (bq) and below (c);
// d = the row's derivative plane at x0, w = image width.
__device__ __forceinline__ void scharr_emit(const unsigned (&a)[4], const unsigned (&bq)[4], const unsigned (&c)[4], int* d, int x0, int w) {
    int t0[10], t1[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int pa = (int)byte_of(a, i + 3), pb = (int)byte_of(bq, i + 3), pc = (int)byte_of(c, i + 3);
        t0[i] = 3 * (pa + pc) + 10 * pb;
        t1[i] = pc - pa;
    }
    unsigned o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ix = t0[i + 2] - t0[i];
        const int iy = 3 * (t1[i] + t1[i + 2]) + 10 * t1[i + 1];
        o[i] = ((unsigned)ix & 0xffffu) | ((unsigned)iy << 16);
    }
    // the last group of a row may run past w: those columns lie in the zero border, keep them 0
    if (x0 + 8 <= w) {
        // non-temporal: the derivative planes are written once and read back sparsely (windows around keypoints) by a
        // later launch; keeping them out of the caches' way is worth 1 % of the headline (48.99 k -> 49.51 k frames/s,
        // two runs each) because the tracker launch of the other stream lives on L2 / Infinity Cache hits
        typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
        nt_u32x4 q0 = {o[0], o[1], o[2], o[3]}, q1 = {o[4], o[5], o[6], o[7]};
        __builtin_nontemporal_store(q0, reinterpret_cast<nt_u32x4*>(d));
        __builtin_nontemporal_store(q1, reinterpret_cast<nt_u32x4*>(d + 4));
    } else {
        // ragged last group of a row: its w - x0 valid pixels as 4 + 2 + 1 (a third of the wavefronts of a 1241-pixel row
        // pass through here, so eight predicated single stores were ~10 % of the kernel's instructions)
        const int nv = w - x0;                             // 1..7
        int o0 = 0;
        if (nv & 4) { *reinterpret_cast<uint4*>(d) = make_uint4(o[0], o[1], o[2], o[3]); o0 = 4; }
        if (nv & 2) { *reinterpret_cast<uint2*>(d + o0) = o0 ? make_uint2(o[4], o[5]) : make_uint2(o[0], o[1]); }
        if (nv & 1) {
            const int jb = nv - 1;
            unsigned vb = o[0];
#pragma unroll
            for (int j = 1; j < 7; ++j) vb = jb == j ? o[j] : vb;
            d[jb] = (int)vb;
        }
    }
}

__device__ __forceinline__ void scharr_body(const VofeLevel& L, int slot0, int bx, int by, int img) {
    const int x0 = (bx * 64 + (threadIdx.x & 63)) * 8;
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x0 >= L.w || y >= L.h) return;
    const long long so = (long long)(slot0 + img) * L.slot_stride;
    const uint8_t* r1 = L.I + so + (long long)y * L.pitch + x0;
    const uint8_t* r0 = r1 - L.pitch;
    const uint8_t* r2 = r1 + L.pitch;
    unsigned a[4], bq[4], c[4];
    U4 va = *reinterpret_cast<const U4*>(r0 - 4), vb = *reinterpret_cast<const U4*>(r1 - 4), vc = *reinterpret_cast<const U4*>(r2 - 4);
    a[0] = va.a; a[1] = va.b; a[2] = va.c; a[3] = va.d;
    bq[0] = vb.a; bq[1] = vb.b; bq[2] = vb.c; bq[3] = vb.d;
    c[0] = vc.a; c[1] = vc.b; c[2] = vc.c; c[3] = vc.d;
    scharr_emit(a, bq, c, L.D + so + (long long)y * L.pitch + x0, x0, L.w);
}

__global__ __launch_bounds__(256) void scharr_deriv_s16x2(const VofeLevel* __restrict__ lv, ScharrPlan plan, int slot0) {
    int level = 0;
    while (level + 1 < plan.levels && (int)blockIdx.x >= plan.blk_start[level + 1]) ++level;
    const VofeLevel L = lv[level];
    const int b = (int)blockIdx.x - plan.blk_start[level];
    scharr_body(L, slot0, b % plan.blk_x[level], b / plan.blk_x[level], (int)blockIdx.y);
}

// One step of the level pipeline.  The pyrDown that produces level k (with its border bands) depends on level k-1 alone, and
// so does every Scharr block of the levels below k: they share a launch -- blocks [0, pd_blocks) are pyrDown items, the rest
// Scharr blocks of up to three level segments.  The upper-level pyrDowns are launch-bound by themselves (9-15 us for 1-4 MB
// of traffic); the host spreads the Scharr blocks of levels 0 .. top-1 over the pyrDown launches so that every launch of
// the chain is bandwidth-bound, and only the top level's Scharr pass is left for a launch of its own.
constexpr int STEP_SEGS = 8;
struct StepPlan {
    int nseg;
    int level[STEP_SEGS], blk_x[STEP_SEGS], first[STEP_SEGS];   // Scharr segment: level, blocks per row of blocks, first block (row-major)
    int start[STEP_SEGS + 1];                                   // cumulative Scharr block counts of the segments (start[0] = 0)
};
// The grid is one-dimensional: the pyrDown blocks of ALL images first, then the Scharr blocks image by image.  (With the
// images on grid.y the pyrDown blocks of the last images were dispatched at the end of the launch, and a pyrDown block's
// fourteen-load latency chain then ran after the bandwidth-bound part was over: the launch took the sum, not the maximum.)
__global__ __launch_bounds__(256) void pyr_step(const VofeLevel* __restrict__ lv, VofeLevel S, VofeLevel Dl, int slot0, int gpr, int items,
                                                int win_w, int win_h, int pd_blocks, int count, StepPlan plan) {
    // (the pyrDown's two level descriptors are kernel arguments: read out of the table in memory they cost the combined kernel
    // 132 instead of 80 VGPRs, i.e. 3 instead of 6 wavefronts per SIMD for its Scharr blocks too)
    const int npd = pd_blocks * count;
    if ((int)blockIdx.x < npd) {
        const int img = (int)blockIdx.x / pd_blocks, pb = (int)blockIdx.x - img * pd_blocks;
        pyr_down_body<true>(S, Dl, slot0, gpr, items, win_w, win_h, pb * 256 + (int)threadIdx.x, img);
    } else {
        const int nsch = plan.start[plan.nseg];
        const int q = (int)blockIdx.x - npd;
        const int img = q / nsch, b = q - img * nsch;
        int seg = 0;
        while (seg + 1 < plan.nseg && b >= plan.start[seg + 1]) ++seg;
        const int i = b - plan.start[seg] + plan.first[seg];
        const VofeLevel L = lv[plan.level[seg]];
        scharr_body(L, slot0, i % plan.blk_x[seg], i / plan.blk_x[seg], img);
    }
}

__device__ __forceinline__ int refl1(int c, int n) {      // REFLECT_101 where one reflection is enough (small overshoot)
    if (n == 1) return 0;
    return c < 0 ? -c : (c >= n ? 2 * n - 2 - c : c);
}

// The tail of the level pipeline: the last one or two levels (their interiors, border bands and derivative planes) from
// level `base` in ONE launch, together with the Scharr blocks that are still open.  The upper pyrDowns are 1-4 MB each: as
// launches of their own they cost a dependent-launch latency each and run at 6-16 % of the memory rate.  Here a block owns a
// tw x th tile of the top level and the matching twice larger tile of the level below it; it computes the regions those tiles
// depend on (halo recompute, ~1.3x) from level `base` in global memory into LDS with the same v_dot4 row arithmetic as
// pyr_down_body, the second level from the first out of LDS, and then writes its own tiles and their Scharr rows from LDS.
// Region cells outside the image are never computed from the source's border (pyrDown of a REFLECT_101 extension is not the
// REFLECT_101 extension of the pyrDown when the source width is even): they are filled by reflection inside the region, columns
// first, then whole rows, before anything reads them.
// Layout of a region row in LDS: cell i (level coordinate lox + i) at byte cell0 + i; cell0 of the lower region is 2, so that
// the 24-byte segments the second pyrDown reads (from cell 16g - 2) are 16-byte aligned; the rows the output pass reads are
// whole dwords around byte cell0 + (X0 - 4 - lox), shifted into place by v_alignbyte.
struct TailPlan {
    int D, base;                       // levels produced (1 or 2) from level `base`
    int tw, th, halo;                  // top-level tile and the halo of its region
    int tiles_x, tiles;                // tiles per row of tiles, tiles per image
    int n[2], m[2], pitch[2], off[2], cell0[2];    // regions: [0] = level base+1, [D-1] = top level
};
constexpr int TAIL_LDS = 8192;

// it / d for 0 <= it < 2^20 and small d, without the integer-division sequence (inv = 1.0f / d)
__device__ __forceinline__ int div_small(int it, float inv) { return (int)(((float)it + 0.5f) * inv); }

__device__ __forceinline__ void tail_fix_region(uint8_t* R, int n, int m, int pitch, int cell0, int lox, int loy, int w, int h) {
    // (block-uniform: interior tiles have nothing to fill and skip both passes and their barriers)
    if (lox >= 0 && loy >= 0 && lox + n <= w && loy + m <= h) return;
    // columns of the in-image rows
    const int nl = max(0, min(n, -lox)), r0 = max(0, w - lox), nr = max(0, n - r0), nc = nl + nr;
    if (nc > 0) {
        const float inv = 1.0f / (float)nc;
        for (int it = (int)threadIdx.x; it < m * nc; it += 256) {
            const int j = div_small(it, inv), c = it - j * nc;
            const int i = c < nl ? c : r0 + (c - nl);
            const int y = loy + j;
            if (y < 0 || y >= h) continue;
            const int si = vofe_reflect101(lox + i, w) - lox;
            if (si < 0 || si >= n) continue;                       // (not a cell anything reads: its mirror would be in the region)
            uint8_t* row = R + j * pitch + cell0;
            row[i] = row[si];
        }
        __syncthreads();
    }
    // rows above and below the image, whole (columns included)
    const int na = max(0, min(m, -loy)), b0 = max(0, h - loy), nb = max(0, m - b0), dwr = pitch >> 2;
    if (na + nb > 0) {
        const float inv = 1.0f / (float)dwr;
        for (int it = (int)threadIdx.x; it < (na + nb) * dwr; it += 256) {
            const int q = div_small(it, inv), e = it - q * dwr;
            const int j = q < na ? q : b0 + (q - na);
            const int sj = vofe_reflect101(loy + j, h) - loy;
            if (sj < 0 || sj >= m) continue;
            reinterpret_cast<unsigned*>(R + j * pitch)[e] = reinterpret_cast<const unsigned*>(R + sj * pitch)[e];
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void pyr_tail_tile(const VofeLevel& S, const VofeLevel& La, const VofeLevel& Lb, int slot, int win_w, int win_h,
                                              const TailPlan& P, int tile, uint8_t* R) {
    const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const int T = P.D - 1;
    int lox[2], loy[2];
    lox[T] = tx * P.tw - P.halo;
    loy[T] = ty * P.th - P.halo;
    if (T == 1) { lox[0] = 2 * lox[1] - 2; loy[0] = 2 * loy[1] - 2; }
    // ---- region 0 from level `base` in global memory
    {
        const int gpr = (P.n[0] + 7) >> 3, prs = (P.m[0] + 1) >> 1;
        const float inv = 1.0f / (float)gpr;
        uint8_t* R0 = R + P.off[0];
        const uint8_t* sbase = S.I + (long long)slot * S.slot_stride;
        for (int it = (int)threadIdx.x; it < gpr * prs; it += 256) {
            const int jp = div_small(it, inv), g = it - jp * gpr;
            const int X0 = lox[0] + 8 * g, Y0 = loy[0] + 2 * jp;                    // both even
            if (X0 >= La.w || X0 + 8 <= 0 || Y0 >= La.h || Y0 + 2 <= 0) continue;   // no pixel of it lies in the image
            const uint8_t* sp = sbase + 2 * X0 - 4;
            unsigned acc[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 128;
            // source rows -2 .. S.h + 3 at most (0 <= Y0 < La.h): one reflection (S.h >= 2 La.h - 1 >= 7: La is taller than the window)
            int ry[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) ry[k] = refl1(2 * Y0 - 2 + k, S.h);
            U4 ra[7];
            U2 rb[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const uint8_t* r = sp + (long long)ry[k] * S.pitch;
                ra[k] = *reinterpret_cast<const U4*>(r);
                rb[k] = *reinterpret_cast<const U2*>(r + 16);
            }
            pyr_down_acc(ra, rb, acc);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int j = 2 * jp + o;
                if (j >= P.m[0]) break;
                unsigned short* d = reinterpret_cast<unsigned short*>(R0 + j * P.pitch[0] + P.cell0[0] + 8 * g);     // cell0 is even
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q] = (unsigned short)((acc[o][2 * q] >> 8) | ((acc[o][2 * q + 1] >> 8) << 8));
            }
        }
    }
    __syncthreads();
    tail_fix_region(R + P.off[0], P.n[0], P.m[0], P.pitch[0], P.cell0[0], lox[0], loy[0], La.w, La.h);
    // ---- region 1 (the top level) from region 0
    if (T == 1) {
        const int gpr = (P.n[1] + 7) >> 3, prs = P.m[1] >> 1;                       // m[1] is even
        const float inv = 1.0f / (float)gpr;
        const uint8_t* R0 = R + P.off[0];
        uint8_t* R1 = R + P.off[1];
        for (int it = (int)threadIdx.x; it < gpr * prs; it += 256) {
            const int jp = div_small(it, inv), g = it - jp * gpr;
            const int X0 = lox[1] + 8 * g, Y0 = loy[1] + 2 * jp;
            if (X0 >= Lb.w || X0 + 8 <= 0 || Y0 >= Lb.h || Y0 + 2 <= 0) continue;
            unsigned acc[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 128;
            U4 ra[7];
            U2 rb[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const uint8_t* r = R0 + (4 * jp + k) * P.pitch[0] + 16 * g;         // cell 16g - 2 of region row 2 * (2 jp) + k
                const uint4 v = *reinterpret_cast<const uint4*>(r);
                const uint2 u = *reinterpret_cast<const uint2*>(r + 16);
                ra[k].a = v.x; ra[k].b = v.y; ra[k].c = v.z; ra[k].d = v.w;
                rb[k].a = u.x; rb[k].b = u.y;
            }
            pyr_down_acc(ra, rb, acc);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                uint8_t* d = R1 + (2 * jp + o) * P.pitch[1] + P.cell0[1] + 8 * g;
#pragma unroll
                for (int q = 0; q < 8; ++q) d[q] = (uint8_t)(acc[o][q] >> 8);
            }
        }
        __syncthreads();
        tail_fix_region(R + P.off[1], P.n[1], P.m[1], P.pitch[1], P.cell0[1], lox[1], loy[1], Lb.w, Lb.h);
    }
    // ---- own tiles of both levels: plane rows (+ border bands) and derivative rows, eight pixels per thread
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r > T) break;
        const VofeLevel& L = r == 0 ? La : Lb;
        const int d = T - r;
        const int ox0 = (tx * P.tw) << d, oy0 = (ty * P.th) << d;
        const int ox1 = min(ox0 + (P.tw << d), L.w), oy1 = min(oy0 + (P.th << d), L.h);
        if (ox0 >= ox1 || oy0 >= oy1) continue;
        const int gpr = (ox1 - ox0 + 7) >> 3, rows = oy1 - oy0;
        const uint8_t* Rr = R + P.off[r];
        const int pitch = P.pitch[r];
        const long long so = (long long)slot * L.slot_stride;
        uint8_t* plane = L.I + so;
        const float inv = 1.0f / (float)gpr;
        for (int it = (int)threadIdx.x; it < gpr * rows; it += 256) {
            const int jr = div_small(it, inv), g = it - jr * gpr;
            const int X0 = ox0 + 8 * g, y = oy0 + jr;
            const int B = P.cell0[r] + (X0 - 4 - lox[r]);                           // byte of pixel X0 - 4 in a region row (>= 0)
            const int ph = B & 3;
            const uint8_t* r1 = Rr + (y - loy[r]) * pitch + (B - ph);
            unsigned a[4], bq[4], c[4];
            {
                const unsigned* p0 = reinterpret_cast<const unsigned*>(r1 - pitch);
                const unsigned* p1 = reinterpret_cast<const unsigned*>(r1);
                const unsigned* p2 = reinterpret_cast<const unsigned*>(r1 + pitch);
                unsigned la[4], lb[4], lc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { la[q] = p0[q]; lb[q] = p1[q]; lc[q] = p2[q]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = __builtin_amdgcn_alignbyte(q < 3 ? la[q + 1] : 0u, la[q], ph);
                    bq[q] = __builtin_amdgcn_alignbyte(q < 3 ? lb[q + 1] : 0u, lb[q], ph);
                    c[q] = __builtin_amdgcn_alignbyte(q < 3 ? lc[q + 1] : 0u, lc[q], ph);
                }
            }
            unsigned out[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = byte_of(bq, 4 + j);
            pyr_store_row<true>(plane, L, X0, y, out, win_w, win_h);
            scharr_emit(a, bq, c, L.D + so + (long long)y * L.pitch + X0, X0, L.w);
        }
    }
}

__global__ __launch_bounds__(256) void pyr_tail(const VofeLevel* __restrict__ lv, VofeLevel S, VofeLevel La, VofeLevel Lb, int slot0,
                                                int win_w, int win_h, TailPlan P, int count, StepPlan plan) {
    __shared__ __attribute__((aligned(16))) uint8_t R[TAIL_LDS];
    const int ntile = P.tiles * count;
    // (tiles first: spreading them over the grid between the riders -- odd stride, so that they still land on all 8 XCDs --
    // made the launch slower, 52 -> 54-65 us for 128 KITTI frames; profiles/r03_pyramid_chain.txt)
    if ((int)blockIdx.x < ntile) {
        const int img = (int)blockIdx.x / P.tiles, tile = (int)blockIdx.x - img * P.tiles;
        pyr_tail_tile(S, La, Lb, slot0 + img, win_w, win_h, P, tile, R);
    } else {
        const int nsch = plan.start[plan.nseg];
        const int q = (int)blockIdx.x - ntile;
        const int img = q / nsch, b = q - img * nsch;
        int seg = 0;
        while (seg + 1 < plan.nseg && b >= plan.start[seg + 1]) ++seg;
        const int i = b - plan.start[seg] + plan.first[seg];
        const VofeLevel L = lv[plan.level[seg]];
        scharr_body(L, slot0, i % plan.blk_x[seg], i / plan.blk_x[seg], img);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused pyramid: ONE launch produces the interiors of up to FUSE_MAX_LEVELS levels and their Scharr
// planes.  A block owns a tw x th tile of the TOP level and the matching 2^d-times larger tiles of
// the finer levels; it loads the level-0 region those tiles depend on into LDS and runs the pyrDown
// chain there (halo recompute instead of a kernel boundary per level: a dependent launch costs
// ~5 us here, the recompute ~1.7x of the arithmetic).  Region at level k-1 = 2*region(k) + 3.
// Out-of-image coordinates are never materialised: taps go through REFLECT_101 on the source level.
// The REFLECT_101 border bands of the planes are filled by pyramid_borders afterwards.
constexpr int FUSE_MAX_LEVELS = 5;
struct FusePlan {
    int levels, tw, th, halo;
    int n[FUSE_MAX_LEVELS], m[FUSE_MAX_LEVELS], off[FUSE_MAX_LEVELS];
};
typedef unsigned u32_unaligned __attribute__((aligned(1)));

// band_w / band_h > 0: the block also writes the REFLECT_101 border bands (band_w columns left and right, band_h rows above
// and below, corners included) whose SOURCE pixels it owns -- every border pixel mirrors exactly one interior pixel, so the
// owners' scattered copies cover the bands without a second launch.  Needs band_w <= w - 1 and band_h <= h - 1 on every level
// (one reflection); the host falls back to pyramid_borders otherwise.
__global__ __launch_bounds__(1024) void pyramid_fused(const uint8_t* __restrict__ src, int spitch, size_t img_stride,
                                                     const VofeLevel* __restrict__ lv, FusePlan P, int slot0,
                                                     int with_deriv, int band_w, int band_h) {
    extern __shared__ uint8_t R[];
    const int T = P.levels - 1;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, nty = (int)blockDim.x >> 5;
    int lox[FUSE_MAX_LEVELS], loy[FUSE_MAX_LEVELS];
    lox[T] = (int)blockIdx.x * P.tw - P.halo;
    loy[T] = (int)blockIdx.y * P.th - P.halo;
#pragma unroll
    for (int k = FUSE_MAX_LEVELS - 1; k >= 1; --k)
        if (k <= T) { lox[k - 1] = 2 * lox[k] - 2; loy[k - 1] = 2 * loy[k] - 2; }
    lox[0] &= ~3;                      // dword-aligned origin of the level-0 region (P.n[0] is widened to match)
    const int slot = slot0 + (int)blockIdx.z;
    {   // level-0 region from the caller's image.  The region's x origin is rounded down to a multiple of 4
        // and its pitch P.n[0] is a multiple of 4, so it is filled with dwords: every thread first issues all
        // of its (unaligned) global loads, then writes them to LDS -- one memory latency for the whole region
        // instead of one per row batch.
        const VofeLevel L = lv[0];
        const uint8_t* im = src + (size_t)blockIdx.z * img_stride;
        unsigned* r0 = reinterpret_cast<unsigned*>(R + P.off[0]);
        const int dwr = P.n[0] >> 2, total = dwr * P.m[0];
        const float inv_dwr = 1.0f / (float)dwr;
        constexpr int U = 4;
        for (int base = (int)threadIdx.x; base < total; base += 1024 * U) {
            unsigned v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + u * 1024;
                v[u] = 0u;
                if (e >= total) continue;
                const int j = (int)(((float)e + 0.5f) * inv_dwr);          // e / dwr, exact for e < 2^20
                const int cx = lox[0] + 4 * (e - j * dwr), cy = loy[0] + j;
                if (cy < 0 || cy >= L.h) continue;
                const uint8_t* row = im + (size_t)cy * spitch;
                if (cx >= 0 && cx + 3 < L.w) {
                    v[u] = *reinterpret_cast<const __attribute__((address_space(1))) u32_unaligned*>(
                        (const __attribute__((address_space(1))) uint8_t*)(row + cx));
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (cx + b >= 0 && cx + b < L.w) v[u] |= (unsigned)row[cx + b] << (8 * b);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + u * 1024;
                if (e < total) r0[e] = v[u];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 1; k < FUSE_MAX_LEVELS; ++k) {
        if (k > T) break;
        const int wk = lv[k].w, hk = lv[k].h, wp = lv[k - 1].w, hp = lv[k - 1].h;
        const uint8_t* rp = R + P.off[k - 1];
        uint8_t* rk = R + P.off[k];
        const int np = P.n[k - 1];
        for (int j = ty; j < P.m[k]; j += nty) {
            const int Y = loy[k] + j;
            if (Y < 0 || Y >= hk) continue;
            int ry[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) ry[r] = (refl1(2 * Y - 2 + r, hp) - loy[k - 1]) * np;
            for (int i = tx; i < P.n[k]; i += 32) {
                const int X = lox[k] + i;
                if (X < 0 || X >= wk) continue;
                int rx[5];
#pragma unroll
                for (int c = 0; c < 5; ++c) rx[c] = refl1(2 * X - 2 + c, wp) - lox[k - 1];
                int acc = 128;
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const uint8_t* row = rp + ry[r];
                    const int hs = (int)row[rx[0]] + (int)row[rx[4]] + 4 * ((int)row[rx[1]] + (int)row[rx[3]]) + 6 * (int)row[rx[2]];
                    acc += ((r == 0 || r == 4) ? 1 : (r == 2 ? 6 : 4)) * hs;
                }
                rk[j * P.n[k] + i] = (uint8_t)(acc >> 8);
            }
        }
        __syncthreads();
    }
    // owned tiles of every level -> global planes (+ Scharr)
#pragma unroll
    for (int k = 0; k < FUSE_MAX_LEVELS; ++k) {
        if (k > T) break;
        const VofeLevel L = lv[k];
        const int d = T - k;
        const int ox0 = ((int)blockIdx.x * P.tw) << d, oy0 = ((int)blockIdx.y * P.th) << d;
        const int ox1 = min(ox0 + (P.tw << d), L.w), oy1 = min(oy0 + (P.th << d), L.h);
        if (ox0 >= ox1 || oy0 >= oy1) continue;
        const uint8_t* rk = R + P.off[k];
        const int nk = P.n[k];
        const long long so = (long long)slot * L.slot_stride;
        for (int y = oy0 + ty; y < oy1; y += nty) {
            const uint8_t* row = rk + (y - loy[k]) * nk - lox[k];
            uint8_t* drow = L.I + so + (long long)y * L.pitch;
            for (int x = ox0 + tx * 4; x < ox1; x += 128) {
                if (x + 4 <= ox1) {
                    const unsigned v = (unsigned)row[x] | ((unsigned)row[x + 1] << 8) | ((unsigned)row[x + 2] << 16) | ((unsigned)row[x + 3] << 24);
                    *reinterpret_cast<unsigned*>(drow + x) = v;
                } else {
                    for (int q = x; q < ox1; ++q) drow[q] = row[q];
                }
            }
            if (band_w > 0) {
                // rows this source row lands in: itself, its mirror above the image, its mirror below it
                const int xl1 = min(ox1 - 1, band_w), xr0 = max(ox0, L.w - 1 - band_w), xr1 = min(ox1 - 1, L.w - 2);
                for (int q = 0; q < 3; ++q) {
                    int yt = y;
                    if (q == 1) { if (y < 1 || y > band_h) continue; yt = -y; }
                    if (q == 2) { if (y < L.h - 1 - band_h || y > L.h - 2) continue; yt = 2 * (L.h - 1) - y; }
                    uint8_t* trow = L.I + so + (long long)yt * L.pitch;
                    if (q)
                        for (int x = ox0 + tx * 4; x < ox1; x += 128) {
                            if (x + 4 <= ox1) {
                                const unsigned v = (unsigned)row[x] | ((unsigned)row[x + 1] << 8) | ((unsigned)row[x + 2] << 16) | ((unsigned)row[x + 3] << 24);
                                *reinterpret_cast<unsigned*>(trow + x) = v;
                            } else {
                                for (int c = x; c < ox1; ++c) trow[c] = row[c];
                            }
                        }
                    for (int x = max(ox0, 1) + tx; x <= xl1; x += 32) trow[-x] = row[x];
                    for (int x = xr0 + tx; x <= xr1; x += 32) trow[2 * (L.w - 1) - x] = row[x];
                }
            }
            if (with_deriv) {
                const uint8_t* ra = rk + (refl1(y - 1, L.h) - loy[k]) * nk - lox[k];
                const uint8_t* rc = rk + (refl1(y + 1, L.h) - loy[k]) * nk - lox[k];
                int* dd = L.D + so + (long long)y * L.pitch;
                for (int x = ox0 + tx; x < ox1; x += 32) {
                    const int xm = refl1(x - 1, L.w), xp = refl1(x + 1, L.w);
                    const int a0 = ra[xm], a1 = ra[x], a2 = ra[xp];
                    const int b0 = row[xm], b2 = row[xp];
                    const int c0 = rc[xm], c1 = rc[x], c2 = rc[xp];
                    const int ix = (3 * (a2 + c2) + 10 * b2) - (3 * (a0 + c0) + 10 * b0);
                    const int iy = 3 * ((c0 - a0) + (c2 - a2)) + 10 * (c1 - a1);
                    dd[x] = (int)(((unsigned)ix & 0xffffu) | ((unsigned)iy << 16));
                }
            }
        }
    }
}

// REFLECT_101 border bands of every level in one launch (reads the interiors written before).  One thread =
// 16 bytes of one padded row; only border groups are enumerated: first the 2*win_h band rows in full, then the
// left and right band groups of the interior rows (incl. the group that straddles the right image edge).
struct BorderPlan {
    int blk_start[VOFE_MAX_LEVELS + 1];
    int ng[VOFE_MAX_LEVELS];      // 16-byte groups per padded row
    int nin[VOFE_MAX_LEVELS];     // groups that lie fully inside the image
    int levels;
};
__global__ __launch_bounds__(256) void pyramid_borders(const VofeLevel* __restrict__ lv, BorderPlan plan, int slot0,
                                                       int win_w, int win_h) {
    int level = 0;
    while (level + 1 < plan.levels && (int)blockIdx.x >= plan.blk_start[level + 1]) ++level;
    const VofeLevel L = lv[level];
    const int ng = plan.ng[level], nin = plan.nin[level], nl = (win_w + 15) / 16;
    int t = ((int)blockIdx.x - plan.blk_start[level]) * 256 + (int)threadIdx.x;
    int y, g;
    const int nband = 2 * win_h * ng;
    if (t < nband) {
        const int row = t / ng;
        g = t - row * ng;
        y = row < win_h ? row - win_h : L.h + (row - win_h);
    } else {
        t -= nband;
        const int nb = ng - nin;
        y = t / nb;
        if (y >= L.h) return;
        const int k = t - y * nb;
        g = k < nl ? k : k + nin;
    }
    const int x0 = (g - nl) * 16;
    uint8_t* base = L.I + (long long)(slot0 + (int)blockIdx.y) * L.slot_stride;
    const uint8_t* srow = base + (long long)vofe_reflect101(y, L.h) * L.pitch;
    *reinterpret_cast<uint4*>(base + (long long)y * L.pitch + x0) = mirror16(srow, x0, L.w);
}

// The same when every level is at least 16 * nl + 17 pixels wide, laid out like pad_copy_level0_wide: nl mirrored
// groups left of the image, the image groups (band rows only; the last one shifted to end at the last pixel), nl
// mirrored groups starting right after the last pixel.  No group straddles an image edge, so no thread gathers bytes.
struct BorderPlanW {
    int blk_start[VOFE_MAX_LEVELS + 1];
    int levels;
};
__global__ __launch_bounds__(256) void pyramid_borders_wide(const VofeLevel* __restrict__ lv, BorderPlanW plan, int slot0,
                                                            int win_w, int win_h) {
    int level = 0;
    while (level + 1 < plan.levels && (int)blockIdx.x >= plan.blk_start[level + 1]) ++level;
    const VofeLevel L = lv[level];
    const int nl = (win_w + 15) / 16, ni = (L.w + 15) / 16, ng = 2 * nl + ni;
    int t = ((int)blockIdx.x - plan.blk_start[level]) * 256 + (int)threadIdx.x;
    int y, g;
    const int nband = 2 * win_h * ng;
    if (t < nband) {
        const int row = t / ng;
        g = t - row * ng;
        y = row < win_h ? row - win_h : L.h + (row - win_h);
    } else {
        t -= nband;
        y = t / (2 * nl);
        if (y >= L.h) return;
        const int k = t - y * 2 * nl;
        g = k < nl ? k : k + ni;
    }
    uint8_t* base = L.I + (long long)(slot0 + (int)blockIdx.y) * L.slot_stride;
    const uint8_t* srow = base + (long long)vofe_reflect101(y, L.h) * L.pitch;
    const int jr = g - nl - ni;                                    // >= 0: right-hand group
    const bool rev = g < nl || jr >= 0;
    const int dx = g < nl ? -16 * (nl - g) : (jr >= 0 ? L.w + 16 * jr : min(16 * (g - nl), L.w - 16));
    const int sx = g < nl ? -dx - 15 : (jr >= 0 ? L.w - 17 - 16 * jr : dx);   // mirrored: out[i] = src[sx + 15 - i]
    uint4 v = load16(srow + sx);
    const uint4 r = reversed16(v);
    if (rev) v = r;
    store16(base + (long long)y * L.pitch + dx, v);
}

// Launches the border kernel for levels first_level.. (the wide one when every such level is wide enough).
void launch_borders(const vofe_ctx* ctx, const vofe_pyr* pyr, int first_level, int slot0, int count, hipStream_t st);

BorderPlan border_plan(const vofe_pyr* pyr, int first_level, int* total_blocks) {
    BorderPlan plan;
    plan.levels = pyr->levels;
    int total = 0;
    for (int l = 0; l < pyr->levels; ++l) {
        const int nl = (pyr->win_w + 15) / 16;
        plan.ng[l] = nl + (pyr->lv[l].w + pyr->win_w + 15) / 16;
        plan.nin[l] = pyr->lv[l].w / 16;
        plan.blk_start[l] = total;
        if (l < first_level) continue;
        const long long items = 2ll * pyr->win_h * plan.ng[l] + (long long)pyr->lv[l].h * (plan.ng[l] - plan.nin[l]);
        total += (int)((items + 255) / 256);
    }
    plan.blk_start[pyr->levels] = total;
    *total_blocks = total;
    return plan;
}

inline int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }   // floor division

void launch_borders(const vofe_ctx* ctx, const vofe_pyr* pyr, int first_level, int slot0, int count, hipStream_t st) {
    const int ww = pyr->win_w, wh = pyr->win_h, nl = (ww + 15) / 16;
    bool wide = !ctx->knobs.pad_copy_generic;                       // VOFE_PAD_COPY=generic forces the any-width kernels (tests)
    for (int l = first_level; l < pyr->levels; ++l) wide = wide && pyr->lv[l].w >= 16 * nl + 17;
    int total = 0;
    if (wide) {
        BorderPlanW plan;
        plan.levels = pyr->levels;
        for (int l = 0; l < pyr->levels; ++l) {
            plan.blk_start[l] = total;
            if (l < first_level) continue;
            const long long ng = 2 * nl + (pyr->lv[l].w + 15) / 16;
            total += (int)((2ll * wh * ng + 2ll * nl * pyr->lv[l].h + 255) / 256);
        }
        plan.blk_start[pyr->levels] = total;
        VOFE_TIMED(const_cast<vofe_ctx*>(ctx), "pyramid_borders_wide");
        if (total > 0) hipLaunchKernelGGL(pyramid_borders_wide, dim3(total, count), dim3(256), 0, st, pyr->d_lv, plan, slot0, ww, wh);
    } else {
        const BorderPlan plan = border_plan(pyr, first_level, &total);
        VOFE_TIMED(const_cast<vofe_ctx*>(ctx), "pyramid_borders");
        if (total > 0) hipLaunchKernelGGL(pyramid_borders, dim3(total, count), dim3(256), 0, st, pyr->d_lv, plan, slot0, ww, wh);
    }
}

}  // namespace

int vofe_launch_pyramid(vofe_ctx* ctx, vofe_pyr* pyr, int slot0, int count, const uint8_t* d_img,
                        int pitch, size_t img_stride) {
    hipStream_t st = ctx->stream;
    const int ww = pyr->win_w, wh = pyr->win_h;
    // Small frames are launch/latency bound: one fused launch for all levels wins (KITTI pair 21.6 vs 30 us).
    // Large frames are bandwidth bound and the halo recompute of the fused kernel costs more than the extra
    // launches (3840x2160 pair: 134 us fused vs 77 us per-level), so they take the per-level kernels.
    const char force = ctx->knobs.pyramid;               // VOFE_PYRAMID: "fused" / "levels" force one path (tests, measurements)
    const bool small = (long long)pyr->height * pyr->width * count <= 2000000ll;      // measured cross-over
    const bool fused = force ? force == 'f' : small;
    if (pyr->levels <= FUSE_MAX_LEVELS && fused) {
        // fused path: one launch (interiors + Scharr + border bands), or two when a level is too small for that
        FusePlan P;
        P.levels = pyr->levels;
        const int T = P.levels - 1;
        P.tw = P.th = (T == 0) ? 64 : (T == 1 ? 32 : (T == 2 ? 16 : 8));
        // a single small frame has fewer square tiles than the chip has CUs (KITTI: 120 blocks): half-height tiles put a
        // block on every CU at 1.2x the halo work (one-frame build 27.2 -> 25.5 us; a pair, 240 blocks, is better off square)
        {
            const VofeLevel& Lq = pyr->lv[T];
            const long long blocks = (long long)((Lq.w + P.tw - 1) / P.tw) * ((Lq.h + P.th - 1) / P.th) * count;
            if (blocks < 192 && P.th >= 8) P.th /= 2;
        }
        P.halo = pyr->with_deriv ? 1 : 0;
        P.n[T] = P.tw + 2 * P.halo; P.m[T] = P.th + 2 * P.halo;
        for (int k = T; k >= 1; --k) { P.n[k - 1] = 2 * P.n[k] + 3; P.m[k - 1] = 2 * P.m[k] + 3; }
        P.n[0] = (P.n[0] + 3 + 3) & ~3;       // level-0 region: origin rounded down to x % 4 == 0, dword pitch
        int off = 0;
        for (int k = 0; k <= T; ++k) { P.off[k] = off; off += (P.n[k] * P.m[k] + 15) / 16 * 16; }
        for (int k = T + 1; k < FUSE_MAX_LEVELS; ++k) { P.n[k] = P.m[k] = P.off[k] = 0; }
        const VofeLevel& Lt = pyr->lv[T];
        dim3 grid((Lt.w + P.tw - 1) / P.tw, (Lt.h + P.th - 1) / P.th, count);
        if (grid.y > 65535) { vofe_set_error("pyramid: image too tall"); return VOFE_E_UNSUPPORTED; }
        // the border bands in the same launch when one reflection reaches them on every level (the owners of the source
        // pixels scatter them); VOFE_PYR_BORDERS=separate or VOFE_PAD_COPY=generic keep the second launch (tests, A/B)
        const int band_w = 16 * ((ww + 15) / 16);
        bool own_bands = !ctx->knobs.pyr_borders_separate && !ctx->knobs.pad_copy_generic;
        for (int l = 0; l < pyr->levels; ++l) own_bands = own_bands && pyr->lv[l].w >= band_w + 17 && pyr->lv[l].h >= wh + 1;
        {
        VOFE_TIMED(ctx, "pyramid_fused");
        hipLaunchKernelGGL(pyramid_fused, grid, dim3(1024), (size_t)off, st, d_img, pitch, img_stride, pyr->d_lv, P, slot0,
                           pyr->with_deriv, own_bands ? band_w : 0, own_bands ? wh : 0);
        }
        if (!own_bands) launch_borders(ctx, pyr, 0, slot0, count, st);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return vofe_hip_fail(e, "pyramid launch");
        return VOFE_OK;
    }
    {
        const VofeLevel& L = pyr->lv[0];
        const int nb = (ww + 15) / 16;                              // mirrored groups per side
        if (L.w >= 16 * nb + 17 && !ctx->knobs.pad_copy_generic) {   // VOFE_PAD_COPY=generic forces the any-width kernel (tests)
            const int ni = (L.w + 15) / 16;
            const long long items = (long long)(2 * nb + ni) * (L.h + 2 * wh);
            if (items > 0x7fffff00ll) { vofe_set_error("pyramid: image too large"); return VOFE_E_UNSUPPORTED; }
            VOFE_TIMED(ctx, "pad_copy_level0_wide");
            hipLaunchKernelGGL(pad_copy_level0_wide, dim3((unsigned)((items + 255) / 256), count), dim3(256), 0, st, d_img,
                               pitch, img_stride, L, slot0, wh, nb, ni, nb, (int)items);
        } else {
            const int gx0 = fdiv(-ww, 16), gx1 = fdiv(L.w + ww - 1, 16);
            const int ngx = gx1 - gx0 + 1;
            const long long items = (long long)ngx * (L.h + 2 * wh);
            if (items > 0x7fffff00ll) { vofe_set_error("pyramid: image too large"); return VOFE_E_UNSUPPORTED; }
            VOFE_TIMED(ctx, "pad_copy_level0");
            hipLaunchKernelGGL(pad_copy_level0, dim3((unsigned)((items + 255) / 256), count), dim3(256), 0, st, d_img, pitch,
                               img_stride, L, slot0, wh, gx0, ngx, (int)items);
        }
    }
    // border bands of levels 1..: written by pyr_down_u8 itself when one reflection reaches them on every such level
    // (VOFE_PYR_BORDERS=separate keeps the border launch: tests, A/B), else by one pyramid_borders launch behind the chain
    bool own_bands = !ctx->knobs.pyr_borders_separate;
    for (int l = 1; l < pyr->levels; ++l) own_bands = own_bands && pyr->lv[l].w >= ww + 1 && pyr->lv[l].h >= wh + 1;
    // level pipeline (needs the bands in the pyrDown launches and a derivative plane; VOFE_PYR_STEP=0 keeps the plain sequence):
    // launch k = pyrDown (k-1 -> k) + a share of the Scharr blocks of the levels below k; the last one or two levels and what
    // is left of the Scharr blocks in one pyr_tail launch (VOFE_PYR_TAIL=0: pyrDown launches up to the top level, then the
    // top level's Scharr pass alone)
    const bool stepped = own_bands && pyr->with_deriv && pyr->levels > 1 && pyr->levels <= STEP_SEGS && !ctx->knobs.pyr_step_off;
    const int top = pyr->levels - 1;
    const bool tail = stepped && pyr->levels >= 3 && !ctx->knobs.pyr_tail_off;
    const int base = tail ? (top - 2 > 1 ? top - 2 : 1) : top;                 // levels above `base` are the tail's
    // Scharr blocks of levels 0 .. riders-1 in level order ride on the pipeline's launches; launch k (1-based, the tail being
    // the last) sees levels 0 .. k-1 and takes half of what is left (the finest levels are the bulk)
    const int riders = tail ? base + 1 : top;
    long long cum[VOFE_MAX_LEVELS + 1];
    cum[0] = 0;
    for (int q = 0; q < riders && stepped; ++q)
        cum[q + 1] = cum[q] + (long long)((pyr->lv[q].w + 511) / 512) * ((pyr->lv[q].h + 3) / 4);
    auto share = [&](int k) {                                       // blocks handed out after launch k
        if (k >= riders) return cum[riders];
        const double denom = 1.0 - 1.0 / (double)(1ll << riders);
        long long c = (long long)((double)cum[riders] * (1.0 - 1.0 / (double)(1ll << k)) / denom);
        return c < cum[k] ? c : cum[k];
    };
    auto rider_plan = [&](long long c0, long long c1) {
        StepPlan plan;
        plan.nseg = 0;
        plan.start[0] = 0;
        for (int q = 0; q < riders; ++q) {
            const long long a = c0 > cum[q] ? c0 : cum[q], b = c1 < cum[q + 1] ? c1 : cum[q + 1];
            if (b <= a) continue;
            const int sgi = plan.nseg++;
            plan.level[sgi] = q; plan.blk_x[sgi] = (pyr->lv[q].w + 511) / 512; plan.first[sgi] = (int)(a - cum[q]);
            plan.start[sgi + 1] = plan.start[sgi] + (int)(b - a);
        }
        for (int sgi = plan.nseg; sgi < STEP_SEGS; ++sgi) { plan.level[sgi] = 0; plan.blk_x[sgi] = 1; plan.first[sgi] = 0; plan.start[sgi + 1] = plan.start[plan.nseg]; }
        return plan;
    };
    for (int l = 1; l <= base; ++l) {
        const VofeLevel& S = pyr->lv[l - 1];
        const VofeLevel& D = pyr->lv[l];
        const int gpr = (D.w + 7) / 8;
        const int items = gpr * ((D.h + 1) / 2);
        const int pd_blocks = (items + 255) / 256;
        if (stepped) {
            const StepPlan plan = rider_plan(share(l - 1), share(l));
            const long long total = ((long long)pd_blocks + plan.start[plan.nseg]) * count;
            if (total > 0x7fffffffll) { vofe_set_error("pyramid: image too large"); return VOFE_E_UNSUPPORTED; }
            VOFE_TIMED(ctx, "pyr_step", l);
            hipLaunchKernelGGL(pyr_step, dim3((unsigned)total), dim3(256), 0, st, pyr->d_lv, S, D, slot0, gpr, items, ww, wh,
                               pd_blocks, count, plan);
            continue;
        }
        VOFE_TIMED(ctx, "pyr_down_u8", l);
        if (own_bands)
            hipLaunchKernelGGL(pyr_down_u8<true>, dim3(pd_blocks, count), dim3(256), 0, st, S, D, slot0, gpr, items, ww, wh);
        else
            hipLaunchKernelGGL(pyr_down_u8<false>, dim3(pd_blocks, count), dim3(256), 0, st, S, D, slot0, gpr, items, ww, wh);
    }
    if (tail) {
        TailPlan P;
        memset(&P, 0, sizeof(P));
        P.D = top - base;                                           // 1 or 2
        P.base = base;
        const int T = P.D - 1;
        if (P.D == 2) { P.tw = 64; P.th = ctx->knobs.pyr_tail_th == 8 ? 8 : 16; P.halo = 1; P.cell0[0] = 2; P.cell0[1] = 3; }
        else { P.tw = 128; P.th = 32; P.halo = 2; P.cell0[0] = 2; }   // (halo 2: the region's origin must be even for the aligned source loads)
        P.n[T] = P.tw + 2 * P.halo; P.m[T] = P.th + 2 * P.halo;
        if (T == 1) { P.n[0] = 2 * P.n[1] + 3; P.m[0] = 2 * P.m[1] + 3; }
        int off = 0;
        for (int r = 0; r <= T; ++r) {
            int pitch = P.cell0[r] + (P.n[r] + 7) / 8 * 8 + 16;
            if (r == 0 && T == 1 && pitch < 16 * ((P.n[1] + 7) / 8) + 8) pitch = 16 * ((P.n[1] + 7) / 8) + 8;
            P.pitch[r] = (pitch + 15) / 16 * 16;
            P.off[r] = off;
            off += P.pitch[r] * P.m[r];
        }
        if (off > TAIL_LDS) { vofe_set_error("pyramid: tail plan needs %d bytes of LDS", off); return VOFE_E_UNSUPPORTED; }
        const VofeLevel& Lt = pyr->lv[top];
        P.tiles_x = (Lt.w + P.tw - 1) / P.tw;
        P.tiles = P.tiles_x * ((Lt.h + P.th - 1) / P.th);
        const StepPlan plan = rider_plan(share(base), share(base + 1));
        const long long total = ((long long)P.tiles + plan.start[plan.nseg]) * count;
        if (total > 0x7fffffffll) { vofe_set_error("pyramid: image too large"); return VOFE_E_UNSUPPORTED; }
        if (ctx->knobs.pyr_tail_split) {                            // measurement: tiles and riders as two launches
            {
                VOFE_TIMED(ctx, "pyr_tail tiles", base + 1);
                hipLaunchKernelGGL(pyr_tail, dim3((unsigned)(P.tiles * count)), dim3(256), 0, st, pyr->d_lv, pyr->lv[base], pyr->lv[base + 1],
                                   pyr->lv[top], slot0, ww, wh, P, count, rider_plan(0, 0));
            }
            TailPlan Q = P;
            Q.tiles = 0;
            VOFE_TIMED(ctx, "pyr_tail riders", base + 1);
            hipLaunchKernelGGL(pyr_tail, dim3((unsigned)(plan.start[plan.nseg] * count)), dim3(256), 0, st, pyr->d_lv, pyr->lv[base], pyr->lv[base + 1],
                               pyr->lv[top], slot0, ww, wh, Q, count, plan);
        } else {
            VOFE_TIMED(ctx, "pyr_tail", base + 1);
            hipLaunchKernelGGL(pyr_tail, dim3((unsigned)total), dim3(256), 0, st, pyr->d_lv, pyr->lv[base], pyr->lv[base + 1],
                               pyr->lv[top], slot0, ww, wh, P, count, plan);
        }
    }
    if (pyr->levels > 1 && !own_bands) {
        // border bands of levels 1.. (level 0 got its border from pad_copy_level0: no blocks for it)
        launch_borders(ctx, pyr, 1, slot0, count, st);
    }
    if (pyr->with_deriv) {
        ScharrPlan plan;
        plan.levels = pyr->levels;
        int total = 0;
        const int first = tail ? pyr->levels : (stepped ? pyr->levels - 1 : 0);     // the pipeline has done every level but the last, its tail that one too
        for (int l = 0; l < pyr->levels; ++l) {
            plan.blk_start[l] = total;
            plan.blk_x[l] = (pyr->lv[l].w + 511) / 512;
            if (l >= first) total += plan.blk_x[l] * ((pyr->lv[l].h + 3) / 4);
        }
        plan.blk_start[pyr->levels] = total;
        if (total > 0) {
            VOFE_TIMED(ctx, stepped ? "scharr_deriv_s16x2 last" : "scharr_deriv_s16x2");
            hipLaunchKernelGGL(scharr_deriv_s16x2, dim3(total, count), dim3(256), 0, st, pyr->d_lv, plan, slot0);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "pyramid launch");
    return VOFE_OK;
}
