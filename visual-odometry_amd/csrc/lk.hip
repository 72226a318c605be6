// Pyramidal Lucas-Kanade tracker for gfx950: one wavefront (64 lanes) per keypoint, all pyramid
// levels, the optional backward pass and the forward-backward check inside one launch.
//
// Replaces cv::LKTrackerInvoker (cv2.calcOpticalFlowPyrLK at reference continuous.py:145-146,
// 150-151, initialization.py:51-52) and the glue of ContinuousVO._track_keypoints
// (continuous.py:131-164).  Arithmetic per SURVEY.md A.5 with the canonical choices stated in
// oracle/vofe_oracle.cpp: integer window sums are exact (order-free), float steps are evaluated
// in the written order with -ffp-contract=off, so results are bit-identical to the oracle.
//
// Mapping: lane -> (column x of the window, row group g); a lane owns up to R vertically adjacent
// window pixels, so the bilinear taps of row r+1 are reused as the top taps of the next pixel.
// The template window (I, Ix, Iy at the fixed prev position) lives in registers; the search image
// J is staged once per level into LDS as (J[x] | J[x+1] << 16) pairs around the start position
// (margin LK_MARGIN px) and re-staged only if the Gauss-Newton walk leaves it.  Window sums are
// reduced with DPP row rotations + 4 readlanes.  Not HBM- or MFMA-bound: gather + VALU + LDS.
#include <cmath>

#include <cstring>

#include "vofe_internal.h"

namespace {

#ifndef LK_MARGIN_PX
#define LK_MARGIN_PX 4
#endif
constexpr int LK_MARGIN = LK_MARGIN_PX;
constexpr int W_BITS = 14;

struct LKArgs {
    const VofeLevel* lv_a;     // pyramid in the "prev" role of the forward pass
    const VofeLevel* lv_b;     // pyramid in the "next" role of the forward pass
    int n_levels;
    int slot_a, slot_b;
    const float2* prev_pts;
    const int* img_idx;
    int n_images;              // valid image indices are [0, n_images); anything else loses the point (status 0)
    float2* next_pts;
    uint8_t* status;
    float* err;
    int n;
    int win_w, win_h;
    int groups, rows_per_group;
    int max_count, flags, fb;
    float min_eig_thr, fb_thresh;
    float min_eig_num;         // smallest x with fl(x / (2 win_w win_h)) >= min_eig_thr: "minEig < thr" == "numerator < min_eig_num"
    double eps2;
    float eps2_lo, eps2_hi;    // f32 brackets of eps2 for the cheap convergence pre-test
    int rh, rs;                // LDS region: rows, row stride in entries (8 * segs)
    int segs, inv_segs;        // 8-entry segments per row; (t * inv_segs) >> 20 == t / segs for all tasks
    unsigned long long* counters;   // optional {points, point-levels entered, GN iterations}
};

__device__ __forceinline__ int floor_i(float v) {
    const int i = (int)__builtin_floorf(v);                              // v_floor_f32 + v_cvt_i32_f32, exact below 2^30
    return __builtin_fabsf(v) < 1073741824.f ? i : INT_MIN;              // NaN / huge -> INT_MIN (fails every bounds test)
}
// floor to int32 by v_floor_f32 + v_cvt_i32_f32 alone: the conversion saturates (huge -> INT_MAX / INT_MIN) and maps NaN to 0;
// callers that cannot tolerate the latter test for NaN themselves
__device__ __forceinline__ int floor_sat(float v) { return (int)__builtin_floorf(v); }
__device__ __forceinline__ int round_i(float v) { return __float2int_rn(v); }
// Correctly rounded 1 / d for d whose reciprocal is a normal number: the refinement steps of the IEEE division sequence the
// compiler emits for 1.f / d (reciprocal estimate, one Newton step, quotient, two residual corrections), without its operand
// scaling and special-case fix-up, which are identities on that range.  Used for the 2x2 determinant, which has passed the
// FLT_EPSILON test and is bounded by the window's structure tensor (<= ~1e10).
__device__ __forceinline__ float rcp_exact_normal(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;                                   // 1.0f * r
    float rem = __builtin_fmaf(-d, q, 1.0f);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(rem, r, q);
}
// 1 / 2^level, exactly, from the exponent field (the IEEE division sequence of level_scale(level) is twelve instructions)
__device__ __forceinline__ float level_scale(int level) { return __builtin_bit_cast(float, (127 - level) << 23); }

// signed 2-way dot product: pixel pairs (0..255) x weight pairs.  Signed on purpose: the fourth
// bilinear weight 16384 - w00 - w01 - w10 can come out as -1 after rounding.
typedef short ss2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sdot2(unsigned a, unsigned b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(ss2, a), __builtin_bit_cast(ss2, b), c, false);
}

// Two window pixels per register.  `pair` holds a value in its low half; the result keeps that half and takes
// (src >> sh) & 0xffff as its high half: the arithmetic shift every bilinear sample ends with, done by the SDWA form of the
// shift so that it deposits its result in the upper word -- packing two differences costs no instruction beyond the two
// shifts they need anyway, and the sums of products b1 / b2 then take one v_dot2 per pixel PAIR instead of one multiply-add
// per pixel.  (`sh` lives in a register: SDWA takes no literal.)
// Hazard note (gfx940 family): a VALU instruction that reads a register directly after an SDWA instruction wrote PART of it
// (dst_sel != DWORD) needs one wait state; LLVM inserts it for SDWA instructions it selected itself but cannot see into inline
// assembly.  The build therefore checks the generated assembly (tools/isa_audit.py --hazard, run by build.py): no such
// instruction may directly follow this one.  -DLK_SDWA_NOP appends the wait state to the assembly text instead (one s_nop per
// packed pair, ~2 % of the iteration's instructions) should a future compiler schedule a reader there.
#ifdef LK_SDWA_NOP
#define LK_SDWA_TAIL "\n\ts_nop 0"
#else
#define LK_SDWA_TAIL ""
#endif
__device__ __forceinline__ int shr_into_high_half(int pair, int src, int sh) {
    asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" LK_SDWA_TAIL
        : "+v"(pair) : "v"(sh), "v"(src));
    return pair;
}
// exact wave-wide sum of one int32 per lane, as int64 (all 64 lanes must be active)
__device__ __forceinline__ int row_sum16(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);   // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false);   // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, false);   // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false);   // row_ror:1
    return v;
}
__device__ __forceinline__ int wave_sum_small(int v) {                 // |sum| must fit int32
    v = row_sum16(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
// (float) of the exact 64-bit wave sum (one rounding, like the oracle's (float)int64).  The split
// into 16-bit halves keeps every partial inside int32; when the total itself fits int32 (the
// common case) a single v_cvt_f32_i32 replaces the 64-bit conversion sequence.
__device__ __forceinline__ float wave_sum_to_float(int v) {
    const int lo = wave_sum_small(v & 0xffff);            // 0 <= lo < 2^22
    const int hi = wave_sum_small(v >> 16);
    if (hi >= -32768 && hi < 32000) return (float)(hi * 65536 + lo);
    return (float)((long long)hi * 65536 + (long long)lo);
}

// Common case: every lane's partial is below 2^25 in magnitude, so the 64-lane total fits int32 and
// one DPP reduction per value replaces the two half-word reductions (same exact integer, same
// single rounding).  The test is wave-uniform.
__device__ __forceinline__ bool partials_fit(int a, int b, int c = 0) {
    const unsigned m = (unsigned)abs(a) | (unsigned)abs(b) | (unsigned)abs(c);
    return !__any(m >= (1u << 25));
}

struct Wts { unsigned w0, w1; int w00, w01, w10, w11; };
// weights packed by v_cvt_pk_i16_i32 (values are within int16: one instruction per pair instead of shift + and-or)
__device__ __forceinline__ unsigned pack_i16(int lo, int hi) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(lo, hi)); }
// The 14-bit bilinear weights from the fractional parts a = fx - floor(fx), b = fy - floor(fy):
//   w00 = round((1-a)(1-b) 2^14), w01 = round(a (1-b) 2^14), w10 = round((1-a) b 2^14), w11 = 2^14 - w00 - w01 - w10.
// The factor 2^14 is applied to the x terms first: scaling by a power of two commutes with rounding, fl(fl(u v) 2^14) ==
// fl((u 2^14) v) bit for bit, and five multiplications replace six.
__device__ __forceinline__ Wts bilinear_weights_frac(float a, float b) {
    Wts q;
    const float u = 1.f - a, v = 1.f - b;
    const float u14 = u * (float)(1 << W_BITS), a14 = a * (float)(1 << W_BITS);
    q.w00 = round_i(u14 * v);
    q.w01 = round_i(a14 * v);
    q.w10 = round_i(u14 * b);
    q.w11 = (1 << W_BITS) - q.w00 - q.w01 - q.w10;
    q.w0 = pack_i16(q.w00, q.w01);
    q.w1 = pack_i16(q.w10, q.w11);
    return q;
}
__device__ __forceinline__ Wts bilinear_weights(float fx, float fy, int ix, int iy) {
    return bilinear_weights_frac(fx - (float)ix, fy - (float)iy);
}
// the three partial sums of the structure tensor: a11, a22 >= 0 (sums of squares), a12 signed; all inside +-2^25
__device__ __forceinline__ bool partials_fit_tensor(int a11, int a12, int a22) {
    const unsigned m = (unsigned)(a12 + (1 << 25)) | ((unsigned)(a11 | a22) << 1);
    return !__any(m >= (1u << 26));
}
__device__ __forceinline__ bool partials_fit2(int a, int b) {
    const unsigned m = (unsigned)(a + (1 << 25)) | (unsigned)(b + (1 << 25));
    return !__any(m >= (1u << 26));
}

// Explicit global-address-space loads at byte alignment: the plane pointers come out of a table in
// memory, so without the cast the compiler has to emit flat_load (which also ties up lgkmcnt).
#define VOFE_GLOBAL __attribute__((address_space(1)))
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((aligned(1))) u32x2_unaligned;
typedef unsigned short __attribute__((aligned(1))) u16_unaligned;
__device__ __forceinline__ uint2 gload_u64(const void* p) {
    const u32x2 v = *((const VOFE_GLOBAL u32x2_unaligned*)p);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ unsigned gload_u16(const void* p) { return *((const VOFE_GLOBAL u16_unaligned*)p); }
__device__ __forceinline__ unsigned gload_u8(const void* p) { return *((const VOFE_GLOBAL uint8_t*)p); }
// Wave-uniform 64-bit base (SGPR pair) + unsigned 32-bit lane offset: the global_load saddr form, so the
// per-lane address arithmetic stays 32-bit (no v_mad_i64 / 64-bit adds per load).
typedef const VOFE_GLOBAL uint8_t* gbase_t;
__device__ __forceinline__ uint2 gload_u64(gbase_t b, unsigned off) {
    const u32x2 v = *((const VOFE_GLOBAL u32x2_unaligned*)(b + off));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ unsigned gload_u16(gbase_t b, unsigned off) { return *((const VOFE_GLOBAL u16_unaligned*)(b + off)); }
__device__ __forceinline__ unsigned gload_u8(gbase_t b, unsigned off) { return *(b + off); }
// 12 bytes in one load: the 8 staged pixels of a task plus the ninth byte its last pair needs (x = lo, y = hi,
// z = next dword; bytes 9..11 are read and ignored: they lie in the plane's padding at worst)
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 __attribute__((aligned(1))) u32x3_unaligned;
__device__ __forceinline__ u32x3 gload_u96(gbase_t b, unsigned off) { return *((const VOFE_GLOBAL u32x3_unaligned*)(b + off)); }

// Diagnostic build only (-DLK_STAMPS): s_memtime deltas per phase, written to counters[8 + 8*block + k]
// for the first 256 blocks.  Never enabled in the shipped library.
#ifdef LK_STAMPS
#define STAMP_DECL unsigned st_t0 = 0;
#define STAMP_BEGIN() do { __builtin_amdgcn_sched_barrier(0); st_t0 = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_END(k) do { __builtin_amdgcn_sched_barrier(0); unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); st_acc[k] += t_ - st_t0; st_t0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_N 12
#else
#define STAMP_DECL
#define STAMP_BEGIN() do {} while (0)
#define STAMP_END(k) do {} while (0)
#define STAMP_N 1
#endif

// ---- staging of the search image J into LDS -------------------------------------------------
// LDS image: rh rows x rs entries (rs = 8 * segs >= rw), entry = J[x] | J[x+1] << 16, top-left pixel
// (rx0, ry0).  One task = one row x 8 entries = 9 consecutive bytes of J (an unaligned 8-byte load
// plus one byte); a lane runs up to ST_BATCH tasks per batch with all loads issued before the first
// LDS write, so a staging pass costs one memory latency, not one per task.
constexpr int ST_BATCH = 2;
struct StageRegs { uint2 v[ST_BATCH]; unsigned nb[ST_BATCH]; };

// WIDE: one 12-byte load per task instead of 8 + 1 bytes.  Measured: a gain for the 21x21 kernels (fewer
// vector-memory instructions), a 20 % loss for the 31x31 launch of the stress configuration, so only the
// compile-time 21x21 paths take it.
template <bool WIDE>
__device__ __forceinline__ void stage_load(StageRegs& s, const uint8_t* __restrict__ J0, const VofeLevel& B, int win_w,
                                           int win_h, int rx0, int ry0, int segs, int inv_segs, int tasks, int t0, int lane) {
    const int ymin = -win_h, ymax = B.h + win_h - 1;
    // uniform base = first byte any task can touch (row ymin, column rx0); rx0 >= -win_w - 4 lies in the padding
    const gbase_t base = (gbase_t)(J0 + (long long)ymin * B.pitch + rx0);
#pragma unroll
    for (int k = 0; k < ST_BATCH; ++k) {
        const int t = t0 + k * 64 + lane;
        const int row = (int)(((unsigned)t * (unsigned)inv_segs) >> 20);
        const int seg = t - row * segs;
        s.v[k] = make_uint2(0u, 0u);
        s.nb[k] = 0u;
        if (t < tasks) {
            const int gy = min(max(ry0 + row, ymin), ymax);
            const unsigned off = (unsigned)((gy - ymin) * B.pitch + seg * 8);
            if (WIDE) {
                const u32x3 q = gload_u96(base, off);
                s.v[k] = make_uint2(q.x, q.y);
                s.nb[k] = q.z;
            } else {
                s.v[k] = gload_u64(base, off);
                s.nb[k] = gload_u8(base, off + 8u);
            }
        }
    }
}
__device__ __forceinline__ void stage_write(const StageRegs& s, unsigned* lds, int segs, int inv_segs, int tasks, int t0,
                                            int lane) {
#pragma unroll
    for (int k = 0; k < ST_BATCH; ++k) {
        const int t = t0 + k * 64 + lane;
        if (t < tasks) {
            const unsigned lo = s.v[k].x, hi = s.v[k].y, nb = s.nb[k];
            uint4 a, b;
            a.x = __builtin_amdgcn_perm(0u, lo, 0x0c010c00u);
            a.y = __builtin_amdgcn_perm(0u, lo, 0x0c020c01u);
            a.z = __builtin_amdgcn_perm(0u, lo, 0x0c030c02u);
            a.w = __builtin_amdgcn_perm(hi, lo, 0x0c040c03u);
            b.x = __builtin_amdgcn_perm(0u, hi, 0x0c010c00u);
            b.y = __builtin_amdgcn_perm(0u, hi, 0x0c020c01u);
            b.z = __builtin_amdgcn_perm(0u, hi, 0x0c030c02u);
            b.w = __builtin_amdgcn_perm(nb, hi, 0x0c040c03u);
            uint4* d = reinterpret_cast<uint4*>(lds + t * 8);       // row * rs + seg * 8 == t * 8
            d[0] = a; d[1] = b;
        }
    }
}
// remaining batches (windows wider than ~31 px) and re-staging after the walk left the region
template <bool WIDE>
__device__ __forceinline__ void stage_rest(unsigned* lds, const uint8_t* __restrict__ J0, const VofeLevel& B, int win_w, int win_h,
                                           int rx0, int ry0, int segs, int inv_segs, int tasks, int t_begin, int lane) {
    for (int t0 = t_begin; t0 < tasks; t0 += 64 * ST_BATCH) {
        StageRegs s;
        stage_load<WIDE>(s, J0, B, win_w, win_h, rx0, ry0, segs, inv_segs, tasks, t0, lane);
        stage_write(s, lds, segs, inv_segs, tasks, t0, lane);
    }
}

// one wave per level pays while the extra waves find idle SIMDs: measured (tools/lp_sweep.sh, KITTI / Parking frames, fwd + bwd)
// 300 points 38.2 -> 34.5 us, 500 points 39.2 -> 35.8 us, break-even near 1000 points (4 waves per SIMD), 30 % slower at 2000
constexpr int LK_LP_MAX_POINTS = 640;
constexpr int LK_X2_MIN_POINTS = 3200;     // measured with the rows packed two per register (lk_bench.py, KITTI): equal to one per wavefront at 2000-3000 points, 7 % ahead at 3500, 11 % at 4000, 21 % at 6000
constexpr int LK_NO_ERR = 1 << 16;     // internal flag: the backward pass of the FB check has no use for err

// WC != 0: compile-time square window WC x WC, so window geometry, the LDS row stride and all row offsets fold
// to constants (LDS reads take immediate offsets).  When R divides WC the mapping is regular (REG): every active
// lane owns exactly R rows and idle lanes sit on row 0 with ny == 0; otherwise the last row group is ragged and
// rows are clamped / masked per lane as in the run-time case.  WC == 0: everything at run time.
template <int R, int WC>
__device__ __forceinline__ void lk_one(const LKArgs& a, const VofeLevel* __restrict__ LA, const VofeLevel* __restrict__ LB,
                                       int slotA, int slotB, float px, float py, float& outx, float& outy,
                                       int& status, float& err, int flags, unsigned* lds, int lane, int x, int y0,
                                       int ny, unsigned& n_lv, unsigned& n_it, unsigned* st_acc) {
    STAMP_DECL
    (void)st_acc;
    constexpr bool REG = WC != 0 && WC % R == 0;
    constexpr bool WIDE = WC == 21;                  // 12-byte staging loads: measured slower on the larger windows
    const int w = WC ? WC : a.win_w, h = WC ? WC : a.win_h;
    const float hx = (float)(w - 1) * 0.5f, hy = (float)(h - 1) * 0.5f;
    const int top = a.n_levels - 1;
    constexpr int SEGS_C = (WC + 1 + 2 * LK_MARGIN + 7) / 8;
    const int segs = WC ? SEGS_C : a.segs, rs = 8 * segs;
    const int inv_segs = WC ? (1 << 20) / SEGS_C + 1 : a.inv_segs;
    const int tasks = (WC ? WC + 1 + 2 * LK_MARGIN : a.rh) * segs;
#define LK_ROW(r) (REG ? (r) : min((r), ny))
    for (int level = top; level >= 0; --level) {
        STAMP_BEGIN();
        const VofeLevel A = LA[level];
        const VofeLevel B = LB[level];
        const float sc = level_scale(level);
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
        STAMP_END(8);
        // prefetch the search region around the start position (loads only; written to LDS after the
        // template window has been computed, so both memory latencies overlap)
        const uint8_t* J0 = B.I + (long long)slotB * B.slot_stride;
        bool staged = false;
        int rx0 = 0, ry0 = 0;
        StageRegs pre;
        {
            const int sx = floor_i(npx - hx), sy = floor_i(npy - hy);
            if (!(sx < -w || sx >= B.w || sy < -h || sy >= B.h)) {
                rx0 = sx - LK_MARGIN; ry0 = sy - LK_MARGIN;
                staged = true;
            }
        }
        // ---- template window: I, Ix, Iy at the fixed prev position -> registers
        constexpr int RP = (R + 1) / 2;              // rows in pairs, two per register (see shr_into_high_half)
        int Cval[R];                                  // 256 - 512 * I: the seed that turns the bilinear J sample into J - I
        unsigned IXp[RP], IYp[RP];                    // (Ix[2q] | Ix[2q+1] << 16), (Iy[2q] | Iy[2q+1] << 16)
        float fA11, fA12, fA22;
        {
            const Wts q = bilinear_weights(ppx, ppy, ipx, ipy);
            // uniform bases at the window's top-left pixel; lane offsets are (y0 + r) * pitch + x >= 0
            const long long base = (long long)slotA * A.slot_stride + (long long)ipy * A.pitch + ipx;
            const gbase_t Ib = (gbase_t)(A.I + base);
            const gbase_t Db = (gbase_t)(A.D + base);
            const unsigned off0 = (unsigned)(y0 * A.pitch + x);
            unsigned pI[R + 1];
            unsigned pX[R + 1], pY[R + 1];     // (Ix[x] | Ix[x+1] << 16), (Iy[x] | Iy[x+1] << 16)
            if (staged) stage_load<WIDE>(pre, J0, B, w, h, rx0, ry0, segs, inv_segs, tasks, 0, lane);
#pragma unroll
            for (int r = 0; r <= R; ++r) {
                const unsigned off = off0 + (unsigned)(LK_ROW(r) * A.pitch);
                const unsigned two = gload_u16(Ib, off);
                const uint2 dd = gload_u64(Db, off * 4u);
                pI[r] = __builtin_amdgcn_perm(0u, two, 0x0c010c00u);
                pX[r] = __builtin_amdgcn_perm(dd.y, dd.x, 0x05040100u);
                pY[r] = __builtin_amdgcn_perm(dd.y, dd.x, 0x07060302u);
            }
            int a11 = 0, a12 = 0, a22 = 0;
            // rows a lane does not own must contribute Ix = Iy = 0.  With the regular mapping (WC) a lane owns
            // all R rows or none, so zero weights do it: (0 + 8192) >> 14 == 0.
            const unsigned dw0 = (REG && ny == 0) ? 0u : q.w0, dw1 = (REG && ny == 0) ? 0u : q.w1;
            const int fourteen = 14;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int iv = sdot2(pI[k], q.w0, sdot2(pI[k + 1], q.w1, 256)) >> 9;
                Cval[k] = 256 - (iv << 9);
            }
#pragma unroll
            for (int qd = 0; qd < RP; ++qd) {
                const int k = 2 * qd;
                int xp = sdot2(pX[k], dw0, sdot2(pX[k + 1], dw1, 8192)) >> 14;      // |Ix|, |Iy| <= 4080: half a register each
                int yp = sdot2(pY[k], dw0, sdot2(pY[k + 1], dw1, 8192)) >> 14;
                if (!REG && k >= ny) { xp = 0; yp = 0; }
                if (k + 1 < R && (REG || k + 1 < ny)) {
                    xp = shr_into_high_half(xp, sdot2(pX[k + 1], dw0, sdot2(pX[k + 2], dw1, 8192)), fourteen);
                    yp = shr_into_high_half(yp, sdot2(pY[k + 1], dw0, sdot2(pY[k + 2], dw1, 8192)), fourteen);
                } else {
                    xp &= 0xffff; yp &= 0xffff;            // no row above this one (or not this lane's)
                }
                IXp[qd] = (unsigned)xp; IYp[qd] = (unsigned)yp;
                a11 = sdot2((unsigned)xp, (unsigned)xp, a11); a12 = sdot2((unsigned)xp, (unsigned)yp, a12);
                a22 = sdot2((unsigned)yp, (unsigned)yp, a22);
            }
            if (partials_fit_tensor(a11, a12, a22)) {
                fA11 = (float)wave_sum_small(a11); fA12 = (float)wave_sum_small(a12); fA22 = (float)wave_sum_small(a22);
            } else {
                fA11 = wave_sum_to_float(a11); fA12 = wave_sum_to_float(a12); fA22 = wave_sum_to_float(a22);
            }
        }
        STAMP_END(0);
        const float FLT_SCALE = 1.f / (float)(1 << 20);
        const float A11 = fA11 * FLT_SCALE, A12 = fA12 * FLT_SCALE, A22 = fA22 * FLT_SCALE;
        float Dd = A11 * A22 - A12 * A12;
        // (threshold test on the numerator, division only where the quotient is an output: see LKArgs::min_eig_num)
        const float eigNum = A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12);
        if (flags & 8) err = eigNum;               // (the numerator: divided once, behind the level loop)
        if (eigNum < a.min_eig_num || Dd < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        Dd = rcp_exact_normal(Dd);
        const float Dds = Dd * FLT_SCALE;          // 2^-20 commutes with every rounding below
        STAMP_END(1);
        npx -= hx; npy -= hy;
        float pdx = 0.f, pdy = 0.f;
        if (staged) {
            __syncthreads();
            stage_write(pre, lds, segs, inv_segs, tasks, 0, lane);
            stage_rest<WIDE>(lds, J0, B, w, h, rx0, ry0, segs, inv_segs, tasks, 64 * ST_BATCH, lane);
            __syncthreads();
        }
        STAMP_END(2);
        const int nine = 9;
        // fast position test while the staged region is "safe" (no window position inside it leaves the image) and the
        // position is no NaN (one can only enter at the start of a level): being inside the region implies the bounds test
        bool fast = staged && rx0 >= -w && rx0 + 2 * LK_MARGIN < B.w && ry0 >= -h && ry0 + 2 * LK_MARGIN < B.h &&
                    !__builtin_isunordered(npx, npy);
        for (int j = 0; j < a.max_count; ++j) {
            const float flx = __builtin_floorf(npx), fly = __builtin_floorf(npy);
            int inx = (int)flx, iny = (int)fly;
            if (!(fast && (unsigned)(inx - rx0) <= (unsigned)(2 * LK_MARGIN) && (unsigned)(iny - ry0) <= (unsigned)(2 * LK_MARGIN))) {
                inx = floor_i(npx); iny = floor_i(npy);
                if (inx < -w || inx >= B.w || iny < -h || iny >= B.h) {
                    if (level == 0) status = 0;
                    break;
                }
                if (!staged || inx < rx0 || inx > rx0 + 2 * LK_MARGIN || iny < ry0 || iny > ry0 + 2 * LK_MARGIN) {
                    __syncthreads();
                    rx0 = inx - LK_MARGIN; ry0 = iny - LK_MARGIN;
                    stage_rest<WIDE>(lds, J0, B, w, h, rx0, ry0, segs, inv_segs, tasks, 0, lane);
                    staged = true;
                    __syncthreads();
                    fast = rx0 >= -w && rx0 + 2 * LK_MARGIN < B.w && ry0 >= -h && ry0 + 2 * LK_MARGIN < B.h;
                }
            }
            ++n_it;
            STAMP_END(6);
            const Wts q = bilinear_weights_frac(npx - flx, npy - fly);
            const unsigned* lp = lds + (iny - ry0 + y0) * rs + (inx - rx0) + x;
            unsigned P[R + 1];
#pragma unroll
            for (int r = 0; r <= R; ++r) P[r] = lp[LK_ROW(r) * rs];
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int qd = 0; qd < RP; ++qd) {
                const int k = 2 * qd;
                int dp = sdot2(P[k], q.w0, sdot2(P[k + 1], q.w1, Cval[k])) >> 9;          // J - I, |.| <= 8160: half a register
                if (k + 1 < R) dp = shr_into_high_half(dp, sdot2(P[k + 1], q.w0, sdot2(P[k + 2], q.w1, Cval[k + 1])), nine);
                // (rows without a partner, or not this lane's, meet a zero half in IXp / IYp)
                b1 = sdot2((unsigned)dp, IXp[qd], b1);
                b2 = sdot2((unsigned)dp, IYp[qd], b2);
            }
            STAMP_END(3);
            float fb1, fb2;
            if (partials_fit2(b1, b2)) { fb1 = (float)wave_sum_small(b1); fb2 = (float)wave_sum_small(b2); }
            else { fb1 = wave_sum_to_float(b1); fb2 = wave_sum_to_float(b2); }
            STAMP_END(4);
            const float dx = (A12 * fb2 - A22 * fb1) * Dds;
            const float dy = (A12 * fb1 - A11 * fb2) * Dds;
            npx += dx; npy += dy;
            outx = npx + hx; outy = npy + hy;
            STAMP_END(5);
            // delta.ddot(delta) <= eps^2 is defined in double; an f32 pre-test (relative error < 2e-7)
            // decides all but the borderline cases, which take the exact path
            const float d2 = dx * dx + dy * dy;
            bool conv = d2 < a.eps2_lo;
            if (__any(!conv && !(d2 > a.eps2_hi)))          // ballot: keeps the f64 sequence out of the common path
                conv = (double)dx * (double)dx + (double)dy * (double)dy <= a.eps2;
            if (conv) break;
            // |(double)(float)(dx+pdx)| < 0.01  <=>  |f32 value| <= 0.01f (the largest float below 0.01)
            if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        STAMP_BEGIN();
        if (status && level == 0 && !(flags & 8) && !(flags & LK_NO_ERR)) {
            const float ex = outx - hx, ey = outy - hy;
            const int inx = floor_i(ex), iny = floor_i(ey);
            if (inx < -w || inx >= B.w || iny < -h || iny >= B.h) { status = 0; continue; }
            if (!staged || inx < rx0 || inx > rx0 + 2 * LK_MARGIN || iny < ry0 || iny > ry0 + 2 * LK_MARGIN) {
                __syncthreads();
                rx0 = inx - LK_MARGIN; ry0 = iny - LK_MARGIN;
                stage_rest<WIDE>(lds, J0, B, w, h, rx0, ry0, segs, inv_segs, tasks, 0, lane);
                staged = true;
                __syncthreads();
            }
            const Wts q = bilinear_weights(ex, ey, inx, iny);
            const unsigned* lp = lds + (iny - ry0 + y0) * rs + (inx - rx0) + x;
            unsigned P[R + 1];
#pragma unroll
            for (int r = 0; r <= R; ++r) P[r] = lp[LK_ROW(r) * rs];
            int se = 0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int diff = sdot2(P[k], q.w0, sdot2(P[k + 1], q.w1, Cval[k])) >> 9;
                se += (REG || k < ny) ? abs(diff) : 0;
            }
            if (REG && ny == 0) se = 0;
            err = wave_sum_to_float(se) / (float)(32 * w * h);
        }
        __syncthreads();     // all LDS reads of this level done before the next level restages
        STAMP_END(9);
    }
    // OPTFLOW_LK_GET_MIN_EIGENVALS: err holds the numerator of the last level that reached the test (or the 0 of a level-0
    // window outside the image); the one division it needs happens here instead of at every level
    if (flags & 8) err = err / (float)(2 * w * h);
}

#undef LK_ROW
// Register budget decides the occupancy (21x21: 93 VGPRs -> 5 waves per SIMD).  Capping the allocation
// further (amdgpu_waves_per_eu) was measured: the spills cost more than the occupancy gives.
template <int R, int WC>
__global__ __launch_bounds__(64) void lk_track_kernel(LKArgs a) {
    extern __shared__ unsigned lds[];
#ifdef LK_STAMPS
    const unsigned long long st_wave0 = __builtin_amdgcn_s_memtime();
#endif
    const int i = blockIdx.x;
    const int lane = threadIdx.x;
    const int ww = WC ? WC : a.win_w, wh = WC ? WC : a.win_h;
    const int g = lane / ww;
    const int x = lane - g * ww;
    constexpr bool REG = WC != 0 && WC % R == 0;
    const int rpg = WC ? R : a.rows_per_group;
    const bool active = g < (WC ? (WC + R - 1) / R : a.groups);
    const int y0 = REG ? (active ? g * R : 0) : min(g * rpg, wh);
    const int ny = REG ? (active ? R : 0) : (active ? min(rpg, wh - y0) : 0);

    const float2 p = a.prev_pts[i];
    int idx = a.img_idx ? a.img_idx[i] : 0;
    const bool bad_idx = (unsigned)idx >= (unsigned)a.n_images;      // wave-uniform: never address a slot outside the batch
    if (bad_idx) idx = 0;
    float fx = 0.f, fy = 0.f;
    if (a.flags & 4) { const float2 q = a.next_pts[i]; fx = q.x; fy = q.y; }
    int st = bad_idx ? 0 : 1;
    float er = 0.f;
    float bx = 0.f, by = 0.f;
    const int ndir = bad_idx ? 0 : (a.fb ? 2 : 1);
    unsigned n_lv = 0, n_it = 0;
    unsigned st_acc[STAMP_N] = {};
#ifdef LK_STAMPS
    __builtin_amdgcn_s_waitcnt(0);                                               // the point and its image index have arrived
    st_acc[10] = (unsigned)(__builtin_amdgcn_s_memtime() - st_wave0);                        // kernel prologue
#endif
    for (int dir = 0; dir < ndir; ++dir) {      // dir 1 = backward pass from the forward result
        const VofeLevel* LA = dir ? a.lv_b : a.lv_a;
        const VofeLevel* LB = dir ? a.lv_a : a.lv_b;
        const int sa = (dir ? a.slot_b : a.slot_a) + idx, sb = (dir ? a.slot_a : a.slot_b) + idx;
        float qx = dir ? fx : p.x, qy = dir ? fy : p.y;
        float ox = dir ? 0.f : fx, oy = dir ? 0.f : fy, e = 0.f;
        int s = 1;
        lk_one<R, WC>(a, LA, LB, sa, sb, qx, qy, ox, oy, s, e, dir ? LK_NO_ERR : a.flags, lds, lane, x, y0, ny, n_lv, n_it, st_acc);
        if (dir == 0) { fx = ox; fy = oy; st = s; er = e; }
        else { bx = ox; by = oy; }
        // a keypoint lost by the forward pass stays lost whatever the backward pass finds (continuous.py:158 only
        // clears status bits) and none of its backward results are returned: skip that pass (wave-uniform)
        if (!st) break;
    }
    if (a.fb) {
        const float dx = fabsf(p.x - bx), dy = fabsf(p.y - by);
        float d = dx > dy ? dx : dy;
        if (dx != dx || dy != dy) d = __builtin_nanf("");
        st = st & (d < a.fb_thresh ? 1 : 0);
    }
    if (bad_idx) { st = 0; er = 0.f; }
    if (lane == 0) {
        a.next_pts[i] = make_float2(fx, fy);
        a.status[i] = (uint8_t)st;
        a.err[i] = er;
        if (a.counters) {
            atomicAdd(a.counters + 0, 1ull);
            atomicAdd(a.counters + 1, (unsigned long long)n_lv);
            atomicAdd(a.counters + 2, (unsigned long long)n_it);
#ifdef LK_STAMPS
            st_acc[7] = (unsigned)(__builtin_amdgcn_s_memtime() - st_wave0);            // whole wavefront
            if (blockIdx.x < 256) for (int k = 0; k < STAMP_N; ++k) a.counters[8 + STAMP_N * blockIdx.x + k] = st_acc[k];
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Level-parallel kernel for SMALL launches (the reference's real load: a few hundred landmarks and candidates per frame).
// Such a launch leaves most SIMDs with one wavefront or none, and lasts as long as one keypoint's dependent chain:
// 2 directions x n_levels x (template set-up + Gauss-Newton iterations).  Only the iterations depend on the coarser
// level's result; the template window, the structure tensor and the minimum-eigenvalue test of a level depend on the
// fixed previous point alone.  So a keypoint gets one wavefront PER LEVEL (a workgroup of n_levels waves): all levels'
// set-ups run at the same time, then the levels iterate in turn, coarse to fine, handing the position over through
// LDS (one workgroup barrier per level).  The chain per direction shrinks from n_levels set-ups + iterations to one
// set-up + iterations.  Each wave also stages a search region ahead of time around where the walk will most likely
// start (forward: zero flow; backward: the original point, where a successful track must return to); the region is a
// cache of pixel pairs, a wrong guess only costs the re-staging the iteration loop does anyway.  Arithmetic per level
// is lk_one's, statement for statement, so results are bit-identical.  Compile-time windows only (21x21, 11x11).
__device__ __forceinline__ void wave_lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

constexpr int LK_LP_MAX_LEVELS = 4;
constexpr int LK_LP_HAND = 16;              // LDS words in front of the regions: hand-over position, status

template <int R, int WC>
__device__ __forceinline__ void lk_lp_dir(const LKArgs& a, const VofeLevel* __restrict__ LA, const VofeLevel* __restrict__ LB,
                                          int slotA, int slotB, float px, float py, float gx, float gy, float& outx, float& outy,
                                          int& status, float& err, int flags, unsigned* lds, float* hand, int level, int lane,
                                          int x, int y0, int ny, unsigned& n_lv, unsigned& n_it) {
    constexpr bool REG = WC % R == 0;
    constexpr bool WIDE = WC == 21;
    constexpr int w = WC, h = WC;
    const float hx = (float)(w - 1) * 0.5f, hy = (float)(h - 1) * 0.5f;
    const int top = a.n_levels - 1;
    constexpr int segs = (WC + 1 + 2 * LK_MARGIN + 7) / 8, rs = 8 * segs;
    constexpr int inv_segs = (1 << 20) / segs + 1;
    constexpr int tasks = (WC + 1 + 2 * LK_MARGIN) * segs;
#define LK_ROW(r) (REG ? (r) : min((r), ny))
    // ---- this level's fixed part (all levels at once) ----------------------------------------------------------
    const VofeLevel A = LA[level];
    const VofeLevel B = LB[level];
    const float sc = level_scale(level);
    float ppx = px * sc, ppy = py * sc;
    const float tpx = ppx, tpy = ppy;                 // the walk's start at the top level
    ppx -= hx; ppy -= hy;
    const int ipx = floor_i(ppx), ipy = floor_i(ppy);
    const bool in_bounds = !(ipx < -w || ipx >= A.w || ipy < -h || ipy >= A.h);
    const uint8_t* J0 = B.I + (long long)slotB * B.slot_stride;
    bool staged = false, eig_ok = false;
    int rx0 = 0, ry0 = 0;
    constexpr int RP = (R + 1) / 2;                  // rows in pairs, two per register (see shr_into_high_half)
    int Cval[R];
    unsigned IXp[RP], IYp[RP];
    const int nine = 9, fourteen = 14;
    float A11 = 0.f, A12 = 0.f, A22 = 0.f, Dd = 0.f;
    if (in_bounds) {
        StageRegs pre;
        {
            const int sx = floor_i(gx * sc - hx), sy = floor_i(gy * sc - hy);
            if (!(sx < -w || sx >= B.w || sy < -h || sy >= B.h)) {
                rx0 = sx - LK_MARGIN; ry0 = sy - LK_MARGIN;
                staged = true;
            }
        }
        float fA11, fA12, fA22;
        {
            const Wts q = bilinear_weights(ppx, ppy, ipx, ipy);
            const long long base = (long long)slotA * A.slot_stride + (long long)ipy * A.pitch + ipx;
            const gbase_t Ib = (gbase_t)(A.I + base);
            const gbase_t Db = (gbase_t)(A.D + base);
            const unsigned off0 = (unsigned)(y0 * A.pitch + x);
            unsigned pI[R + 1];
            unsigned pX[R + 1], pY[R + 1];
            if (staged) stage_load<WIDE>(pre, J0, B, w, h, rx0, ry0, segs, inv_segs, tasks, 0, lane);
#pragma unroll
            for (int r = 0; r <= R; ++r) {
                const unsigned off = off0 + (unsigned)(LK_ROW(r) * A.pitch);
                const unsigned two = gload_u16(Ib, off);
                const uint2 dd = gload_u64(Db, off * 4u);
                pI[r] = __builtin_amdgcn_perm(0u, two, 0x0c010c00u);
                pX[r] = __builtin_amdgcn_perm(dd.y, dd.x, 0x05040100u);
                pY[r] = __builtin_amdgcn_perm(dd.y, dd.x, 0x07060302u);
            }
            int a11 = 0, a12 = 0, a22 = 0;
            const unsigned dw0 = (REG && ny == 0) ? 0u : q.w0, dw1 = (REG && ny == 0) ? 0u : q.w1;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int iv = sdot2(pI[k], q.w0, sdot2(pI[k + 1], q.w1, 256)) >> 9;
                Cval[k] = 256 - (iv << 9);
            }
#pragma unroll
            for (int qd = 0; qd < RP; ++qd) {
                const int k = 2 * qd;
                int xp = sdot2(pX[k], dw0, sdot2(pX[k + 1], dw1, 8192)) >> 14;
                int yp = sdot2(pY[k], dw0, sdot2(pY[k + 1], dw1, 8192)) >> 14;
                if (!REG && k >= ny) { xp = 0; yp = 0; }
                if (k + 1 < R && (REG || k + 1 < ny)) {
                    xp = shr_into_high_half(xp, sdot2(pX[k + 1], dw0, sdot2(pX[k + 2], dw1, 8192)), fourteen);
                    yp = shr_into_high_half(yp, sdot2(pY[k + 1], dw0, sdot2(pY[k + 2], dw1, 8192)), fourteen);
                } else {
                    xp &= 0xffff; yp &= 0xffff;
                }
                IXp[qd] = (unsigned)xp; IYp[qd] = (unsigned)yp;
                a11 = sdot2((unsigned)xp, (unsigned)xp, a11); a12 = sdot2((unsigned)xp, (unsigned)yp, a12);
                a22 = sdot2((unsigned)yp, (unsigned)yp, a22);
            }
            if (partials_fit_tensor(a11, a12, a22)) {
                fA11 = (float)wave_sum_small(a11); fA12 = (float)wave_sum_small(a12); fA22 = (float)wave_sum_small(a22);
            } else {
                fA11 = wave_sum_to_float(a11); fA12 = wave_sum_to_float(a12); fA22 = wave_sum_to_float(a22);
            }
        }
        const float FLT_SCALE0 = 1.f / (float)(1 << 20);
        A11 = fA11 * FLT_SCALE0; A12 = fA12 * FLT_SCALE0; A22 = fA22 * FLT_SCALE0;
        Dd = A11 * A22 - A12 * A12;
        // (threshold test on the numerator: see LKArgs::min_eig_num; this kernel never reports the quotient)
        const float eigNum = A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12);
        eig_ok = !(eigNum < a.min_eig_num || Dd < 1.1920928955078125e-07f);
        Dd = rcp_exact_normal(Dd) * FLT_SCALE0;        // (read only where eig_ok; 2^-20 folded in: it commutes with every rounding of the update)
        if (staged) {                                  // (uniform) the region guessed above; this wave's reads of the other direction are done
            wave_lds_order();
            stage_write(pre, lds, segs, inv_segs, tasks, 0, lane);
            stage_rest<WIDE>(lds, J0, B, w, h, rx0, ry0, segs, inv_segs, tasks, 64 * ST_BATCH, lane);
            wave_lds_order();
        }
    }
    // ---- the levels in turn, coarse to fine ----------------------------------------------------------------------
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    for (int s = top; s >= 0; --s) {
        if (s == level) {
            float npx, npy;
            if (level == top) {
                if (flags & 4) { npx = outx * sc; npy = outy * sc; }
                else { npx = tpx; npy = tpy; }
            } else { npx = hand[0] * 2.f; npy = hand[1] * 2.f; }
            float ox = npx, oy = npy;
            if (!in_bounds) {
                if (level == 0) { status = 0; err = 0.f; }
            } else {
                ++n_lv;
                if (!eig_ok) {
                    if (level == 0) status = 0;
                } else {
                    npx -= hx; npy -= hy;
                    float pdx = 0.f, pdy = 0.f;
                    bool lost = false;
                    // fast position test, as in lk_one: inside a "safe" staged region nothing else needs testing
                    bool fast = staged && rx0 >= -w && rx0 + 2 * LK_MARGIN < B.w && ry0 >= -h && ry0 + 2 * LK_MARGIN < B.h &&
                                !__builtin_isunordered(npx, npy);
                    for (int j = 0; j < a.max_count; ++j) {
                        const float flx = __builtin_floorf(npx), fly = __builtin_floorf(npy);
                        int inx = (int)flx, iny = (int)fly;
                        if (!(fast && (unsigned)(inx - rx0) <= (unsigned)(2 * LK_MARGIN) && (unsigned)(iny - ry0) <= (unsigned)(2 * LK_MARGIN))) {
                            inx = floor_i(npx); iny = floor_i(npy);
                            if (inx < -w || inx >= B.w || iny < -h || iny >= B.h) {
                                if (level == 0) status = 0;
                                lost = true;
                                break;
                            }
                            if (!staged || inx < rx0 || inx > rx0 + 2 * LK_MARGIN || iny < ry0 || iny > ry0 + 2 * LK_MARGIN) {
                                wave_lds_order();
                                rx0 = inx - LK_MARGIN; ry0 = iny - LK_MARGIN;
                                stage_rest<WIDE>(lds, J0, B, w, h, rx0, ry0, segs, inv_segs, tasks, 0, lane);
                                staged = true;
                                wave_lds_order();
                                fast = rx0 >= -w && rx0 + 2 * LK_MARGIN < B.w && ry0 >= -h && ry0 + 2 * LK_MARGIN < B.h;
                            }
                        }
                        ++n_it;
                        const Wts q = bilinear_weights_frac(npx - flx, npy - fly);
                        const unsigned* lp = lds + (iny - ry0 + y0) * rs + (inx - rx0) + x;
                        unsigned P[R + 1];
#pragma unroll
                        for (int r = 0; r <= R; ++r) P[r] = lp[LK_ROW(r) * rs];
                        int b1 = 0, b2 = 0;
#pragma unroll
                        for (int qd = 0; qd < RP; ++qd) {
                            const int k = 2 * qd;
                            int dp = sdot2(P[k], q.w0, sdot2(P[k + 1], q.w1, Cval[k])) >> 9;
                            if (k + 1 < R) dp = shr_into_high_half(dp, sdot2(P[k + 1], q.w0, sdot2(P[k + 2], q.w1, Cval[k + 1])), nine);
                            b1 = sdot2((unsigned)dp, IXp[qd], b1);
                            b2 = sdot2((unsigned)dp, IYp[qd], b2);
                        }
                        float fb1, fb2;
                        if (partials_fit2(b1, b2)) { fb1 = (float)wave_sum_small(b1); fb2 = (float)wave_sum_small(b2); }
                        else { fb1 = wave_sum_to_float(b1); fb2 = wave_sum_to_float(b2); }
                        const float dx = (A12 * fb2 - A22 * fb1) * Dd;
                        const float dy = (A12 * fb1 - A11 * fb2) * Dd;
                        npx += dx; npy += dy;
                        ox = npx + hx; oy = npy + hy;
                        const float d2 = dx * dx + dy * dy;
                        bool conv = d2 < a.eps2_lo;
                        if (__any(!conv && !(d2 > a.eps2_hi)))
                            conv = (double)dx * (double)dx + (double)dy * (double)dy <= a.eps2;
                        if (conv) break;
                        if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                            ox -= dx * 0.5f; oy -= dy * 0.5f;
                            break;
                        }
                        pdx = dx; pdy = dy;
                    }
                    (void)lost;
                    if (status && level == 0 && !(flags & LK_NO_ERR)) {
                        const float ex = ox - hx, ey = oy - hy;
                        const int inx = floor_i(ex), iny = floor_i(ey);
                        if (inx < -w || inx >= B.w || iny < -h || iny >= B.h) {
                            status = 0;
                        } else {
                            if (!staged || inx < rx0 || inx > rx0 + 2 * LK_MARGIN || iny < ry0 || iny > ry0 + 2 * LK_MARGIN) {
                                wave_lds_order();
                                rx0 = inx - LK_MARGIN; ry0 = iny - LK_MARGIN;
                                stage_rest<WIDE>(lds, J0, B, w, h, rx0, ry0, segs, inv_segs, tasks, 0, lane);
                                staged = true;
                                wave_lds_order();
                            }
                            const Wts q = bilinear_weights(ex, ey, inx, iny);
                            const unsigned* lp = lds + (iny - ry0 + y0) * rs + (inx - rx0) + x;
                            unsigned P[R + 1];
#pragma unroll
                            for (int r = 0; r <= R; ++r) P[r] = lp[LK_ROW(r) * rs];
                            int se = 0;
#pragma unroll
                            for (int k = 0; k < R; ++k) {
                                const int diff = sdot2(P[k], q.w0, sdot2(P[k + 1], q.w1, Cval[k])) >> 9;
                                se += (REG || k < ny) ? abs(diff) : 0;
                            }
                            if (REG && ny == 0) se = 0;
                            err = wave_sum_to_float(se) / (float)(32 * w * h);
                        }
                    }
                }
            }
            outx = ox; outy = oy;
            if (lane == 0) {
                hand[0] = ox; hand[1] = oy;
                if (level == 0) hand[2] = status ? 1.f : 0.f;
            }
        }
        __syncthreads();          // every wave of the keypoint, every level: the hand-over
    }
#undef LK_ROW
}

template <int R, int WC>
__global__ __launch_bounds__(64 * LK_LP_MAX_LEVELS) void lk_track_kernel_lp(LKArgs a) {
    extern __shared__ unsigned lds[];
    constexpr int REGION = (WC + 1 + 2 * LK_MARGIN) * 8 * ((WC + 1 + 2 * LK_MARGIN + 7) / 8);
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, level = threadIdx.x >> 6;      // one wave per pyramid level (blockDim = 64 * n_levels)
    float* hand = reinterpret_cast<float*>(lds);
    unsigned* region = lds + LK_LP_HAND + level * REGION;
    const int g = lane / WC;
    const int x = lane - g * WC;
    constexpr bool REG = WC % R == 0;
    const bool active = g < (WC + R - 1) / R;
    const int y0 = REG ? (active ? g * R : 0) : min(g * R, WC);
    const int ny = REG ? (active ? R : 0) : (active ? min(R, WC - y0) : 0);

    const float2 p = a.prev_pts[i];
    int idx = a.img_idx ? a.img_idx[i] : 0;
    const bool bad_idx = (unsigned)idx >= (unsigned)a.n_images;      // block-uniform
    if (bad_idx) idx = 0;
    float fx = 0.f, fy = 0.f;
    if (a.flags & 4) { const float2 q = a.next_pts[i]; fx = q.x; fy = q.y; }
    int st = bad_idx ? 0 : 1;
    float er = 0.f;
    float bx = 0.f, by = 0.f;
    const int ndir = bad_idx ? 0 : (a.fb ? 2 : 1);
    unsigned n_lv = 0, n_it = 0;
    for (int dir = 0; dir < ndir; ++dir) {
        const VofeLevel* LA = dir ? a.lv_b : a.lv_a;
        const VofeLevel* LB = dir ? a.lv_a : a.lv_b;
        const int sa = (dir ? a.slot_b : a.slot_a) + idx, sb = (dir ? a.slot_a : a.slot_b) + idx;
        const float qx = dir ? fx : p.x, qy = dir ? fy : p.y;
        // where the walk is expected to start at level 0: the initial guess going forward, the original point coming back
        const float gx = dir ? p.x : ((a.flags & 4) ? fx : p.x), gy = dir ? p.y : ((a.flags & 4) ? fy : p.y);
        float ox = dir ? 0.f : fx, oy = dir ? 0.f : fy, e = 0.f;
        int s = 1;
        lk_lp_dir<R, WC>(a, LA, LB, sa, sb, qx, qy, gx, gy, ox, oy, s, e, dir ? LK_NO_ERR : a.flags, region, hand, level, lane, x, y0,
                         ny, n_lv, n_it);
        // the level-0 wave left the result in the hand-over words (its barrier has passed): every wave takes it from there
        const float rx = hand[0], ry = hand[1];
        const int rs_ = hand[2] != 0.f ? 1 : 0;
        if (dir == 0) { fx = rx; fy = ry; st = rs_; er = e; }
        else { bx = rx; by = ry; }
        __syncthreads();                      // hand-over words read by all before the next direction rewrites them
        if (!st) break;                       // block-uniform
    }
    if (a.fb) {
        const float dx = fabsf(p.x - bx), dy = fabsf(p.y - by);
        float d = dx > dy ? dx : dy;
        if (dx != dx || dy != dy) d = __builtin_nanf("");
        st = st & (d < a.fb_thresh ? 1 : 0);
    }
    if (bad_idx) { st = 0; er = 0.f; }
    if (lane == 0) {
        if (level == 0) {
            a.next_pts[i] = make_float2(fx, fy);
            a.status[i] = (uint8_t)st;
            a.err[i] = er;
        }
        if (a.counters) {
            if (level == 0) atomicAdd(a.counters + 0, 1ull);
            atomicAdd(a.counters + 1, (unsigned long long)n_lv);
            atomicAdd(a.counters + 2, (unsigned long long)n_it);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two keypoints per wavefront (21x21 windows, large launches).  The single-point kernel spends more
// than half of its VALU instructions on work that is uniform across the wave (bilinear weights, the
// 2x2 solve, convergence tests, bounds, addressing); a wave64 instruction costs 4 SIMD cycles whether
// it serves one keypoint or two.  Here each half of the wave (32 lanes) owns one keypoint; a lane
// owns two 7-row strips of the window (strip s = lane32 + 32 q, column s % 21, rows 7 (s / 21) .. +6,
// 63 strips in all), each half has its own LDS search region, and everything that was wave-uniform
// becomes half-uniform lane data.  The halves diverge freely (different level outcomes, iteration
// counts, restaging): the compiler's EXEC masking keeps that correct, and an exhausted half only idles.
// Arithmetic is identical to lk_one, so results stay bit-identical.
namespace x2 {
constexpr int W = 21, R = 7;
constexpr int RW = W + 1 + 2 * LK_MARGIN, SEGS = (RW + 7) / 8, RS = 8 * SEGS, RH = W + 1 + 2 * LK_MARGIN;
constexpr int TASKS = RH * SEGS, REGION = RH * RS;
static_assert(SEGS == 4, "task -> (row, seg) uses shifts");
// PPW keypoints per wavefront (2 or 4): a group of GL = 64 / PPW lanes owns one keypoint, a lane owns NS strips
template <int PPW> struct Cfg {
    static constexpr int GL = 64 / PPW, NS = (3 * W + GL - 1) / GL, TB = (TASKS + GL - 1) / GL;
};

__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

constexpr int RP = (R + 1) / 2;          // register pairs per strip (the last one holds a single row)

// exact sum over the lanes of a group (one or two DPP rows), result in every lane of the group
template <int PPW>
__device__ __forceinline__ int half_sum(int v, int partner_addr) {
    v = row_sum16(v);
    if (PPW == 2) v += __builtin_amdgcn_ds_bpermute(partner_addr, v);
    return v;
}
template <int PPW>
__device__ __forceinline__ float half_sum_to_float(int v, int partner_addr) {
    const int lo = half_sum<PPW>(v & 0xffff, partner_addr);
    const int hi = half_sum<PPW>(v >> 16, partner_addr);
    return (float)((long long)hi * 65536 + (long long)lo);
}

// Both sums of an iteration in ONE reduction: lanes 0-7 of every DPP row reduce b1, lanes 8-15 reduce b2 (`hi8` = lane & 8).
// own = the group's total of the sum this lane's half-row reduced, other = the total of the other sum.  Integer adds: exact in
// any order, same totals as two half_sum calls; 8 instructions instead of 12.
template <int PPW>
__device__ __forceinline__ void half_sum_pair(int b1, int b2, bool hi8, int partner_addr, int& own, int& other) {
    const int m = hi8 ? b2 : b1, m2 = hi8 ? b1 : b2;
    int z = m + __builtin_amdgcn_update_dpp(0, m2, 0x128, 0xF, 0xF, false);      // row_ror:8: the other half-row's lane of the same sum
    z += __builtin_amdgcn_update_dpp(0, z, 0x141, 0xF, 0xF, false);              // row_half_mirror
    z += __builtin_amdgcn_update_dpp(0, z, 0xB1, 0xF, 0xF, false);               // quad_perm:[1,0,3,2]
    z += __builtin_amdgcn_update_dpp(0, z, 0x4E, 0xF, 0xF, false);               // quad_perm:[2,3,0,1]
    if (PPW == 2) z += __builtin_amdgcn_ds_bpermute(partner_addr, z);
    own = z;
    other = __builtin_amdgcn_update_dpp(0, z, 0x128, 0xF, 0xF, false);
}

template <int PPW> struct Stage { uint2 v[Cfg<PPW>::TB]; unsigned nb[Cfg<PPW>::TB]; };
// Addresses are a wave-uniform base (the plane's first reachable byte: row -W, column -PADX of slot 0) plus an
// unsigned 32-bit lane offset that includes the half's slot, so loads keep the saddr form although the two
// halves read different images.  The host takes this kernel only when a whole plane set fits 32-bit offsets.
constexpr int PADX = W + 8;
__device__ __forceinline__ gbase_t plane_base(const uint8_t* I0, int pitch) { return (gbase_t)(I0 - (long long)W * pitch - PADX); }
template <int PPW>
__device__ __forceinline__ void stage_load(Stage<PPW>& s, gbase_t Jb, unsigned slot_off, const VofeLevel& B, int rx0, int ry0, int l32) {
    const int ymin = -W, ymax = B.h + W - 1;
#pragma unroll
    for (int k = 0; k < Cfg<PPW>::TB; ++k) {
        const int t = k * Cfg<PPW>::GL + l32;
        const int row = t >> 2, seg = t & 3;
        s.v[k] = make_uint2(0u, 0u);
        s.nb[k] = 0u;
        if (t < TASKS) {
            const int gy = min(max(ry0 + row, ymin), ymax);
            // (rows and pitch are far below 2^24: v_mad_u32_u24 instead of the 64-bit multiply-add)
            const unsigned off = slot_off + __umul24((unsigned)(gy + W), (unsigned)B.pitch) + (unsigned)(rx0 + PADX + seg * 8);
            const u32x3 q = gload_u96(Jb, off);
            s.v[k] = make_uint2(q.x, q.y);
            s.nb[k] = q.z;
        }
    }
}
template <int PPW>
__device__ __forceinline__ void stage_write(const Stage<PPW>& s, unsigned* region, int l32) {
#pragma unroll
    for (int k = 0; k < Cfg<PPW>::TB; ++k) {
        const int t = k * Cfg<PPW>::GL + l32;
        if (t < TASKS) {
            const unsigned lo = s.v[k].x, hi = s.v[k].y, nb = s.nb[k];
            uint4 a, b;
            a.x = __builtin_amdgcn_perm(0u, lo, 0x0c010c00u);
            a.y = __builtin_amdgcn_perm(0u, lo, 0x0c020c01u);
            a.z = __builtin_amdgcn_perm(0u, lo, 0x0c030c02u);
            a.w = __builtin_amdgcn_perm(hi, lo, 0x0c040c03u);
            b.x = __builtin_amdgcn_perm(0u, hi, 0x0c010c00u);
            b.y = __builtin_amdgcn_perm(0u, hi, 0x0c020c01u);
            b.z = __builtin_amdgcn_perm(0u, hi, 0x0c030c02u);
            b.w = __builtin_amdgcn_perm(nb, hi, 0x0c040c03u);
            uint4* d = reinterpret_cast<uint4*>(region + t * 8);
            d[0] = a; d[1] = b;
        }
    }
}
template <int PPW>
__device__ __forceinline__ void restage(unsigned* region, gbase_t Jb, unsigned slot_off, const VofeLevel& B, int rx0, int ry0, int l32) {
    Stage<PPW> s;
    lds_order();
    stage_load<PPW>(s, Jb, slot_off, B, rx0, ry0, l32);
    stage_write<PPW>(s, region, l32);
    lds_order();
}

// PRE: the search region's loads are issued before the template window is computed (both latencies overlap inside the
// wavefront) at the price of the 12 staging registers being live across the template set-up -- 150 VGPRs = 3 wavefronts per
// SIMD.  Without it the region is loaded after the minimum-eigenvalue test: 127 VGPRs = 4 wavefronts per SIMD, which hide the
// latency ACROSS wavefronts instead (and a level that fails the test stages nothing).
// CNT: the instance that feeds the optional work counters (vofe_lk_set_counters: point-levels entered, iterations); the
// instance without them carries neither the two registers nor the per-iteration increment.
template <int PPW, bool PRE, bool CNT>
__device__ __forceinline__ void lk_one(const LKArgs& a, const VofeLevel* __restrict__ LA, const VofeLevel* __restrict__ LB,
                                       int slotA, int slotB, float px, float py, float& outx, float& outy,
                                       int& status, float& err, int flags, unsigned* region, int l32, int partner_addr,
                                       const int (&sx)[Cfg<PPW>::NS], const int (&sy0)[Cfg<PPW>::NS],
                                       const bool (&sok)[Cfg<PPW>::NS], unsigned& n_lv, unsigned& n_it) {
    constexpr int NS = Cfg<PPW>::NS;
    constexpr int w = W, h = W;
    const float hx = (float)(w - 1) * 0.5f, hy = (float)(h - 1) * 0.5f;
    const int top = a.n_levels - 1;
    for (int level = top; level >= 0; --level) {
        const VofeLevel A = LA[level];
        const VofeLevel B = LB[level];
        const float sc = level_scale(level);
        float ppx = px * sc, ppy = py * sc;
        float npx, npy;
        if (level == top) {
            if (flags & 4) { npx = outx * sc; npy = outy * sc; }
            else { npx = ppx; npy = ppy; }
        } else { npx = outx * 2.f; npy = outy * 2.f; }
        outx = npx; outy = npy;
        ppx -= hx; ppy -= hy;
        // floor without floor_i's range select: v_cvt_i32_f32 saturates, so a huge coordinate still fails the bounds test, and
        // the one value it maps into the image (NaN -> 0) is caught by the unordered compare
        const int ipx = floor_sat(ppx), ipy = floor_sat(ppy);
        if (__builtin_isunordered(ppx, ppy) || ipx < -w || ipx >= A.w || ipy < -h || ipy >= A.h) {
            if (level == 0) { status = 0; err = 0.f; }
            continue;
        }
        if (CNT) ++n_lv;
        const gbase_t Jb = plane_base(B.I, B.pitch);
        const unsigned jslot = (unsigned)slotB * (unsigned)B.slot_stride;
        bool staged = false;
        int rx0 = 0, ry0 = 0;
        Stage<PPW> pre;
        {
            // (a guess of where the iterations start: a NaN position stages a region nobody reads, the iteration's own test
            // deals with it)
            const int fx = floor_sat(npx - hx), fy = floor_sat(npy - hy);
            if (!(fx < -w || fx >= B.w || fy < -h || fy >= B.h)) {
                rx0 = fx - LK_MARGIN; ry0 = fy - LK_MARGIN;
                staged = true;
            }
        }
        // template window in registers.  Cval = 256 - 512 * I is the accumulator seed of the bilinear J sample, so
        // that (seed + taps) >> 9 is directly diff = J - I ((x - 512 k) >> 9 == (x >> 9) - k for arithmetic
        // shifts).  (Pairing pixels for v_dot2 in the b1 / b2 sums was measured slower: v_dot2 is half rate.)
        // Ix and Iy stay in separate registers: v_mad_i32_i24 (4 cycles) per product instead of an SDWA multiply
        // plus a share of a 3-operand add.
        int Cval[NS][R];
        unsigned IXp[NS][RP], IYp[NS][RP];              // (Ix[2q] | Ix[2q+1] << 16), (Iy[2q] | Iy[2q+1] << 16); upper half 0 past the strip
        float fA11, fA12, fA22;
        {
            const Wts q = bilinear_weights(ppx, ppy, ipx, ipy);
            const int fourteen = 14;
            const gbase_t Ib = plane_base(A.I, A.pitch);
            const gbase_t Db = (gbase_t)((const uint8_t*)A.D - 4 * ((long long)W * A.pitch + PADX));
            const unsigned tbase = (unsigned)slotA * (unsigned)A.slot_stride + (unsigned)((ipy + W) * A.pitch + ipx + PADX);
            if (PRE && staged) stage_load<PPW>(pre, Jb, jslot, B, rx0, ry0, l32);
            int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned off0 = tbase + (unsigned)(sy0[s] * A.pitch + sx[s]);
                unsigned pI[R + 1], pX[R + 1], pY[R + 1];
#pragma unroll
                for (int r = 0; r <= R; ++r) {
                    // (the row step stays in the lane offset: moved into scalar bases -- one s_add pair per load, no VALU -- the
                    // launch was 1 % slower)
                    const unsigned off = off0 + (unsigned)(r * A.pitch);
                    const unsigned two = gload_u16(Ib, off);
                    const uint2 dd = gload_u64(Db, off * 4u);
                    pI[r] = __builtin_amdgcn_perm(0u, two, 0x0c010c00u);
                    pX[r] = __builtin_amdgcn_perm(dd.y, dd.x, 0x05040100u);
                    pY[r] = __builtin_amdgcn_perm(dd.y, dd.x, 0x07060302u);
                }
                // a strip that does not exist (the 64th) contributes Ix = Iy = 0 through zero weights
                const unsigned dw0 = sok[s] ? q.w0 : 0u, dw1 = sok[s] ? q.w1 : 0u;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int iv = sdot2(pI[k], q.w0, sdot2(pI[k + 1], q.w1, 256)) >> 9;
                    Cval[s][k] = 256 - (iv << 9);
                }
#pragma unroll
                for (int qd = 0; qd < RP; ++qd) {
                    // |Ix|, |Iy| <= 4080: two rows per register (the odd row's shift deposits it in the upper half), and the
                    // structure tensor by pairs as well
                    const int k = 2 * qd;
                    int xp = sdot2(pX[k], dw0, sdot2(pX[k + 1], dw1, 8192)) >> 14;
                    int yp = sdot2(pY[k], dw0, sdot2(pY[k + 1], dw1, 8192)) >> 14;
                    if (k + 1 < R) {
                        xp = shr_into_high_half(xp, sdot2(pX[k + 1], dw0, sdot2(pX[k + 2], dw1, 8192)), fourteen);
                        yp = shr_into_high_half(yp, sdot2(pY[k + 1], dw0, sdot2(pY[k + 2], dw1, 8192)), fourteen);
                    } else {
                        xp &= 0xffff; yp &= 0xffff;             // a lone last row: nothing above it
                    }
                    IXp[s][qd] = (unsigned)xp; IYp[s][qd] = (unsigned)yp;
                    a11 = sdot2((unsigned)xp, (unsigned)xp, a11); a12 = sdot2((unsigned)xp, (unsigned)yp, a12);
                    a22 = sdot2((unsigned)yp, (unsigned)yp, a22);
                }
            }
            if (partials_fit_tensor(a11, a12, a22)) {
                fA11 = (float)half_sum<PPW>(a11, partner_addr); fA12 = (float)half_sum<PPW>(a12, partner_addr);
                fA22 = (float)half_sum<PPW>(a22, partner_addr);
            } else {
                fA11 = half_sum_to_float<PPW>(a11, partner_addr); fA12 = half_sum_to_float<PPW>(a12, partner_addr);
                fA22 = half_sum_to_float<PPW>(a22, partner_addr);
            }
        }
        const float FLT_SCALE = 1.f / (float)(1 << 20);
        const float A11 = fA11 * FLT_SCALE, A12 = fA12 * FLT_SCALE, A22 = fA22 * FLT_SCALE;
        float Dd = A11 * A22 - A12 * A12;
        // the threshold test on the numerator (correctly rounded division is monotone; the host found the matching bound), the
        // division itself only where the quotient is an output
        const float eigNum = A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12);
        if (flags & 8) err = eigNum;               // (the numerator: divided once, behind the level loop)
        if (eigNum < a.min_eig_num || Dd < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        Dd = rcp_exact_normal(Dd);
        const float Dds = Dd * FLT_SCALE;
        // lanes 0-7 of a DPP row own the b1 total (dx = A12 other - A22 own, dy = A12 own - A11 other), lanes 8-15 the b2 total
        const bool hi8 = (l32 & 8) != 0;
        const float cxo = hi8 ? A12 : -A22, cxt = hi8 ? -A22 : A12, cyo = hi8 ? -A11 : A12, cyt = hi8 ? A12 : -A11;
        npx -= hx; npy -= hy;
        float pdx = 0.f, pdy = 0.f;
        if (staged) {
            lds_order();
            if (!PRE) stage_load<PPW>(pre, Jb, jslot, B, rx0, ry0, l32);
            stage_write<PPW>(pre, region, l32);
            lds_order();
        }
        const int nine = 9;
        // Fast test of an iteration's window position: while the staged region lies where no window position inside it can
        // leave the image ("safe"), being inside the region (two unsigned compares on the offsets the LDS address needs anyway)
        // implies the bounds test.  A NaN position can only enter at the start of a level (every update below is finite) and
        // takes the full test, like positions near the image border and the first iteration after a level without a region.
        bool fast = staged && rx0 >= -w && rx0 + 2 * LK_MARGIN < B.w && ry0 >= -h && ry0 + 2 * LK_MARGIN < B.h &&
                    !__builtin_isunordered(npx, npy);
        for (int j = 0; j < a.max_count; ++j) {
            const float flx = __builtin_floorf(npx), fly = __builtin_floorf(npy);
            int inx = (int)flx, iny = (int)fly;
            if (!(fast && (unsigned)(inx - rx0) <= (unsigned)(2 * LK_MARGIN) && (unsigned)(iny - ry0) <= (unsigned)(2 * LK_MARGIN))) {
                inx = floor_i(npx); iny = floor_i(npy);
                if (inx < -w || inx >= B.w || iny < -h || iny >= B.h) {
                    if (level == 0) status = 0;
                    break;
                }
                if (!staged || inx < rx0 || inx > rx0 + 2 * LK_MARGIN || iny < ry0 || iny > ry0 + 2 * LK_MARGIN) {
                    rx0 = inx - LK_MARGIN; ry0 = iny - LK_MARGIN;
                    restage<PPW>(region, Jb, jslot, B, rx0, ry0, l32);
                    staged = true;
                    fast = rx0 >= -w && rx0 + 2 * LK_MARGIN < B.w && ry0 >= -h && ry0 + 2 * LK_MARGIN < B.h;
                }
            }
            if (CNT) ++n_it;
            // (float)inx == floor(npx) here (|npx| < 2^24 inside the image), so the fractional parts come from the floors
            const Wts q = bilinear_weights_frac(npx - flx, npy - fly);
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned* lp = region + (iny - ry0 + sy0[s]) * RS + (inx - rx0) + sx[s];
                unsigned P[R + 1];
#pragma unroll
                for (int r = 0; r <= R; ++r) P[r] = lp[r * RS];
#pragma unroll
                for (int qd = 0; qd < RP; ++qd) {
                    const int k = 2 * qd;
                    int dp = sdot2(P[k], q.w0, sdot2(P[k + 1], q.w1, Cval[s][k])) >> 9;        // |diff| <= 8160: a half register
                    if (k + 1 < R) dp = shr_into_high_half(dp, sdot2(P[k + 1], q.w0, sdot2(P[k + 2], q.w1, Cval[s][k + 1])), nine);
                    // (a lone last row leaves its sign extension in the upper half: the template's upper half is 0 there)
                    b1 = sdot2((unsigned)dp, IXp[s][qd], b1);
                    b2 = sdot2((unsigned)dp, IYp[s][qd], b2);
                }
            }
            // FLT_SCALE = 2^-20 commutes with every rounding here, so it is folded into the reciprocal once per level.  The two
            // products of dx (and of dy) are added in either order (a - b == a + (-b), addition commutes), so each lane takes
            // them as own / other with the coefficients its half-row was given above.
            float fo, ft;
            if (partials_fit2(b1, b2)) {
                int own, other;
                half_sum_pair<PPW>(b1, b2, hi8, partner_addr, own, other);
                fo = (float)own; ft = (float)other;
            } else {
                const float fb1 = half_sum_to_float<PPW>(b1, partner_addr), fb2 = half_sum_to_float<PPW>(b2, partner_addr);
                fo = hi8 ? fb2 : fb1; ft = hi8 ? fb1 : fb2;
            }
            const float dx = (cxt * ft + cxo * fo) * Dds;
            const float dy = (cyt * ft + cyo * fo) * Dds;
            npx += dx; npy += dy;
            outx = npx + hx; outy = npy + hy;
            const float d2 = dx * dx + dy * dy;
            bool conv = d2 < a.eps2_lo;
            if (__any(!conv && !(d2 > a.eps2_hi)))          // ballot: keeps the f64 sequence out of the common path
                conv = conv || (double)dx * (double)dx + (double)dy * (double)dy <= a.eps2;
            if (conv) break;
            if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0 && !(flags & 8) && !(flags & LK_NO_ERR)) {
            const float ex = outx - hx, ey = outy - hy;
            const int inx = floor_i(ex), iny = floor_i(ey);
            if (inx < -w || inx >= B.w || iny < -h || iny >= B.h) { status = 0; continue; }
            if (!staged || inx < rx0 || inx > rx0 + 2 * LK_MARGIN || iny < ry0 || iny > ry0 + 2 * LK_MARGIN) {
                rx0 = inx - LK_MARGIN; ry0 = iny - LK_MARGIN;
                restage<PPW>(region, Jb, jslot, B, rx0, ry0, l32);
                staged = true;
            }
            const Wts q = bilinear_weights(ex, ey, inx, iny);
            int se = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned* lp = region + (iny - ry0 + sy0[s]) * RS + (inx - rx0) + sx[s];
                unsigned P[R + 1];
#pragma unroll
                for (int r = 0; r <= R; ++r) P[r] = lp[r * RS];
                int ss = 0;
#pragma unroll
                for (int k = 0; k < R; ++k) ss += abs(sdot2(P[k], q.w0, sdot2(P[k + 1], q.w1, Cval[s][k])) >> 9);
                se += sok[s] ? ss : 0;
            }
            err = half_sum_to_float<PPW>(se, partner_addr) / (float)(32 * w * h);
        }
        lds_order();
    }
    // OPTFLOW_LK_GET_MIN_EIGENVALS: err holds the numerator of the last level that reached the test; one division here
    if (flags & 8) err = err / (float)(2 * w * h);
}
}  // namespace x2

// Diagnostic builds only (tools/diag_w4.sh, never the shipped library): -DLK_X2_WAVES_PER_EU=4 makes the compiler hold the
// kernel at 128 VGPRs by spilling to scratch memory (DESIGN.md section 8 has what that build showed).
#ifdef LK_X2_WAVES_PER_EU
#define LK_X2_ATTR __attribute__((amdgpu_waves_per_eu(LK_X2_WAVES_PER_EU, LK_X2_WAVES_PER_EU)))
#else
#define LK_X2_ATTR
#endif
// A/B builds only: extra dynamic LDS per workgroup of the shipped x2 launch caps its wavefronts per SIMD below what its
// registers allow (leaving registers for another stream's kernels); profiles/r03_lk_experiments.txt has what that measured.
#ifndef LK_X2_LDS_EXTRA
#define LK_X2_LDS_EXTRA 0
#endif
template <int PPW, bool PRE, bool CNT>
__global__ __launch_bounds__(64) LK_X2_ATTR void lk_track_kernel_x2(LKArgs a) {
    extern __shared__ unsigned lds[];
    constexpr int GL = x2::Cfg<PPW>::GL, NS = x2::Cfg<PPW>::NS;
    const int lane = threadIdx.x, l32 = lane & (GL - 1), half = lane / GL;
    int i = (int)blockIdx.x * PPW + half;
    const bool valid = i < a.n;
    if (!valid) i = a.n - 1;                      // a ragged tail: spare groups shadow the last point, write nothing
    unsigned* region = lds + half * x2::REGION;
    const int partner_addr = (lane ^ 16) << 2;
    int sx[NS], sy0[NS];
    bool sok[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int seg = l32 + GL * s;
        sok[s] = seg < 3 * x2::W;
        const int g = sok[s] ? seg / x2::W : 0;
        sx[s] = sok[s] ? seg - g * x2::W : 0;
        sy0[s] = g * x2::R;
    }
    const float2 p = a.prev_pts[i];
    int idx = a.img_idx ? a.img_idx[i] : 0;
    const bool bad_idx = (unsigned)idx >= (unsigned)a.n_images;      // group-uniform: tracked in slot 0, reported lost
    if (bad_idx) idx = 0;
    float fx = 0.f, fy = 0.f;
    if (a.flags & 4) { const float2 q = a.next_pts[i]; fx = q.x; fy = q.y; }
    int st = 1;
    float er = 0.f;
    float bx = 0.f, by = 0.f;
    const int ndir = a.fb ? 2 : 1;
    unsigned n_lv = 0, n_it = 0;
    for (int dir = 0; dir < ndir; ++dir) {
        const VofeLevel* LA = dir ? a.lv_b : a.lv_a;
        const VofeLevel* LB = dir ? a.lv_a : a.lv_b;
        const int sa = (dir ? a.slot_b : a.slot_a) + idx, sb = (dir ? a.slot_a : a.slot_b) + idx;
        float qx = dir ? fx : p.x, qy = dir ? fy : p.y;
        float ox = dir ? 0.f : fx, oy = dir ? 0.f : fy, e = 0.f;
        int s = 1;
        x2::lk_one<PPW, PRE, CNT>(a, LA, LB, sa, sb, qx, qy, ox, oy, s, e, dir ? LK_NO_ERR : a.flags, region, l32, partner_addr, sx, sy0,
                        sok, n_lv, n_it);
        if (dir == 0) { fx = ox; fy = oy; st = s; er = e; }
        else { bx = ox; by = oy; }
        if (!__any(st)) break;                    // every keypoint of this wavefront was lost going forward
    }
    if (a.fb) {
        const float dx = fabsf(p.x - bx), dy = fabsf(p.y - by);
        float d = dx > dy ? dx : dy;
        if (dx != dx || dy != dy) d = __builtin_nanf("");
        st = st & (d < a.fb_thresh ? 1 : 0);
    }
    if (bad_idx) { st = 0; er = 0.f; }
    if (l32 == 0 && valid) {
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

// Smallest float x (in the order -inf < ... < +inf) with fl(x / c) >= thr, c > 0: fl(x / c) < thr exactly when x < that bound,
// because correctly rounded division by a positive constant is monotone in x.  NaN threshold: the test is never true -> NaN.
// Bisection over the floats' order-preserving integer image; the host's float division is the same IEEE operation as the
// kernels' division sequence.
inline unsigned ordered_bits(float f) { unsigned u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
inline float from_ordered_bits(unsigned k) { unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &u, 4); return f; }
float min_eig_numerator_bound(float thr, float c) {
    if (thr != thr) return thr;
    const float inf = __builtin_inff();
    volatile float q = inf / c;
    if (!(q >= thr)) return __builtin_nanf("");           // not even +inf passes (thr = NaN is handled above; unreachable otherwise)
    unsigned lo = ordered_bits(-inf), hi = ordered_bits(inf);           // invariant: hi passes; everything below lo fails
    while (lo < hi) {
        const unsigned mid = lo + (hi - lo) / 2;
        volatile float v = from_ordered_bits(mid);
        volatile float r = v / c;
        if (r >= thr) hi = mid; else lo = mid + 1;
    }
    return from_ordered_bits(lo);
}

template <int R, int WC = 0>
void launch_r(const LKArgs& a, size_t lds_bytes, hipStream_t st) {
    hipLaunchKernelGGL((lk_track_kernel<R, WC>), dim3(a.n), dim3(64), lds_bytes, st, a);
}

}  // namespace

int vofe_launch_lk(vofe_ctx* ctx, const vofe_pyr* prev, int prev_slot, const vofe_pyr* next,
                   int next_slot, const float* d_prev_pts, const int* d_img_idx, int n_images, float* d_next_pts,
                   uint8_t* d_status, float* d_err, int n, int max_count, double eps2, int flags,
                   float min_eig_thr, int fb, float fb_thresh) {
    LKArgs a;
    a.lv_a = prev->d_lv; a.lv_b = next->d_lv;
    a.n_levels = prev->levels;
    a.slot_a = prev_slot; a.slot_b = next_slot;
    a.prev_pts = reinterpret_cast<const float2*>(d_prev_pts);
    a.img_idx = d_img_idx;
    a.n_images = d_img_idx ? n_images : 1;
    a.next_pts = reinterpret_cast<float2*>(d_next_pts);
    a.status = d_status; a.err = d_err; a.n = n;
    a.win_w = prev->win_w; a.win_h = prev->win_h;
    if (a.win_w > 64) { vofe_set_error("vofe_lk_track: winSize.width %d > 64 is not supported", a.win_w); return VOFE_E_UNSUPPORTED; }
    a.groups = 64 / a.win_w;
    if (a.groups > a.win_h) a.groups = a.win_h;
    a.rows_per_group = (a.win_h + a.groups - 1) / a.groups;
    if (a.rows_per_group > 64) { vofe_set_error("vofe_lk_track: winSize %dx%d too tall (max 64 rows per lane)", a.win_w, a.win_h); return VOFE_E_UNSUPPORTED; }
    a.max_count = max_count; a.flags = flags; a.fb = fb;
    a.min_eig_thr = min_eig_thr; a.fb_thresh = fb_thresh; a.eps2 = eps2;
    a.min_eig_num = min_eig_numerator_bound(min_eig_thr, (float)(2 * a.win_w * a.win_h));
    a.eps2_lo = nextafterf((float)(eps2 * (1.0 - 1e-6)), -1.f);
    a.eps2_hi = nextafterf((float)(eps2 * (1.0 + 1e-6)), 1e30f);
    a.counters = ctx->lk_counters;
    const int rw = a.win_w + 1 + 2 * LK_MARGIN;
    a.rh = a.win_h + 1 + 2 * LK_MARGIN;
    a.segs = (rw + 7) / 8;
    a.rs = a.segs * 8;
    a.inv_segs = (1 << 20) / a.segs + 1;
    for (int t = 0; t < a.rh * a.segs; ++t)
        if ((int)(((unsigned)t * (unsigned)a.inv_segs) >> 20) != t / a.segs) {
            vofe_set_error("vofe_lk_track: internal reciprocal check failed");
            return VOFE_E_UNSUPPORTED;
        }
    const size_t lds_bytes = (size_t)a.rs * a.rh * 4;
    const int R = a.rows_per_group;
    hipStream_t st = ctx->stream;
    const char* force = ctx->knobs.lk[0] ? ctx->knobs.lk : nullptr;   // VOFE_LK: "generic" skips the 21x21 specialisation (tests)
    // large 21x21 launches saturate the vector ALUs: two keypoints per wavefront share the uniform work there;
    // small launches are latency bound and keep one keypoint per wavefront ("x2" / "x1" force a kernel)
    // (its 32-bit lane offsets must span every slot of a level's derivative plane)
    const unsigned long long span_a = (unsigned long long)prev->slots * (unsigned long long)prev->lv[0].slot_stride * 4ull;
    const unsigned long long span_b = (unsigned long long)next->slots * (unsigned long long)next->lv[0].slot_stride * 4ull;
    const bool x2_ok = a.win_w == 21 && a.win_h == 21 && !(force && force[0] == 'g') && span_a < 0xf0000000ull && span_b < 0xf0000000ull;
    // small launches of the reference's two windows: one wavefront per pyramid level ("lp" forces it for any point count;
    // not with OPTFLOW_LK_GET_MIN_EIGENVALS, whose err output follows the levels in order)
    const bool lp_ok = a.win_w == a.win_h && (a.win_w == 21 || a.win_w == 11) && a.n_levels >= 2 && a.n_levels <= LK_LP_MAX_LEVELS &&
                       !(flags & 8) && !(force && force[0] != 'l');
    const size_t lp_lds = ((size_t)LK_LP_HAND + (size_t)a.n_levels * a.rs * a.rh) * 4;
    if (x2_ok && force && force[0] == 'x' && force[1] == '4')
        hipLaunchKernelGGL((lk_track_kernel_x2<4, true, true>), dim3((a.n + 3) / 4), dim3(64), (size_t)4 * x2::REGION * 4, st, a);
    else if (x2_ok && force && force[0] == 'x' && force[1] == '2' && force[2] == 'p')          // "x2p": the prefetching variant (3 waves per SIMD)
        hipLaunchKernelGGL((lk_track_kernel_x2<2, true, true>), dim3((a.n + 1) / 2), dim3(64), (size_t)2 * x2::REGION * 4, st, a);
    else if (x2_ok && ((force && force[0] == 'x' && force[1] == '2') || (!force && n >= LK_X2_MIN_POINTS)))
    {
        if (a.counters) hipLaunchKernelGGL((lk_track_kernel_x2<2, false, true>), dim3((a.n + 1) / 2), dim3(64), (size_t)2 * x2::REGION * 4 + LK_X2_LDS_EXTRA, st, a);
        else hipLaunchKernelGGL((lk_track_kernel_x2<2, false, false>), dim3((a.n + 1) / 2), dim3(64), (size_t)2 * x2::REGION * 4 + LK_X2_LDS_EXTRA, st, a);
    }
    else if (lp_ok && a.win_w == 21 && ((force && force[0] == 'l') || (!force && n <= LK_LP_MAX_POINTS)))
        hipLaunchKernelGGL((lk_track_kernel_lp<7, 21>), dim3(a.n), dim3(64 * a.n_levels), lp_lds, st, a);
    else if (lp_ok && a.win_w == 11 && ((force && force[0] == 'l') || (!force && n <= LK_LP_MAX_POINTS)))
        hipLaunchKernelGGL((lk_track_kernel_lp<3, 11>), dim3(a.n), dim3(64 * a.n_levels), lp_lds, st, a);
    else if (a.win_w == 21 && a.win_h == 21 && !(force && force[0] == 'g')) launch_r<7, 21>(a, lds_bytes, st);
    // the other windows the reference uses (constants.py:49 Parking 11x11) and the stress configuration's 31x31:
    // compile-time geometry with a ragged last row group (4 x 3 rows for 11, 16 + 15 rows for 31)
    else if (a.win_w == 11 && a.win_h == 11 && !(force && force[0] == 'g')) launch_r<3, 11>(a, lds_bytes, st);
    else if (a.win_w == 31 && a.win_h == 31 && !(force && force[0] == 'g')) launch_r<16, 31>(a, lds_bytes, st);
    else if (R <= 2) launch_r<2>(a, lds_bytes, st);
    else if (R <= 3) launch_r<3>(a, lds_bytes, st);
    else if (R <= 4) launch_r<4>(a, lds_bytes, st);
    else if (R <= 7) launch_r<7>(a, lds_bytes, st);
    else if (R <= 12) launch_r<12>(a, lds_bytes, st);
    else if (R <= 16) launch_r<16>(a, lds_bytes, st);
    else if (R <= 24) launch_r<24>(a, lds_bytes, st);
    else if (R <= 32) launch_r<32>(a, lds_bytes, st);
    else if (R <= 41) launch_r<41>(a, lds_bytes, st);
    else launch_r<64>(a, lds_bytes, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "lk_track launch");
    return VOFE_OK;
}
