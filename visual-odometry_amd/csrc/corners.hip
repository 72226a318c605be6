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
#include <vector>

#include "vofe_internal.h"

namespace {

constexpr int TX = 64, TY = 16;                 // score tile (outputs) per 256-thread block
constexpr int SORT_CH = 8192;                   // keys sorted in LDS per block (64 KiB)
constexpr int SORT_THREADS = 1024;
constexpr int SELECT_ROUNDS = 8;                // rounds launched between host checks

struct GfttHdr {
    unsigned max_key;       // order-preserving key of the global score maximum (0 = none)
    unsigned n_cand;
    unsigned n_acc;
    unsigned pad;
    unsigned undecided[SELECT_ROUNDS];
};

__device__ __forceinline__ unsigned float_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
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
        if (tid == 0 && sMax) atomicMax(&hdr[blockIdx.z].max_key, sMax);
    }
}

// ---------------------------------------------------------------------------------------------
// threshold + 3x3 local maximum + compaction.  state: 0 none, 1 undecided.
constexpr int NMS_ROWS = 8;      // rows per thread: a 256-thread block covers 64 x 32 pixels
__global__ __launch_bounds__(256) void nms_compact(const float* __restrict__ score, int h, int w, double quality,
                                                   GfttHdr* __restrict__ hdr, uint8_t* __restrict__ state,
                                                   unsigned* __restrict__ cand, const uint8_t* __restrict__ mask) {
    __shared__ unsigned s_wave[4];
    __shared__ unsigned s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = blockIdx.y * (4 * NMS_ROWS) + wv;
    const size_t ib = (size_t)blockIdx.z * h * w;
    const float* sc = score + ib;
    const unsigned mk = hdr[blockIdx.z].max_key;
    const float thr = mk ? (float)((double)key_float(mk) * quality) : 0.f;
    unsigned long long bal[NMS_ROWS];
    unsigned total = 0;
#pragma unroll
    for (int k = 0; k < NMS_ROWS; ++k) {
        const int y = y0 + 4 * k;
        bool is = false;
        if (mk && x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
            const float* p = sc + (size_t)y * w + x;
            const float v = p[0];
            if (v > thr && v != 0.f && (!mask || mask[(size_t)y * w + x])) {
                float m = fmaxf(fmaxf(p[-w - 1], p[-w]), p[-w + 1]);
                m = fmaxf(m, fmaxf(p[-1], p[1]));
                m = fmaxf(m, fmaxf(fmaxf(p[w - 1], p[w]), p[w + 1]));
                is = v >= m;
            }
        }
        if (x < w && y < h) state[ib + (size_t)y * w + x] = is ? 1 : 0;
        bal[k] = __ballot(is);
        total += (unsigned)__popcll(bal[k]);
    }
    if (lane == 0) s_wave[wv] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = t ? atomicAdd(&hdr[blockIdx.z].n_cand, t) : 0u;
    }
    __syncthreads();
    unsigned pos = s_base;
    for (int q = 0; q < wv; ++q) pos += s_wave[q];
#pragma unroll
    for (int k = 0; k < NMS_ROWS; ++k) {
        if ((bal[k] >> lane) & 1ull)
            cand[ib + pos + __popcll(bal[k] & ((1ull << lane) - 1))] = (unsigned)((y0 + 4 * k) * w + x);
        pos += (unsigned)__popcll(bal[k]);
    }
}

// ---------------------------------------------------------------------------------------------
// one round of the min-distance fixed point: one wavefront per candidate, grid-stride.
// state: 1 undecided, 2 accepted, 3 rejected.  Any interleaving of rounds gives the same fixed
// point because a decision is only taken from final (2/3) states of higher-ranked neighbours.
__global__ __launch_bounds__(256) void select_round(const float* __restrict__ score, int h, int w, int r, float md2,
                                                    GfttHdr* __restrict__ hdr, uint8_t* __restrict__ state,
                                                    const unsigned* __restrict__ cand, int round) {
    const size_t ib = (size_t)blockIdx.y * h * w;
    const float* sc = score + ib;
    uint8_t* stt = state + ib;
    const unsigned n = hdr[blockIdx.y].n_cand;
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int side = 2 * r + 1, area = side * side;
    for (unsigned c = wave; c < n; c += nwaves) {
        const unsigned p = cand[ib + c];
        if (stt[p] != 1) continue;
        const int py = (int)(p / (unsigned)w), px = (int)(p - (unsigned)py * w);
        const unsigned pbits = __float_as_uint(sc[p]);
        bool rej = false, blocked = false;
        for (int e = lane; e < area; e += 64) {
            const int dy = e / side - r, dx = e - (dy + r) * side - r;
            const int qx = px + dx, qy = py + dy;
            if (qx < 0 || qx >= w || qy < 0 || qy >= h) continue;
            if ((float)(dx * dx + dy * dy) >= md2) continue;
            const unsigned q = (unsigned)(qy * w + qx);
            const unsigned s = stt[q];
            if (s == 0 || s == 3 || q == p) continue;
            const unsigned qbits = __float_as_uint(sc[q]);
            const bool higher = qbits > pbits || (qbits == pbits && q > p);   // scores are > 0: bit order = value order
            if (!higher) continue;
            if (s == 2) rej = true; else blocked = true;
        }
        const bool any_rej = __any(rej), any_blk = __any(blocked);
        if (lane == 0) {
            if (any_rej) stt[p] = 3;
            else if (!any_blk) stt[p] = 2;
            else hdr[blockIdx.y].undecided[round] = 1u;      // flag only: a counter here is one contended atomic per wave
        }
    }
}

// gather accepted candidates as 64-bit keys (score bits << 32 | linear index)
__global__ __launch_bounds__(256) void gather_accepted(const float* __restrict__ score, int h, int w,
                                                       GfttHdr* __restrict__ hdr, const uint8_t* __restrict__ state,
                                                       const unsigned* __restrict__ cand, unsigned long long* __restrict__ keys,
                                                       int take_all) {
    const size_t ib = (size_t)blockIdx.y * h * w;
    const unsigned n = hdr[blockIdx.y].n_cand;
    const int lane = threadIdx.x & 63;
    for (unsigned c0 = (blockIdx.x * 256 + threadIdx.x) & ~63u; c0 < n; c0 += gridDim.x * 256) {
        const unsigned c = c0 + lane;
        bool ok = false;
        unsigned p = 0;
        if (c < n) { p = cand[ib + c]; ok = take_all || state[ib + p] == 2; }
        const unsigned long long bal = __ballot(ok);
        if (!bal) continue;
        unsigned base = 0;
        const int leader = __ffsll((long long)bal) - 1;
        if (lane == leader) base = atomicAdd(&hdr[blockIdx.y].n_acc, (unsigned)__popcll(bal));
        base = (unsigned)__shfl((int)base, leader);
        if (ok) keys[ib + base + __popcll(bal & ((1ull << lane) - 1))] =
                    ((unsigned long long)__float_as_uint(score[ib + p]) << 32) | p;
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

int ensure_ws(vofe_ctx* ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return VOFE_OK;
    if (ctx->ws) { hipStreamSynchronize(ctx->stream); hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    hipError_t e = hipMalloc(&ctx->ws, bytes);
    if (e != hipSuccess) return vofe_hip_fail(e, "hipMalloc(gftt workspace)");
    ctx->ws_bytes = bytes;
    return VOFE_OK;
}
int ensure_pinned(vofe_ctx* ctx, size_t ints) {
    if (ctx->h_pinned_ints >= ints) return VOFE_OK;
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    ctx->h_pinned = nullptr; ctx->h_pinned_ints = 0;
    hipError_t e = hipHostMalloc((void**)&ctx->h_pinned, ints * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) return vofe_hip_fail(e, "hipHostMalloc");
    ctx->h_pinned_ints = ints;
    return VOFE_OK;
}

}  // namespace

int vofe_launch_corner_score(vofe_ctx* ctx, const uint8_t* d_img, int batch, int h, int w, int pitch,
                             size_t img_stride, int harris, float k, float* d_score,
                             unsigned* d_hdr, const uint8_t* d_mask) {
    dim3 grid((w + TX - 1) / TX, (h + TY - 1) / TY, batch);
    if (grid.y > 65535 || grid.z > 65535) { vofe_set_error("corner_score: image or batch too large"); return VOFE_E_UNSUPPORTED; }
    hipLaunchKernelGGL(corner_score_f32, grid, dim3(256), 0, ctx->stream, d_img, h, w, pitch, img_stride, harris, k,
                       d_score, reinterpret_cast<GfttHdr*>(d_hdr), d_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vofe_hip_fail(e, "corner_score launch");
    return VOFE_OK;
}

int vofe_gftt_impl(vofe_ctx* ctx, const uint8_t* d_img, int batch, int h, int w, int pitch,
                   size_t img_stride, int max_corners, double quality, double min_distance,
                   const uint8_t* d_mask, int harris, double k, float* d_corners_xy, int capacity,
                   int* h_n_out) {
    hipStream_t st = ctx->stream;
    const size_t px = (size_t)h * w;
    if (px >= (1ull << 31)) { vofe_set_error("vofe_gftt: image too large"); return VOFE_E_UNSUPPORTED; }
    // workspace: hdr | score f32 | cand u32 | keys u64 | state u8
    const size_t hdr_bytes = ((sizeof(GfttHdr) * batch + 255) / 256) * 256;
    const size_t tot = hdr_bytes + batch * px * (4 + 4 + 8 + 1) + 1024;
    int rc = ensure_ws(ctx, tot);
    if (rc) return rc;
    rc = ensure_pinned(ctx, (sizeof(GfttHdr) / 4) * (size_t)batch);
    if (rc) return rc;
    uint8_t* base = static_cast<uint8_t*>(ctx->ws);
    GfttHdr* hdr = reinterpret_cast<GfttHdr*>(base);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(base + hdr_bytes);
    float* score = reinterpret_cast<float*>(base + hdr_bytes + batch * px * 8);
    unsigned* cand = reinterpret_cast<unsigned*>(base + hdr_bytes + batch * px * 12);
    uint8_t* state = base + hdr_bytes + batch * px * 16;

    VOFE_HIP(hipMemsetAsync(hdr, 0, hdr_bytes, st));
    rc = vofe_launch_corner_score(ctx, d_img, batch, h, w, pitch, img_stride, harris, (float)k, score,
                                  reinterpret_cast<unsigned*>(hdr), d_mask);
    if (rc) return rc;
    hipLaunchKernelGGL(nms_compact, dim3((w + 63) / 64, (h + 4 * NMS_ROWS - 1) / (4 * NMS_ROWS), batch), dim3(256), 0, st, score, h, w, quality, hdr,
                       state, cand, d_mask);
    const bool suppress = min_distance >= 1.0;
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
    const int sel_blocks = 512;
    for (int pass = 0;; ++pass) {
        if (suppress) {
            if (pass > 0) {
                // clear the round counters and the accepted count, keep max_key / n_cand
                for (int b = 0; b < batch; ++b) {
                    VOFE_HIP(hipMemsetAsync(&hdr[b].n_acc, 0, sizeof(unsigned), st));
                    VOFE_HIP(hipMemsetAsync(hdr[b].undecided, 0, sizeof(unsigned) * SELECT_ROUNDS, st));
                }
            }
            for (int round = 0; round < SELECT_ROUNDS; ++round)
                hipLaunchKernelGGL(select_round, dim3(sel_blocks, batch), dim3(256), 0, st, score, h, w, r, md2, hdr, state,
                                   cand, round);
        }
        hipLaunchKernelGGL(gather_accepted, dim3(256, batch), dim3(256), 0, st, score, h, w, hdr, state, cand, keys,
                           suppress ? 0 : 1);
        VOFE_HIP(hipMemcpyAsync(hh, hdr, sizeof(GfttHdr) * batch, hipMemcpyDeviceToHost, st));
        VOFE_HIP(hipStreamSynchronize(st));
        bool done = true;
        if (suppress)
            for (int b = 0; b < batch; ++b)
                if (hh[b].undecided[SELECT_ROUNDS - 1]) done = false;
        if (done) break;
        if (pass > 1000000) { vofe_set_error("vofe_gftt: selection did not converge"); return VOFE_E_HIP; }
    }
    unsigned nmax = 0;
    for (int b = 0; b < batch; ++b) {
        const unsigned n = hh[b].n_acc;
        nmax = n > nmax ? n : nmax;
        h_n_out[b] = (int)((max_corners > 0 && n > (unsigned)max_corners) ? (unsigned)max_corners : n);
    }
    if (nmax == 0) return VOFE_OK;
    int max_out = capacity;
    if (max_corners > 0 && max_corners < max_out) max_out = max_corners;
    const unsigned chunks = (nmax + SORT_CH - 1) / SORT_CH;
    const size_t lds = (size_t)SORT_CH * 8;
    if (chunks == 1) {
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
    return VOFE_OK;
}
