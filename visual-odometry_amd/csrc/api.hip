// C ABI of libvofe.so (declared in include/vofe.h): argument checks that mirror the cv2 asserts,
// context / pyramid object management, and dispatch to the kernel launchers.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "vofe_internal.h"

static thread_local char g_err[512] = "";

void vofe_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int vofe_hip_fail(hipError_t e, const char* what) {
    vofe_set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return VOFE_E_HIP;
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
static bool env_is(const char* name, char first) {
    const char* v = getenv(name);
    return v && v[0] == first;
}
void vofe_read_knobs(VofeKnobs* k) {
    memset(k, 0, sizeof(*k));
    if (const char* v = getenv("VOFE_LK")) { strncpy(k->lk, v, sizeof(k->lk) - 1); }
    if (const char* v = getenv("VOFE_PYRAMID")) k->pyramid = (v[0] == 'f' || v[0] == 'l') ? v[0] : 0;
    k->pad_copy_generic = env_is("VOFE_PAD_COPY", 'g');
    k->score_tile = env_is("VOFE_CORNER_SCORE", 't');
    k->score_generic = env_is("VOFE_CORNER_SCORE", 'g');
    k->score_rs = env_int("VOFE_SCORE_RS", 0);
    k->nms_rows = env_int("VOFE_NMS_ROWS", 0);
    k->pyr_borders_separate = env_is("VOFE_PYR_BORDERS", 's');
    k->pyr_step_off = env_is("VOFE_PYR_STEP", '0');
    k->pyr_tail_off = env_is("VOFE_PYR_TAIL", '0');
    k->pyr_tail_split = env_is("VOFE_PYR_TAIL", 's');
    k->pyr_tail_th = env_int("VOFE_PYR_TAIL_TH", 0);
    k->gftt_bitonic = env_is("VOFE_GFTT_SORT", 'b');
    k->gftt_debug = getenv("VOFE_GFTT_DEBUG") != nullptr;
    if (const char* v = getenv("VOFE_GFTT_CHAIN")) k->gftt_chain = v[0] == 's' ? 's' : (v[0] == 'r' ? 'r' : 0);
    k->sel_blocks = env_int("VOFE_SEL_BLOCKS", 1024);
    k->sel_blocks_set = getenv("VOFE_SEL_BLOCKS") != nullptr;
    k->sel_launches = env_int("VOFE_SEL_LAUNCHES", 1);
    k->sel_passes = env_int("VOFE_SEL_PASSES", 2);
    k->sel_spins = env_int("VOFE_SEL_SPINS", 64);
    k->sel_prefix = env_int("VOFE_SEL_PREFIX", 0);
    k->sel_full = env_int("VOFE_SEL_FULL", 0);
}

static inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

// Every launching entry point makes the ctx's device current for the calling thread (a thread-local read when it
// already is), so a process that drives several GPUs from one thread cannot launch on the wrong one.
static int bind_device(const vofe_ctx* ctx) {
    // The launchers check hipGetLastError() after a launch; that call also returns whatever an EARLIER runtime call of this
    // thread left behind (its caller has had that code already), which would be reported as this entry point's failure.
    (void)hipGetLastError();
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == ctx->device) return VOFE_OK;
    VOFE_HIP(hipSetDevice(ctx->device));
    return VOFE_OK;
}
#define VOFE_BIND(ctx) do { int rc_ = bind_device(ctx); if (rc_) return rc_; } while (0)

extern "C" {

const char* vofe_last_error(void) { return g_err; }
int vofe_version(void) { return 100; }

int vofe_create(int device, void* hip_stream, vofe_ctx** out) {
    if (!out) { vofe_set_error("vofe_create: out is NULL"); return VOFE_E_ARG; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        vofe_set_error("vofe_create: no HIP device available (%s); this library has no CPU fallback",
                       e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return VOFE_E_HIP;
    }
    if (device < 0 || device >= ndev) { vofe_set_error("vofe_create: device %d out of range [0,%d)", device, ndev); return VOFE_E_ARG; }
    VOFE_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    VOFE_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vofe_set_error("vofe_create: device %d is %s; libvofe is built for gfx950 only", device, prop.gcnArchName);
        return VOFE_E_UNSUPPORTED;
    }
    vofe_ctx* c = new (std::nothrow) vofe_ctx();
    if (!c) { vofe_set_error("vofe_create: out of host memory"); return VOFE_E_NOMEM; }
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->ws = nullptr; c->ws_bytes = 0;
    c->h_pinned = nullptr; c->d_pinned = nullptr; c->h_pinned_ints = 0;
    c->gftt_clean_batch = 0; c->gftt_clean_ws = nullptr; c->gftt_last_ncand = 0;
    c->lk_counters = nullptr;
    c->prof = nullptr;
    vofe_read_knobs(&c->knobs);
    int rc = vofe_gftt_setup(c);
    if (rc) { if (c->h_pinned) hipHostFree(c->h_pinned); delete c; return rc; }
    *out = c;
    return VOFE_OK;
}

int vofe_reload_env(vofe_ctx* ctx) {
    if (!ctx) { vofe_set_error("vofe_reload_env: ctx is NULL"); return VOFE_E_ARG; }
    vofe_read_knobs(&ctx->knobs);
    return VOFE_OK;
}

int vofe_destroy(vofe_ctx* ctx) {
    if (!ctx) return VOFE_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->ws) hipFree(ctx->ws);
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    if (ctx->prof) {
        for (auto& e : ctx->prof->entries) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
        delete ctx->prof;
    }
    delete ctx;
    return VOFE_OK;
}

int vofe_synchronize(vofe_ctx* ctx) {
    if (!ctx) { vofe_set_error("vofe_synchronize: ctx is NULL"); return VOFE_E_ARG; }
    VOFE_BIND(ctx);
    VOFE_HIP(hipStreamSynchronize(ctx->stream));
    return VOFE_OK;
}

// ------------------------------------------------------------------------------- pyramids
int vofe_pyramid_create(vofe_ctx* ctx, int slots, int height, int width, int win_w, int win_h,
                        int max_level, int with_deriv, vofe_pyr** out) {
    if (!ctx || !out) { vofe_set_error("vofe_pyramid_create: NULL argument"); return VOFE_E_ARG; }
    *out = nullptr;
    if (slots <= 0 || height <= 0 || width <= 0) { vofe_set_error("vofe_pyramid_create: empty image or no slots"); return VOFE_E_ARG; }
    if (max_level < 0 || win_w <= 2 || win_h <= 2) {      // cv2 assert: maxLevel >= 0 && winSize > 2
        vofe_set_error("vofe_pyramid_create: need maxLevel >= 0 and winSize > 2 (got %d, %dx%d)", max_level, win_w, win_h);
        return VOFE_E_ARG;
    }
    if (height + 2 * win_h > 65535) { vofe_set_error("vofe_pyramid_create: image too tall (%d rows)", height); return VOFE_E_UNSUPPORTED; }
    VOFE_BIND(ctx);
    vofe_pyr* p = new (std::nothrow) vofe_pyr();
    if (!p) { vofe_set_error("vofe_pyramid_create: out of host memory"); return VOFE_E_NOMEM; }
    memset(p, 0, sizeof(*p));
    p->slots = slots; p->height = height; p->width = width;
    p->win_w = win_w; p->win_h = win_h; p->with_deriv = with_deriv ? 1 : 0;
    p->pad_x = (int)align_up(win_w + 20, 16);
    p->pad_y = win_h;
    // stop when the next level would not exceed the window
    int w = width, h = height, n = 0;
    long long total = 0;
    long long base[VOFE_MAX_LEVELS];
    for (int level = 0; level <= max_level && n < VOFE_MAX_LEVELS; ++level) {
        VofeLevel& L = p->lv[n];
        L.w = w; L.h = h;
        L.pitch = (int)align_up(w + 2 * p->pad_x, 64);
        L.rows = h + 2 * p->pad_y;
        L.slot_stride = align_up((long long)L.pitch * L.rows, 64);
        base[n] = total;
        total += L.slot_stride * slots;
        ++n;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win_w || h <= win_h) break;
    }
    p->levels = n;
    hipError_t e = hipMalloc((void**)&p->d_I, (size_t)total + 256);
    if (e != hipSuccess) { delete p; return vofe_hip_fail(e, "hipMalloc(pyramid images)"); }
    e = hipMemsetAsync(p->d_I, 0, (size_t)total + 256, ctx->stream);
    if (e != hipSuccess) { hipFree(p->d_I); delete p; return vofe_hip_fail(e, "hipMemsetAsync(pyramid images)"); }
    if (p->with_deriv) {
        e = hipMalloc((void**)&p->d_D, ((size_t)total + 64) * 4);
        if (e != hipSuccess) { hipFree(p->d_I); delete p; return vofe_hip_fail(e, "hipMalloc(pyramid derivatives)"); }
        // the zero border of the derivative planes (cv: BORDER_CONSTANT) is never written by a kernel: it must be cleared
        e = hipMemsetAsync(p->d_D, 0, ((size_t)total + 64) * 4, ctx->stream);
        if (e != hipSuccess) { hipFree(p->d_I); hipFree(p->d_D); delete p; return vofe_hip_fail(e, "hipMemsetAsync(pyramid derivatives)"); }
    }
    for (int l = 0; l < n; ++l) {
        VofeLevel& L = p->lv[l];
        const long long o = base[l] + (long long)p->pad_y * L.pitch + p->pad_x;
        L.I = p->d_I + o;
        L.D = p->d_D ? p->d_D + o : nullptr;
    }
    e = hipMalloc((void**)&p->d_lv, sizeof(VofeLevel) * VOFE_MAX_LEVELS);
    if (e != hipSuccess) { hipFree(p->d_I); if (p->d_D) hipFree(p->d_D); delete p; return vofe_hip_fail(e, "hipMalloc(level table)"); }
    e = hipMemcpyAsync(p->d_lv, p->lv, sizeof(VofeLevel) * VOFE_MAX_LEVELS, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { vofe_pyramid_free(ctx, p); return vofe_hip_fail(e, "pyramid table upload"); }
    *out = p;
    return VOFE_OK;
}

int vofe_pyramid_free(vofe_ctx* ctx, vofe_pyr* pyr) {
    if (!pyr) return VOFE_OK;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    if (pyr->d_I) hipFree(pyr->d_I);
    if (pyr->d_D) hipFree(pyr->d_D);
    if (pyr->d_lv) hipFree(pyr->d_lv);
    delete pyr;
    return VOFE_OK;
}

int vofe_pyramid_build(vofe_ctx* ctx, vofe_pyr* pyr, int slot0, int count, const uint8_t* d_img,
                       int pitch, size_t img_stride) {
    if (!ctx || !pyr || !d_img) { vofe_set_error("vofe_pyramid_build: NULL argument"); return VOFE_E_ARG; }
    if (slot0 < 0 || count <= 0 || slot0 + count > pyr->slots) {
        vofe_set_error("vofe_pyramid_build: slots [%d,%d) outside [0,%d)", slot0, slot0 + count, pyr->slots);
        return VOFE_E_ARG;
    }
    if (pitch < pyr->width) { vofe_set_error("vofe_pyramid_build: pitch %d < width %d", pitch, pyr->width); return VOFE_E_ARG; }
    if (count > 65535) { vofe_set_error("vofe_pyramid_build: count %d > 65535", count); return VOFE_E_UNSUPPORTED; }
    VOFE_BIND(ctx);
    return vofe_launch_pyramid(ctx, pyr, slot0, count, d_img, pitch, img_stride);
}

int vofe_pyramid_levels(const vofe_pyr* pyr) { return pyr ? pyr->levels : VOFE_E_ARG; }

int vofe_pyramid_level_size(const vofe_pyr* pyr, int level, int* width, int* height) {
    if (!pyr || level < 0 || level >= pyr->levels) { vofe_set_error("vofe_pyramid_level_size: bad level"); return VOFE_E_ARG; }
    if (width) *width = pyr->lv[level].w;
    if (height) *height = pyr->lv[level].h;
    return VOFE_OK;
}

int vofe_pyramid_download_padded(vofe_ctx* ctx, const vofe_pyr* pyr, int slot, int level,
                                 int pad_x, int pad_y, uint8_t* h_image) {
    if (!ctx || !pyr || !h_image || slot < 0 || slot >= pyr->slots || level < 0 || level >= pyr->levels ||
        pad_x < 0 || pad_y < 0 || pad_x > pyr->win_w || pad_y > pyr->win_h) {
        vofe_set_error("vofe_pyramid_download_padded: bad argument");
        return VOFE_E_ARG;
    }
    const VofeLevel& L = pyr->lv[level];
    VOFE_BIND(ctx);
    VOFE_HIP(hipStreamSynchronize(ctx->stream));
    const uint8_t* src = L.I + (long long)slot * L.slot_stride - (long long)pad_y * L.pitch - pad_x;
    const size_t wbytes = (size_t)L.w + 2 * pad_x;
    VOFE_HIP(hipMemcpy2D(h_image, wbytes, src, (size_t)L.pitch, wbytes, (size_t)L.h + 2 * pad_y, hipMemcpyDeviceToHost));
    return VOFE_OK;
}

int vofe_pyramid_download(vofe_ctx* ctx, const vofe_pyr* pyr, int slot, int level,
                          uint8_t* h_image, int16_t* h_deriv) {
    if (!ctx || !pyr || slot < 0 || slot >= pyr->slots || level < 0 || level >= pyr->levels) {
        vofe_set_error("vofe_pyramid_download: bad argument");
        return VOFE_E_ARG;
    }
    const VofeLevel& L = pyr->lv[level];
    VOFE_BIND(ctx);
    VOFE_HIP(hipStreamSynchronize(ctx->stream));
    if (h_image)
        VOFE_HIP(hipMemcpy2D(h_image, (size_t)L.w, L.I + (long long)slot * L.slot_stride, (size_t)L.pitch, (size_t)L.w,
                             (size_t)L.h, hipMemcpyDeviceToHost));
    if (h_deriv) {
        if (!L.D) { vofe_set_error("vofe_pyramid_download: pyramid has no derivative planes"); return VOFE_E_ARG; }
        VOFE_HIP(hipMemcpy2D(h_deriv, (size_t)L.w * 4, L.D + (long long)slot * L.slot_stride, (size_t)L.pitch * 4,
                             (size_t)L.w * 4, (size_t)L.h, hipMemcpyDeviceToHost));
    }
    return VOFE_OK;
}

// ------------------------------------------------------------------------------- LK
static int lk_common(const char* who, vofe_ctx* ctx, const vofe_pyr* prev, int prev_slot, const vofe_pyr* next,
                     int next_slot, const float* d_prev_pts, const int* d_img_idx, int& n_images, float* d_next_pts,
                     uint8_t* d_status, float* d_err, int n, int need_next_deriv) {
    if (!ctx || !prev || !next) { vofe_set_error("%s: NULL ctx or pyramid", who); return VOFE_E_ARG; }
    if (n < 0) { vofe_set_error("%s: n < 0", who); return VOFE_E_ARG; }
    if (n > 0 && (!d_prev_pts || !d_next_pts || !d_status || !d_err)) { vofe_set_error("%s: NULL point buffer", who); return VOFE_E_ARG; }
    if (prev->height != next->height || prev->width != next->width || prev->win_w != next->win_w ||
        prev->win_h != next->win_h || prev->levels != next->levels) {
        vofe_set_error("%s: prev and next pyramids differ in geometry (cv2 asserts equal size/type)", who);
        return VOFE_E_ARG;
    }
    if (!prev->with_deriv || (need_next_deriv && !next->with_deriv)) {
        vofe_set_error("%s: pyramid built without derivative planes", who);
        return VOFE_E_ARG;
    }
    if (!d_img_idx) n_images = 1;
    if (n_images < 1) { vofe_set_error("%s: n_images must be >= 1 with an image index", who); return VOFE_E_ARG; }
    // the kernels address slot + img_idx[i]: the whole index range must lie inside both pyramid objects
    if (prev_slot < 0 || next_slot < 0 || (long long)prev_slot + n_images > prev->slots ||
        (long long)next_slot + n_images > next->slots) {
        vofe_set_error("%s: slots [%d,+%d) / [%d,+%d) outside the pyramids (%d / %d slots)", who, prev_slot, n_images,
                       next_slot, n_images, prev->slots, next->slots);
        return VOFE_E_ARG;
    }
    return VOFE_OK;
}
static void lk_criteria(int type, int& max_count, double& eps) {
    max_count = (type & 1) ? (max_count < 0 ? 0 : (max_count > 100 ? 100 : max_count)) : 30;
    eps = (type & 2) ? (eps < 0 ? 0 : (eps > 10 ? 10 : eps)) : 0.01;
    eps *= eps;
}

int vofe_lk_track(vofe_ctx* ctx, const vofe_pyr* prev, int prev_slot, const vofe_pyr* next,
                  int next_slot, const float* d_prev_pts, const int* d_img_idx, int n_images, float* d_next_pts,
                  uint8_t* d_status, float* d_err, int n, int crit_type, int max_count, double eps,
                  int flags, double min_eig_thr) {
    int rc = lk_common("vofe_lk_track", ctx, prev, prev_slot, next, next_slot, d_prev_pts, d_img_idx, n_images, d_next_pts,
                       d_status, d_err, n, 0);
    if (rc || n == 0) return rc;
    VOFE_BIND(ctx);
    lk_criteria(crit_type, max_count, eps);
    if (vofe_lk_needs_wide(prev->win_w, prev->win_h) || ctx->knobs.lk[0] == 'w')
        return vofe_launch_lk_wide(ctx, prev, prev_slot, next, next_slot, d_prev_pts, d_img_idx, n_images, d_next_pts, d_status, d_err,
                                   n, max_count, eps, flags, (float)min_eig_thr, 0, 0.f);
    return vofe_launch_lk(ctx, prev, prev_slot, next, next_slot, d_prev_pts, d_img_idx, n_images, d_next_pts, d_status, d_err, n,
                          max_count, eps, flags, (float)min_eig_thr, 0, 0.f);
}

int vofe_track_fb(vofe_ctx* ctx, const vofe_pyr* prev, int prev_slot, const vofe_pyr* next,
                  int next_slot, const float* d_prev_pts, const int* d_img_idx, int n_images, float* d_next_pts,
                  uint8_t* d_status, float* d_err, int n, int crit_type, int max_count, double eps,
                  double min_eig_thr, float fb_thresh) {
    int rc = lk_common("vofe_track_fb", ctx, prev, prev_slot, next, next_slot, d_prev_pts, d_img_idx, n_images, d_next_pts,
                       d_status, d_err, n, 1);
    if (rc || n == 0) return rc;
    VOFE_BIND(ctx);
    lk_criteria(crit_type, max_count, eps);
    if (vofe_lk_needs_wide(prev->win_w, prev->win_h) || ctx->knobs.lk[0] == 'w')
        return vofe_launch_lk_wide(ctx, prev, prev_slot, next, next_slot, d_prev_pts, d_img_idx, n_images, d_next_pts, d_status, d_err,
                                   n, max_count, eps, 0, (float)min_eig_thr, 1, fb_thresh);
    return vofe_launch_lk(ctx, prev, prev_slot, next, next_slot, d_prev_pts, d_img_idx, n_images, d_next_pts, d_status, d_err, n,
                          max_count, eps, 0, (float)min_eig_thr, 1, fb_thresh);
}

int vofe_profile_begin(vofe_ctx* ctx) {
    if (!ctx) { vofe_set_error("vofe_profile_begin: ctx is NULL"); return VOFE_E_ARG; }
    if (!ctx->prof) ctx->prof = new (std::nothrow) VofeProf();
    if (!ctx->prof) { vofe_set_error("vofe_profile_begin: out of host memory"); return VOFE_E_NOMEM; }
    return VOFE_OK;
}

int vofe_profile_end(vofe_ctx* ctx, int capacity, char* names, float* ms, int* n_out) {
    if (!ctx || !n_out || capacity < 0 || (capacity > 0 && (!names || !ms))) { vofe_set_error("vofe_profile_end: bad argument"); return VOFE_E_ARG; }
    *n_out = 0;
    if (!ctx->prof) return VOFE_OK;
    VOFE_BIND(ctx);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    int n = 0;
    for (auto& en : ctx->prof->entries) {
        float t = 0.f;
        if (e == hipSuccess && n < capacity && hipEventElapsedTime(&t, en.a, en.b) == hipSuccess) {
            strncpy(names + (size_t)n * VOFE_PROFILE_NAME_LEN, en.name.c_str(), VOFE_PROFILE_NAME_LEN - 1);
            names[(size_t)n * VOFE_PROFILE_NAME_LEN + VOFE_PROFILE_NAME_LEN - 1] = 0;
            ms[n++] = t;
        }
        hipEventDestroy(en.a); hipEventDestroy(en.b);
    }
    delete ctx->prof;
    ctx->prof = nullptr;
    *n_out = n;
    if (e != hipSuccess) return vofe_hip_fail(e, "hipStreamSynchronize");
    return VOFE_OK;
}

int vofe_lk_set_counters(vofe_ctx* ctx, unsigned long long* d_counters) {
    if (!ctx) { vofe_set_error("vofe_lk_set_counters: ctx is NULL"); return VOFE_E_ARG; }
    ctx->lk_counters = d_counters;
    return VOFE_OK;
}

// ------------------------------------------------------------------------------- corners
static int corner_args(const char* who, vofe_ctx* ctx, const uint8_t* d_img, int h, int w, int pitch, int block_size, int aperture) {
    if (!ctx || !d_img) { vofe_set_error("%s: NULL argument", who); return VOFE_E_ARG; }
    if (h <= 0 || w <= 0 || pitch < w) { vofe_set_error("%s: bad image geometry %dx%d pitch %d", who, w, h, pitch); return VOFE_E_ARG; }
    // cv2 takes any blockSize >= 1 and an odd Sobel aperture up to 31 (its documentation names 1, 3, 5, 7); here: block 1..31,
    // aperture 1, 3, 5 or 7 (the reference passes neither, continuous.py:380-385: block 3 / aperture 3 has dedicated kernels)
    if (block_size < 1 || aperture < 1 || aperture % 2 == 0) {
        vofe_set_error("%s: blockSize %d / gradientSize %d: need blockSize >= 1 and an odd gradientSize >= 1", who, block_size, aperture);
        return VOFE_E_ARG;
    }
    if (block_size > 31 || aperture > 7) {
        vofe_set_error("%s: blockSize %d / gradientSize %d beyond what is implemented (blockSize <= 31; gradientSize 1, 3, 5, 7)", who, block_size, aperture);
        return VOFE_E_UNSUPPORTED;
    }
    return VOFE_OK;
}

int vofe_corner_score(vofe_ctx* ctx, const uint8_t* d_img, int batch, int height, int width,
                      int pitch, size_t img_stride, int block_size, int aperture, int harris,
                      double k, float* d_score) {
    int rc = corner_args("vofe_corner_score", ctx, d_img, height, width, pitch, block_size, aperture);
    if (rc) return rc;
    if (!d_score || batch <= 0) { vofe_set_error("vofe_corner_score: NULL score buffer or batch <= 0"); return VOFE_E_ARG; }
    VOFE_BIND(ctx);
    return vofe_launch_corner_score(ctx, d_img, batch, height, width, pitch, img_stride, harris, (float)k, d_score, nullptr, nullptr,
                                    block_size, aperture);
}

int vofe_gftt(vofe_ctx* ctx, const uint8_t* d_img, int height, int width, int pitch,
              int max_corners, double quality, double min_distance, const uint8_t* d_mask,
              int block_size, int aperture, int harris, double k, float* d_corners_xy,
              int capacity, int* h_n_out) {
    if (!h_n_out) { vofe_set_error("vofe_gftt: h_n_out is NULL"); return VOFE_E_ARG; }
    *h_n_out = 0;
    if (!(quality > 0) || !(min_distance >= 0) || max_corners < 0) {       // SURVEY A.2 asserts
        vofe_set_error("vofe_gftt: need qualityLevel > 0, minDistance >= 0, maxCorners >= 0");
        return VOFE_E_ARG;
    }
    int rc = corner_args("vofe_gftt", ctx, d_img, height, width, pitch, block_size, aperture);
    if (rc) return rc;
    if (capacity < 0 || (capacity > 0 && !d_corners_xy)) { vofe_set_error("vofe_gftt: bad output buffer"); return VOFE_E_ARG; }
    VOFE_BIND(ctx);
    return vofe_gftt_impl(ctx, d_img, 1, height, width, pitch, 0, max_corners, quality, min_distance, d_mask, harris, k,
                          d_corners_xy, capacity, h_n_out, nullptr, false, block_size, aperture);
}

int vofe_gftt_noalloc(vofe_ctx* ctx, const uint8_t* d_img, int height, int width, int pitch,
                      int max_corners, double quality, double min_distance, const uint8_t* d_mask,
                      int block_size, int aperture, int harris, double k, float* d_corners_xy,
                      int capacity, int* h_n_out) {
    if (!h_n_out) { vofe_set_error("vofe_gftt_noalloc: h_n_out is NULL"); return VOFE_E_ARG; }
    *h_n_out = 0;
    if (!(quality > 0) || !(min_distance >= 0) || max_corners < 0) {
        vofe_set_error("vofe_gftt_noalloc: need qualityLevel > 0, minDistance >= 0, maxCorners >= 0");
        return VOFE_E_ARG;
    }
    int rc = corner_args("vofe_gftt_noalloc", ctx, d_img, height, width, pitch, block_size, aperture);
    if (rc) return rc;
    if (capacity < 0 || (capacity > 0 && !d_corners_xy)) { vofe_set_error("vofe_gftt_noalloc: bad output buffer"); return VOFE_E_ARG; }
    VOFE_BIND(ctx);
    return vofe_gftt_impl(ctx, d_img, 1, height, width, pitch, 0, max_corners, quality, min_distance, d_mask, harris, k,
                          d_corners_xy, capacity, h_n_out, nullptr, true, block_size, aperture);
}

int vofe_gftt_host(vofe_ctx* ctx, const uint8_t* d_img, int height, int width, int pitch,
                   int max_corners, double quality, double min_distance, const uint8_t* d_mask,
                   int block_size, int aperture, int harris, double k, float* d_corners_xy,
                   float* h_corners_xy, int capacity, int* h_n_out) {
    if (!h_n_out || !h_corners_xy) { vofe_set_error("vofe_gftt_host: h_n_out / h_corners_xy is NULL"); return VOFE_E_ARG; }
    *h_n_out = 0;
    if (!(quality > 0) || !(min_distance >= 0) || max_corners < 0) {
        vofe_set_error("vofe_gftt_host: need qualityLevel > 0, minDistance >= 0, maxCorners >= 0");
        return VOFE_E_ARG;
    }
    int rc = corner_args("vofe_gftt_host", ctx, d_img, height, width, pitch, block_size, aperture);
    if (rc) return rc;
    if (capacity <= 0 || !d_corners_xy) { vofe_set_error("vofe_gftt_host: bad output buffer"); return VOFE_E_ARG; }
    VOFE_BIND(ctx);
    return vofe_gftt_impl(ctx, d_img, 1, height, width, pitch, 0, max_corners, quality, min_distance, d_mask, harris, k,
                          d_corners_xy, capacity, h_n_out, h_corners_xy, false, block_size, aperture);
}

int vofe_gftt_batch(vofe_ctx* ctx, const uint8_t* d_img, int batch, int height, int width,
                    int pitch, size_t img_stride, int max_corners, double quality,
                    double min_distance, int block_size, int aperture, int harris, double k,
                    float* d_corners_xy, int capacity, int* h_n_out) {
    if (!h_n_out || batch <= 0) { vofe_set_error("vofe_gftt_batch: h_n_out is NULL or batch <= 0"); return VOFE_E_ARG; }
    for (int b = 0; b < batch; ++b) h_n_out[b] = 0;
    if (!(quality > 0) || !(min_distance >= 0) || max_corners < 0) {
        vofe_set_error("vofe_gftt_batch: need qualityLevel > 0, minDistance >= 0, maxCorners >= 0");
        return VOFE_E_ARG;
    }
    int rc = corner_args("vofe_gftt_batch", ctx, d_img, height, width, pitch, block_size, aperture);
    if (rc) return rc;
    if (capacity < 0 || (capacity > 0 && !d_corners_xy)) { vofe_set_error("vofe_gftt_batch: bad output buffer"); return VOFE_E_ARG; }
    VOFE_BIND(ctx);
    return vofe_gftt_impl(ctx, d_img, batch, height, width, pitch, img_stride, max_corners, quality, min_distance, nullptr,
                          harris, k, d_corners_xy, capacity, h_n_out, nullptr, false, block_size, aperture);
}

int vofe_new_feature_mask(vofe_ctx* ctx, const float* d_features_xy, int n_features,
                          const float* d_existing_xy, int n_existing, double threshold, uint8_t* d_mask) {
    if (!ctx) { vofe_set_error("vofe_new_feature_mask: ctx is NULL"); return VOFE_E_ARG; }
    if (n_features < 0 || n_existing < 0) { vofe_set_error("vofe_new_feature_mask: negative count"); return VOFE_E_ARG; }
    if ((n_features > 0 && (!d_features_xy || !d_mask)) || (n_existing > 0 && !d_existing_xy)) {
        vofe_set_error("vofe_new_feature_mask: NULL buffer");
        return VOFE_E_ARG;
    }
    VOFE_BIND(ctx);
    return vofe_launch_new_feature_mask(ctx, d_features_xy, n_features, d_existing_xy, n_existing, nullptr, 0, threshold, d_mask);
}

int vofe_new_feature_mask2(vofe_ctx* ctx, const float* d_features_xy, int n_features, const float* d_existing_a_xy, int n_a,
                           const float* d_existing_b_xy, int n_b, double threshold, uint8_t* d_mask) {
    if (!ctx) { vofe_set_error("vofe_new_feature_mask2: ctx is NULL"); return VOFE_E_ARG; }
    if (n_features < 0 || n_a < 0 || n_b < 0) { vofe_set_error("vofe_new_feature_mask2: negative count"); return VOFE_E_ARG; }
    if ((n_features > 0 && (!d_features_xy || !d_mask)) || (n_a > 0 && !d_existing_a_xy) || (n_b > 0 && !d_existing_b_xy)) {
        vofe_set_error("vofe_new_feature_mask2: NULL buffer");
        return VOFE_E_ARG;
    }
    VOFE_BIND(ctx);
    return vofe_launch_new_feature_mask(ctx, d_features_xy, n_features, d_existing_a_xy, n_a, d_existing_b_xy, n_b, threshold, d_mask);
}

int vofe_candidate_filter(vofe_ctx* ctx, const float* d_tracked_xy, const uint8_t* d_status,
                          const void* d_first_xy, int first_is_f64, const double* d_transform, int transform_dim,
                          int n, double dist_min, double dist_max, double dist_triangulate, float* d_tracked_out,
                          void* d_first_out, double* d_transform_out, uint8_t* d_triangulate_out,
                          int* d_count_out, int* h_count_out) {
    if (h_count_out) *h_count_out = 0;
    if (!ctx) { vofe_set_error("vofe_candidate_filter: ctx is NULL"); return VOFE_E_ARG; }
    if (n < 0) { vofe_set_error("vofe_candidate_filter: negative count"); return VOFE_E_ARG; }
    if (!d_count_out) { vofe_set_error("vofe_candidate_filter: d_count_out is NULL"); return VOFE_E_ARG; }
    if (d_transform && (transform_dim < 1 || transform_dim > 16)) {
        vofe_set_error("vofe_candidate_filter: transform_dim must be 1..16, got %d", transform_dim);
        return VOFE_E_ARG;
    }
    if (n > 0 && (!d_tracked_xy || !d_status || !d_first_xy || !d_tracked_out || !d_first_out || !d_triangulate_out ||
                  (d_transform && !d_transform_out))) {
        vofe_set_error("vofe_candidate_filter: NULL buffer");
        return VOFE_E_ARG;
    }
    VOFE_BIND(ctx);
    int rc = vofe_launch_candidate_filter(ctx, d_tracked_xy, d_status, d_first_xy, first_is_f64, d_transform, transform_dim, n,
                                          dist_min, dist_max, dist_triangulate, d_tracked_out, d_first_out, d_transform_out,
                                          d_triangulate_out, d_count_out);
    if (rc || !h_count_out) return rc;
    VOFE_HIP(hipMemcpyAsync(h_count_out, d_count_out, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VOFE_HIP(hipStreamSynchronize(ctx->stream));
    return VOFE_OK;
}

int vofe_pair_filter(vofe_ctx* ctx, const float* d_new_xy, const uint8_t* d_flag, const float* d_old_xy, int n,
                     double dist_move, float* d_new_out, float* d_old_out, int* d_count_out, int* h_count_out) {
    if (h_count_out) *h_count_out = 0;
    if (!ctx) { vofe_set_error("vofe_pair_filter: ctx is NULL"); return VOFE_E_ARG; }
    if (n < 0) { vofe_set_error("vofe_pair_filter: negative count"); return VOFE_E_ARG; }
    if (!d_count_out) { vofe_set_error("vofe_pair_filter: d_count_out is NULL"); return VOFE_E_ARG; }
    if (n > 0 && (!d_new_xy || !d_flag || !d_old_xy || !d_new_out || !d_old_out)) {
        vofe_set_error("vofe_pair_filter: NULL buffer");
        return VOFE_E_ARG;
    }
    VOFE_BIND(ctx);
    int rc = vofe_launch_pair_filter(ctx, d_new_xy, d_flag, d_old_xy, n, dist_move, d_new_out, d_old_out, d_count_out);
    if (rc || !h_count_out) return rc;
    VOFE_HIP(hipMemcpyAsync(h_count_out, d_count_out, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VOFE_HIP(hipStreamSynchronize(ctx->stream));
    return VOFE_OK;
}

int vofe_triangulate_candidates(vofe_ctx* ctx, const double* d_K, const double* d_pose, const float* d_cur_xy,
                                const void* d_first_xy, int first_is_f64, const double* d_transform,
                                int transform_dim, const uint8_t* d_flag_in, int n, double alpha_min,
                                uint8_t* d_flag_out, double* d_xyz) {
    if (!ctx) { vofe_set_error("vofe_triangulate_candidates: ctx is NULL"); return VOFE_E_ARG; }
    if (n < 0) { vofe_set_error("vofe_triangulate_candidates: negative count"); return VOFE_E_ARG; }
    if (transform_dim < 12 || transform_dim > 16) {
        vofe_set_error("vofe_triangulate_candidates: transform_dim must be 12..16 (a 3x4 pose per candidate), got %d", transform_dim);
        return VOFE_E_ARG;
    }
    if (!d_K || !d_pose || (n > 0 && (!d_cur_xy || !d_first_xy || !d_transform || !d_flag_in || !d_flag_out || !d_xyz))) {
        vofe_set_error("vofe_triangulate_candidates: NULL buffer");
        return VOFE_E_ARG;
    }
    VOFE_BIND(ctx);
    return vofe_launch_triangulate(ctx, d_K, d_pose, nullptr, d_cur_xy, d_first_xy, first_is_f64, d_transform, transform_dim, d_flag_in, n,
                                   alpha_min, d_flag_out, d_xyz);
}

int vofe_candidate_commit(vofe_ctx* ctx, float* d_cand_xy, void* d_first_xy, int first_is_f64, double* d_transform,
                          int transform_dim, int n_candidates, int cap_candidates, const uint8_t* d_promote, const double* d_xyz,
                          const float* d_features_xy, const uint8_t* d_feature_mask, int n_features, const double* d_pose,
                          float* d_landmarks_xy, int n_landmarks, int cap_landmarks, float* d_track_next, float* d_new_landmarks_xy,
                          double* d_new_landmarks_xyz, int* d_counts) {
    if (!ctx) { vofe_set_error("vofe_candidate_commit: ctx is NULL"); return VOFE_E_ARG; }
    if (n_candidates < 0 || n_features < 0 || n_landmarks < 0 || cap_candidates < n_candidates || cap_landmarks < n_landmarks) {
        vofe_set_error("vofe_candidate_commit: counts must be >= 0 and within the capacities");
        return VOFE_E_ARG;
    }
    if (transform_dim < 12 || transform_dim > 16) { vofe_set_error("vofe_candidate_commit: transform_dim must be 12..16, got %d", transform_dim); return VOFE_E_ARG; }
    if (!d_counts || !d_pose || !d_track_next || (cap_candidates > 0 && (!d_cand_xy || !d_first_xy || !d_transform)) ||
        (cap_landmarks > 0 && !d_landmarks_xy) || (n_candidates > 0 && (!d_promote || !d_xyz || !d_new_landmarks_xy || !d_new_landmarks_xyz)) ||
        (n_features > 0 && (!d_features_xy || !d_feature_mask))) {
        vofe_set_error("vofe_candidate_commit: NULL buffer");
        return VOFE_E_ARG;
    }
    VOFE_BIND(ctx);
    return vofe_launch_candidate_commit(ctx, d_cand_xy, d_first_xy, first_is_f64, d_transform, transform_dim, n_candidates, cap_candidates,
                                        d_promote, d_xyz, d_features_xy, d_feature_mask, n_features, d_pose, nullptr, d_landmarks_xy, n_landmarks,
                                        cap_landmarks, d_track_next, d_new_landmarks_xy, d_new_landmarks_xyz, d_counts);
}

}  // extern "C"
