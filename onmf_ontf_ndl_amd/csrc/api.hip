// C-ABI entry points of libonmf_hip.so and the host-side control of the ONMF minibatch step
// (launch order, the sweep loop and its stopping test).  Signatures: include/onmf_hip.h.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "common.h"

#include <functional>
#include <mutex>

// ---------------------------------------------------------------------------------------------
// context helpers
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_create_error;

// Events that only order this library's own streams on one GPU: nothing on the host ever inspects them, so
// recording one need not make the GPU's writes visible system-wide (the default for an event).
#ifndef EV_ORDER_ONLY
#define EV_ORDER_ONLY (hipEventDisableTiming | hipEventDisableSystemFence)
#endif


int onmf_fail(onmf_ctx* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

int onmf_reserve(onmf_ctx* c, DevBuf& b, size_t bytes, const char* what) {
    if (bytes <= b.bytes) return ONMF_OK;
    if (b.p) {
        HIP_TRY(c, hipFree(b.p));      // synchronises the device: nothing in flight uses it after
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return onmf_fail(c, ONMF_ERR_NOMEM, "hipMalloc of %zu bytes for %s failed: %s", want, what,
                         hipGetErrorString(e));
    }
    b.bytes = want;
    return ONMF_OK;
}

int onmf_reserve_pinned(onmf_ctx* c, size_t bytes) {
    if (bytes <= c->pinned_bytes) return ONMF_OK;
    if (c->pinned) {
        HIP_TRY(c, hipHostFree(c->pinned));
        c->pinned = nullptr;
        c->pinned_bytes = 0;
    }
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    hipError_t e = hipHostMalloc(&c->pinned, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        c->pinned = nullptr;
        return onmf_fail(c, ONMF_ERR_NOMEM, "hipHostMalloc of %zu bytes failed: %s", want,
                         hipGetErrorString(e));
    }
    c->pinned_bytes = want;
    return ONMF_OK;
}


// ---------------------------------------------------------------------------------------------
// ABI
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Static scratch of the entry point that allocates nothing (onmf_update_dict_static): the
// dictionary pass's workspace for d <= 1024, r <= 256.  Every GPU the library runs on holds its own
// copy of this array, at its own address, until the process ends; host code never uses the array's
// name as an address - only what hipGetSymbolAddress returned for the device in question.
// ---------------------------------------------------------------------------------------------
constexpr int STATIC_DICT_MAX_D = 1024;
constexpr int STATIC_DICT_MAX_R = 256;
constexpr size_t STATIC_SCRATCH_FLOATS =
    4 * (size_t)STATIC_DICT_MAX_D * STATIC_DICT_MAX_R + (size_t)STATIC_DICT_MAX_R * STATIC_DICT_MAX_R;   // 4.25 MB
static __device__ __align__(256) float scratch[STATIC_SCRATCH_FLOATS];
static float* g_scratch_ptr[LdsLimit::MAX_DEV];     // per device, filled by onmf_create
// Calls that use the one array of a GPU are ordered behind each other, whatever stream or context they come
// from: an event per device (created by onmf_create, so the entry point itself creates nothing), recorded
// behind every call and waited for by the next.  A capturing stream is left alone (an event recorded outside
// a capture has no place in it): under capture the caller keeps such calls apart.
static hipEvent_t g_scratch_event[LdsLimit::MAX_DEV];
static bool g_scratch_event_recorded[LdsLimit::MAX_DEV];
static std::mutex g_scratch_mutex;

// with `device` current
static int static_scratch_for_device(int device, float** out) {
    *out = nullptr;
    if (device < 0 || device >= LdsLimit::MAX_DEV)
        return onmf_fail(nullptr, ONMF_ERR_ARG, "device %d: the static scratch table covers %d devices", device,
                         LdsLimit::MAX_DEV);
    if (!g_scratch_ptr[device]) {
        float* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return onmf_fail(nullptr, ONMF_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device,
                             hipGetErrorString(e));
        g_scratch_ptr[device] = ptr;
    }
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_event[device] &&
            hipEventCreateWithFlags(&g_scratch_event[device], EV_ORDER_ONLY) != hipSuccess) {
            g_scratch_event[device] = nullptr;
            return onmf_fail(nullptr, ONMF_ERR_HIP, "creating the static scratch's event on device %d failed", device);
        }
    }
    *out = g_scratch_ptr[device];
    return ONMF_OK;
}

extern "C" {

int onmf_abi_version(void) { return ONMF_ABI_VERSION; }

double onmf_host_lambda_max(const float* S_host, int n) { return host_lambda_max(S_host, n); }

int onmf_create(int device, onmf_ctx** out) {
    if (!out) return onmf_fail(nullptr, ONMF_ERR_ARG, "onmf_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return onmf_fail(nullptr, ONMF_ERR_HIP, "onmf_create: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count)
        return onmf_fail(nullptr, ONMF_ERR_ARG, "onmf_create: device %d out of range (%d devices)", device, count);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess)
        return onmf_fail(nullptr, ONMF_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(guard.err));
    float* sp = nullptr;
    int rc = static_scratch_for_device(device, &sp);
    if (rc != ONMF_OK) return rc;
    onmf_ctx* c = new (std::nothrow) onmf_ctx();
    if (!c) return onmf_fail(nullptr, ONMF_ERR_NOMEM, "onmf_create: out of host memory");
    c->device = device;
    c->static_scratch = sp;
    // a few words of host memory that kernels can write (event counts the host reads without waiting)
    void* he = nullptr;
    void* hed = nullptr;
    if (hipHostMalloc(&he, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&hed, he, 0) != hipSuccess) {
        if (he) (void)hipHostFree(he);
        delete c;
        return onmf_fail(nullptr, ONMF_ERR_NOMEM, "onmf_create: no mapped host memory");
    }
    memset(he, 0, 64);
    c->host_events = (int*)he;
    c->host_events_dev = (int*)hed;
    *out = c;
    return ONMF_OK;
}

int onmf_destroy(onmf_ctx* c) {
    if (!c) return ONMF_OK;
    DeviceGuard guard(c->device);
    (void)hipDeviceSynchronize();
    DevBuf* bufs[] = {&c->slab, &c->cm, &c->gram, &c->hbuf[0], &c->hbuf[1], &c->stats, &c->small,
                      &c->qt, &c->sync, &c->idx, &c->dbuf, &c->slab_side, &c->screen, &c->ddelta, &c->slab_prep,
                      &c->gstack, &c->rslab, &c->gram64, &c->cm64, &c->lars_ws, &c->lars_meta, &c->lars_overflow, &c->mt};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->rec_pinned) (void)hipHostFree(c->rec_pinned);
    if (c->host_events) (void)hipHostFree(c->host_events);
    for (hipEvent_t ev : c->ev) (void)hipEventDestroy(ev);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_rec) (void)hipEventDestroy(c->ev_rec);
    if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->prep) (void)hipStreamDestroy(c->prep);
    if (c->ev_prep_fork) (void)hipEventDestroy(c->ev_prep_fork);
    if (c->ev_prep_join) (void)hipEventDestroy(c->ev_prep_join);
    delete c;
    return ONMF_OK;
}

const char* onmf_last_error(onmf_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int onmf_set_option(onmf_ctx* c, int key, int value) {
    if (!c) return ONMF_ERR_ARG;
    switch (key) {
        case ONMF_OPT_PROFILE: c->profile = value; return ONMF_OK;
        case ONMF_OPT_EXACT_STOP: c->exact_stop = value; return ONMF_OK;
        case ONMF_OPT_SWEEP_KERNEL: c->sweep_kernel = value; return ONMF_OK;
        case ONMF_OPT_OVERLAP_DICT: c->overlap_dict = value; return ONMF_OK;
        case ONMF_OPT_SCREEN_STOP: c->screen_stop = value; return ONMF_OK;
        case ONMF_OPT_DICT_KERNEL: c->dict_kernel = value; return ONMF_OK;
        case ONMF_OPT_PERSISTENT_SWEEPS:
            c->persistent_sweeps = value;
            c->persistent_hold = 0;
            return ONMF_OK;
        case ONMF_OPT_DEVICE_RESOLVE: c->device_resolve = value; return ONMF_OK;
        case ONMF_OPT_LARS_SLOTS: c->lars_slots = value < 0 ? 0 : value; return ONMF_OK;
        case ONMF_OPT_DEFER_VERDICT: c->defer_verdict = value; return ONMF_OK;
        case ONMF_OPT_GEMM_TAIL: c->gemm_tail = value < 0 ? 0 : value; return ONMF_OK;
        case ONMF_OPT_VERDICT_LAG3: c->lag3 = value; return ONMF_OK;
        case ONMF_OPT_SPIN_LIMIT_MS:
            c->spin_limit = value == -2 ? SPIN_FORCE_TIMEOUT
                                        : value < 0 ? -1LL : (long long)value * 100000LL;      // 100 MHz ticks
            return ONMF_OK;
        default: return onmf_fail(c, ONMF_ERR_ARG, "onmf_set_option: unknown key %d", key);
    }
}

int onmf_get_counter(onmf_ctx* c, int which, long long* value) {
    if (!c || !value) return ONMF_ERR_ARG;
    switch (which) {
        case ONMF_CNT_SWEEP_TIMEOUTS: *value = c->n_sweep_timeouts; return ONMF_OK;
        case ONMF_CNT_DICT_TIMEOUTS:
            // (passes redone by the host + passes redone on the device, the latter counted by the kernel
            // that redid them: a pass still in flight is not in yet)
            *value = c->n_dict_timeouts + (c->host_events ? *(volatile int*)c->host_events : 0);
            return ONMF_OK;
        case ONMF_CNT_FALLBACK_SOLVES: *value = c->n_fallback_solves; return ONMF_OK;
        case ONMF_CNT_DEVICE_RESOLVES: *value = c->n_device_resolves; return ONMF_OK;
        case ONMF_CNT_LARS_RECODED:
            *value = c->host_events ? ((volatile int*)c->host_events)[ONMF_LARS_HOST_WORDS + 2] : 0;
            return ONMF_OK;
        case ONMF_CNT_SWEEPS_TOTAL: *value = c->sweeps_total; return ONMF_OK;
        case ONMF_CNT_DEFERRED_REDONE: *value = c->n_deferred_redone; return ONMF_OK;
        case ONMF_CNT_LAG3_REDONE: *value = c->n_lag3_redone; return ONMF_OK;
        case ONMF_CNT_LARS_SLOTS_USED:
            *value = c->host_events ? ((volatile int*)c->host_events)[ONMF_LARS_HOST_WORDS] : 0;
            return ONMF_OK;
        default: return onmf_fail(c, ONMF_ERR_ARG, "onmf_get_counter: unknown counter %d", which);
    }
}

int onmf_get_profile(onmf_ctx* c, double* sweep_ms, int* sweep_launches) {
    if (!c) return ONMF_ERR_ARG;
    ONMF_ENTER(c);
    double total = 0.0;
    const int pairs = c->ev_used / 2;
    for (int i = 0; i < pairs; ++i) {
        float ms = 0.f;
        HIP_TRY(c, hipEventSynchronize(c->ev[2 * i + 1]));
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]));
        total += ms;
    }
    if (sweep_ms) *sweep_ms = total;
    if (sweep_launches) *sweep_launches = pairs;
    c->ev_used = 0;
    return ONMF_OK;
}

}  // extern "C"

static int profile_mark(onmf_ctx* c, hipStream_t s) {
    if (!c->profile) return ONMF_OK;
    if (c->ev_used >= (int)c->ev.size()) {
        hipEvent_t e;
        HIP_TRY(c, hipEventCreate(&e));
        c->ev.push_back(e);
    }
    HIP_TRY(c, hipEventRecord(c->ev[c->ev_used++], s));
    return ONMF_OK;
}

// The reference's loop test is `dist > stopping_diff` with dist = s1 / s0 evaluated in floating
// point (src/onmf.py:252, :265): NaN (0/0) ends the loop, +inf (x/0) continues it.
static bool keep_sweeping(double dist, double tol) { return dist > tol; }

// Full stopping test of one sweep: dist = sigma_max(Hnew - Hold) / sigma_max(Hold) through the two
// r x r Gram matrices (MFMA products) and their top eigenvalues (host, double).  Synchronises s.
// With ub_old > 0 (a rigorous upper bound of sigma_max(Hold)) and tol > 0 the cheap half is tried
// first: only the Gram of the difference is formed, and the Lanczos run on it stops as soon as a
// Ritz value (a lower bound of its top eigenvalue) proves dist > tol; *dist is then that bound.
// *sigma_new_ub (optional) receives an upper bound of sigma_max(Hnew) when both halves ran
// (sigma_max(Hold) + sigma_max(Hnew - Hold), widened for the fp32 rounding of the Grams), else -1.
static int exact_dist(onmf_ctx* c, int r, int b, const float* Hnew, const float* Hold, hipStream_t s,
                      double* dist, double ub_old = -1.0, double tol = -1.0, double* sigma_new_ub = nullptr) {
    if (sigma_new_ub) *sigma_new_ub = -1.0;
    const size_t rb = (size_t)r * b;
    const size_t rr = (size_t)r * r;
    if (2 * r <= 128) {
        // Small r: the difference and the old code are stacked into one (2r x b) matrix whose Gram - ONE
        // 128 x 128 output tile, one product launch and one reduction - holds both Gram matrices on its
        // diagonal; one copy, one wait, then the two eigenvalues (the cheap half first, as below).
        const int r2 = 2 * r;
        ONMF_TRY(onmf_reserve(c, c->gstack, 2 * rb * sizeof(float), "stacked stop-test operands"));
        ONMF_TRY(onmf_reserve(c, c->small, (size_t)r2 * r2 * sizeof(float), "stop-test Grams"));
        ONMF_TRY(onmf_reserve_pinned(c, ((size_t)r2 * r2 + 2 * rr) * sizeof(float)));
        float* St = (float*)c->gstack.p;
        float* G = (float*)c->small.p;
        ONMF_TRY(launch_sub_stack(St, Hnew, Hold, rb, s));
        ONMF_TRY(launch_gemm(c, false, false, r2, r2, b, St, b, St, b, 1.f, 0.f, nullptr, G, r2, s));
        HIP_TRY(c, hipMemcpyAsync(c->pinned, G, (size_t)r2 * r2 * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        const float* hg = (const float*)c->pinned;
        float* g1 = (float*)c->pinned + (size_t)r2 * r2;      // the two diagonal blocks, contiguous
        float* g0 = g1 + rr;
        for (int i = 0; i < r; ++i) {
            std::memcpy(g1 + (size_t)i * r, hg + (size_t)i * r2, r * sizeof(float));
            std::memcpy(g0 + (size_t)i * r, hg + (size_t)(r + i) * r2 + r, r * sizeof(float));
        }
        if (ub_old > 0.0 && tol > 0.0) {
            const double need = (tol * ub_old) * (tol * ub_old) * (1.0 + 2e-4);   // margin: fp32 Gram rounding
            const double th = host_lambda_max_enough(g1, r, need);
            if (th >= need) {
                *dist = std::sqrt(th) / ub_old;
                return ONMF_OK;
            }
        }
        const double l1 = host_lambda_max(g1, r);
        const double l0 = host_lambda_max(g0, r);
        *dist = std::sqrt(std::max(l1, 0.0)) / std::sqrt(std::max(l0, 0.0));
        if (sigma_new_ub && l0 > 0.0 && l1 >= 0.0) *sigma_new_ub = (std::sqrt(l0) + std::sqrt(l1)) * (1.0 + 1e-5);
        return ONMF_OK;
    }
    ONMF_TRY(onmf_reserve(c, c->dbuf, rb * sizeof(float), "code difference"));
    ONMF_TRY(onmf_reserve(c, c->small, 2 * rr * sizeof(float), "stop-test Grams"));
    ONMF_TRY(onmf_reserve_pinned(c, 2 * rr * sizeof(float)));
    float* D = (float*)c->dbuf.p;
    float* G1 = (float*)c->small.p;
    float* G0 = G1 + rr;
    const float* hg = (const float*)c->pinned;
    ONMF_TRY(launch_sub(D, Hnew, Hold, rb, s));
    ONMF_TRY(launch_gemm(c, false, false, r, r, b, D, b, D, b, 1.f, 0.f, nullptr, G1, r, s));
    if (ub_old > 0.0 && tol > 0.0) {
        HIP_TRY(c, hipMemcpyAsync(c->pinned, G1, rr * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        const double need = (tol * ub_old) * (tol * ub_old) * (1.0 + 2e-4);   // margin: fp32 Gram rounding
        const double th = host_lambda_max_enough(hg, r, need);
        if (th >= need) {
            *dist = std::sqrt(th) / ub_old;
            return ONMF_OK;
        }
    }
    ONMF_TRY(launch_gemm(c, false, false, r, r, b, Hold, b, Hold, b, 1.f, 0.f, nullptr, G0, r, s));
    HIP_TRY(c, hipMemcpyAsync(c->pinned, G1, 2 * rr * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const double l1 = host_lambda_max(hg, r);
    const double l0 = host_lambda_max(hg + rr, r);
    *dist = std::sqrt(std::max(l1, 0.0)) / std::sqrt(std::max(l0, 0.0));
    if (sigma_new_ub && l0 > 0.0 && l1 >= 0.0) *sigma_new_ub = (std::sqrt(l0) + std::sqrt(l1)) * (1.0 + 1e-5);
    return ONMF_OK;
}

static int sparse_code_impl(onmf_ctx* c, int d, int r, int b, const float* X, int ldx,
                            const float* W, const float* H0, uint64_t seed, float alpha,
                            int sub_iter, float stopping_diff, int coder_mode, float* H,
                            int* sweeps_done, float* dist_trace, hipStream_t s,
                            const std::function<int()>* after = nullptr, bool* after_valid = nullptr,
                            bool second_try = false, const std::function<int()>* behind_sweeps = nullptr,
                            const std::function<int()>* behind_products = nullptr, float radius = -1.f,
                            const std::function<int()>* defer_redo = nullptr) {
    // `defer_redo` (not null: the caller asked for no host wait): the solve returns with everything enqueued
    // and the verdict records unread; onmf_settle_pending reads them at the next entry point and, if they
    // are not conclusive, runs the solve again the ordinary way and then *defer_redo (what `after` did).
    // `after`: work that consumes H (the step's aggregate products).  When the solve is enqueued as a
    // whole, it is enqueued too BEFORE the host waits for the verdicts, so the GPU does not idle
    // through that round trip; *after_valid tells the caller whether that run stands (it does not if
    // the host had to finish the solve: then the caller runs the work again).
    // `behind_products`: other streams' work that the solve does not depend on, enqueued right after the
    // solve's own first kernels are (the host's enqueueing is what a short step waits for).
    // `behind_sweeps`: work that must not run beside the sweeps but needs nothing of them (the step's
    // dictionary pass); it is posted right behind the last sweep launch, ahead of the verdict kernels.
    // `radius` > 0: the reference's r argument (src/onmf.py:260-263), which shortens the first row's first
    // move (see launch_sweep_radius_first); the stopping test is then evaluated per sweep.
    if (after_valid) *after_valid = false;
    if (d <= 0 || r <= 0 || b <= 0 || !X || !W || !H || ldx < b || sub_iter < 0)
        return onmf_fail(c, ONMF_ERR_ARG, "sparse_code: bad arguments (d=%d r=%d b=%d ldx=%d)", d, r, b, ldx);
    const bool within_radius = radius > 0.f;
    if (within_radius && (coder_mode != ONMF_CODER_DAMPED || (size_t)r * 256 > 160 * 1024))
        return onmf_fail(c, ONMF_ERR_ARG, "sparse_code: a radius goes with the damped coder and r <= 640 (r=%d)", r);
    const size_t rb = (size_t)r * b;
    ONMF_TRY(onmf_reserve(c, c->gram, (size_t)r * r * sizeof(float), "Gm"));
    ONMF_TRY(onmf_reserve(c, c->cm, rb * sizeof(float), "Cm"));
    ONMF_TRY(onmf_reserve(c, c->hbuf[0], rb * sizeof(float), "code buffer"));
    ONMF_TRY(onmf_reserve(c, c->hbuf[1], rb * sizeof(float), "code buffer"));
    const int max_groups = (b + 15) / 16 + 64;
    ONMF_TRY(onmf_reserve(c, c->stats, (size_t)max_groups * 2 * sizeof(double), "sweep statistics"));
    // (large enough for everything a solve copies to the host, the full test's Gram matrices included:
    // the staging area must not move while the verdict records in it are being read)
    ONMF_TRY(onmf_reserve_pinned(c, std::max({(size_t)max_groups * 2 * sizeof(double),
                                              (size_t)ScreenLayout::TOTAL * sizeof(float),
                                              (size_t)6 * r * r * sizeof(float)})));
    float* Gm = (float*)c->gram.p;
    float* Cm = (float*)c->cm.p;
    // Gm = W^T W,  Cm = W^T X   (src/onmf.py:242-243).  Gm, the initial code and the screen state are
    // small and independent of the big product: they go to a stream of their own (default priority)
    // and run beside it instead of in front of it.
    // Sweep i reads buf[i % 3] and writes buf[(i + 1) % 3]: three buffers, so that a sweep that ran
    // ahead of the stopping test's verdict has overwritten neither side of the tested pair.  The
    // caller's H is one of them, placed so that a solve that runs all sub_iter sweeps ends in it.
    float* buf[3];
    buf[sub_iter % 3] = H;
    buf[(sub_iter + 1) % 3] = (float*)c->hbuf[0].p;
    buf[(sub_iter + 2) % 3] = (float*)c->hbuf[1].p;
    hipStream_t ps = s;
    // (a stream of its own only where the big product is long enough to hide something behind: forking
    // and joining a stream costs two trips through the command processor, about 10 us each.  Measured
    // without it: d=100 r=16 b=1000 0.318 -> 0.294 ms per step, d=1200 r=100 b=8192 0.530 -> 0.506,
    // d=1024 r=64 b=8192 0.452 -> 0.435, d=1024 r=256 b=65536 1.586 -> 1.633.)
    if (c->overlap_dict && 2.0 * r * (double)b * d >= 8e9) {
        if (!c->prep) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->prep, hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_prep_fork, EV_ORDER_ONLY));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_prep_join, EV_ORDER_ONLY));
        }
        ps = c->prep;
        HIP_TRY(c, hipEventRecord(c->ev_prep_fork, s));
        HIP_TRY(c, hipStreamWaitEvent(ps, c->ev_prep_fork, 0));
    }
    // Workgroups go to the chip's 32 shader engines (8 XCDs x 4) in turn, and a workgroup waits for a
    // slot in ITS engine.  The big product's workgroups are whole rounds of the chip at the headline
    // shape (2 x 512 tiles on 512 slots): a single slot held by another stream's workgroup - the
    // dictionary pass - then costs it a further round, and that workgroup itself, arriving later, is
    // placed only when the product's queue for its engine has drained, i.e. at the product's end,
    // where it keeps a workgroup of the sweeps waiting.  So the main launch is given one workgroup
    // fewer per engine (32 in all): its last round leaves a slot free in every engine, which is where
    // the dictionary pass starts.  The columns taken off are computed at the head of the side stream,
    // contraction split into many short workgroups, while the chip is still filling.
    int tail_cols = 0;
    {
        const int tiles_m = (r + 127) / 128, tiles_n = (b + 127) / 128;
        const int cols = 128 * std::max(1, 32 / tiles_m);
        if (ps != s && tiles_m * tiles_n >= 512 && cols <= b / 4) tail_cols = cols;
    }
    int prep_rc = ONMF_OK;
    // one stream: [W^T X | W^T W] as ONE launch and one reduction further down (they share the left operand)
    const bool one_product = ps == s;
    if (tail_cols)   // [W^T X_tail | W^T W]: one launch and one reduction for both
        prep_rc = launch_gemm_pair(c, true, true, r, d, W, r, X + (b - tail_cols), ldx, tail_cols, W, r, r, 1.f, 0.f,
                                   nullptr, Cm + (b - tail_cols), b, 0.f, nullptr, Gm, r, ps, &c->slab_prep);
    else if (!one_product)
        prep_rc = launch_gemm(c, true, true, r, r, d, W, r, W, r, 1.f, 0.f, nullptr, Gm, r, ps, &c->slab_prep);
    const double tol = (double)stopping_diff;
    const double rank_cap = (double)std::min(r, b);
    const int exact_cd = (coder_mode == ONMF_CODER_EXACT_CD);
    const bool screened = !exact_cd && !within_radius && tol >= 0.0 && !c->exact_stop && c->screen_stop &&
                          c->sweep_kernel != 1 && sweep_tile_covers(r, b) &&
                          sub_iter <= SCREEN_MAX_SWEEPS && keep_sweeping(1.0, tol);
    // one stream, start code drawn here, screened test: the draw rides in the screen's set-up launch (a
    // short step is a chain of launches, and each one costs the host and the device about 5 us)
    const bool fill_with_init = one_product && screened && H0 == nullptr && rb > 0;
    if (prep_rc == ONMF_OK) {
        if (H0) {
            if (H0 != buf[0] &&
                hipMemcpyAsync(buf[0], H0, rb * sizeof(float), hipMemcpyDeviceToDevice, ps) != hipSuccess)
                prep_rc = onmf_fail(c, ONMF_ERR_HIP, "sparse_code: copy of H0 failed");
        } else if (!fill_with_init) {
            prep_rc = launch_fill_uniform(buf[0], rb, seed, ps);
        }
    }
    int init_rc = ONMF_OK;
    const bool probes_new = c->screen_r != r;     // (the set-up below records r)
    if (screened) {
        // The screen's set-up goes in FRONT of the big product: it is needed before the sweeps anyway,
        // and the few microseconds it takes let the small kernels on the other streams (Gm, the start
        // code, the head of the dictionary pass) take their places on the chip before the product's
        // workgroups fill it - behind them they would wait for its first round to finish.
#ifdef ONMF_TRACE_WG
        // (diagnostic build: room behind the screen state for workgroup 0's time marks, 16 per sweep)
        init_rc = onmf_reserve(c, c->screen, ((size_t)ScreenLayout::TOTAL + 2 * 16 * SCREEN_MAX_SWEEPS + 64) * sizeof(float),
                               "stop-test screen");
#else
        init_rc = onmf_reserve(c, c->screen, (size_t)ScreenLayout::TOTAL * sizeof(float), "stop-test screen");
#endif
        if (init_rc == ONMF_OK)
            init_rc = launch_screen_init(c, (float*)c->screen.p, r, s, sub_iter + 1, fill_with_init ? buf[0] : nullptr,
                                         fill_with_init ? rb : 0, seed);
    }
    int cm_rc = init_rc;
    if (cm_rc == ONMF_OK) {
        if (one_product)
            cm_rc = launch_gemm_pair(c, true, true, r, d, W, r, X, ldx, b, W, r, r, 1.f, 0.f, nullptr, Cm, b, 0.f, nullptr,
                                     Gm, r, s, nullptr);
        else
            cm_rc = launch_gemm(c, true, true, r, b - tail_cols, d, W, r, X, ldx, 1.f, 0.f, nullptr, Cm, b, s, nullptr,
                                c->gemm_tail);
    }
    if (behind_products && cm_rc == ONMF_OK && prep_rc == ONMF_OK) cm_rc = (*behind_products)();
    if (ps != s) {                       // always rejoin, also on an error path
        hipError_t e1 = hipEventRecord(c->ev_prep_join, ps);
        hipError_t e2 = hipStreamWaitEvent(s, c->ev_prep_join, 0);
        if (prep_rc == ONMF_OK && cm_rc == ONMF_OK) {
            HIP_TRY(c, e1);
            HIP_TRY(c, e2);
        }
    }
    ONMF_TRY(prep_rc);
    ONMF_TRY(cm_rc);
    const double nan = std::numeric_limits<double>::quiet_NaN();
    int done = 0;

    if (screened) {
        // All sweeps are enqueued without host round trips - as ONE launch that runs them all when
        // the grid is resident as a whole (one workgroup per CU), else one launch per sweep; they
        // evaluate the stopping test themselves (two sweeps back) and end when it says so.
        float* base = (float*)c->screen.p;
        // Ranks whose sweeps carry the Collatz-Wielandt sums (r <= 128) need no launch boundary for a bound
        // of sigma_max(H): their probes are refreshed behind the solve, on its LAST move - the late
        // sweeps are where the test gets close, and the next minibatch's moves resemble this one's sweep
        // for sweep.  The others cut the first launch after the second sweep for both.
        const int cw_rows = screen_cw_rows(r);
        const bool refresh_probes = (c->screen_refresh_due || probes_new) && cw_rows == 0;
        double min_margin = 1e300;       // smallest (lower bound / tol) among the "continue" verdicts
        bool undecided = false, posted = false, chain_clean = true;   // chain = one enqueue of sweeps + verdicts
        bool resolved_on_device = false;
        int full_evals = 0;                // full evaluations of this solve: rounds on the device + tests on the host
        if (c->cu_count == 0) {
            int cus = 0;
            HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            c->cu_count = cus;
        }
        // (after a timeout the one-launch mode rests for persistent_backoff solves, then gets another try)
        const bool fits = sweep_tile_workgroups(c, r, b, true) <= c->cu_count;
        if (fits && c->persistent_sweeps && c->persistent_hold > 0) {
            --c->persistent_hold;
            ++c->n_fallback_solves;
        }
        const bool persistent = c->persistent_sweeps && c->persistent_hold == 0 && fits && !second_try;
        int start = 0;
        while (start < sub_iter) {
            // (with one launch per sweep every count is complete before the launch that reads it starts:
            // the forced timeout of the tests applies to one-launch solves only)
            const long long spin = (!persistent && c->spin_limit == SPIN_FORCE_TIMEOUT) ? -1LL : c->spin_limit;
            ScreenDev sc = {base, start, (float)tol, (float)rank_cap, spin /* 2 s by default */};
            // Kernels with a control wave may act on a test three sweeps later instead of two - their tile waves
            // then never wait for a verdict.  The price: at an inconclusive test the stopping sweep's input is
            // gone (overwritten by the second sweep that ran ahead), so the host cannot evaluate the test in
            // full and runs the solve again in the plain form instead.  Only where that is possible (the start
            // code can be had again) and unlikely (the previous solve needed no host decision).
            const bool start_again = H0 == nullptr || (H0 != buf[0] && H0 != buf[1] && H0 != buf[2]);
            const bool lag3 = persistent && c->lag3 && !second_try && start_again && !c->screen_last_undecided &&
                              sub_iter >= 4 && sweep_has_control_wave_at(c, r, b);
            sc.lag = lag3 ? 3 : 2;
            int i = start;
            // When the probes are due for a power step, it is taken on the move of the SECOND sweep (the
            // first one is the jump away from the random start; the later moves resemble each other
            // and the second), so the launch is cut after that sweep.
            const int refresh_after = sub_iter >= 3 ? 1 : 0;
            while (i < sub_iter) {
                int count = persistent ? sub_iter - i : 1;
                const bool cut = start == 0 && refresh_probes && i <= refresh_after;
                if (cut) count = std::min(count, refresh_after + 1 - i);
                int groups = 0;
                ONMF_TRY(profile_mark(c, s));
                ONMF_TRY(launch_sweep(c, r, b, Gm, Cm, buf, alpha, 0, i, count, (double*)c->stats.p, &groups, s, &sc));
                ONMF_TRY(profile_mark(c, s));
                i += count;
                if (cut && i == refresh_after + 1) {
                    ONMF_TRY(launch_screen_refresh(c, base, r, b, buf[i % 3], buf[(i - 1) % 3], s));
                    // and a tight bound of sigma_max of the code the next sweep starts from (>= 0 now)
                    if (i < sub_iter) ONMF_TRY(launch_screen_cw(c, base, r, b, buf[i % 3], i, s));
                }
            }
            // (the first enqueue of a solve does not run ahead when the previous solve needed the host:
            // chances are this one does too, and the work enqueued here would run twice; once the host
            // has settled a sweep, what follows is short and usually conclusive, so it does again)
            // Small r, where borderline tests are frequent and a step is short (opt-in,
            // ONMF_OPT_DEVICE_RESOLVE): when the previous solve needed a full evaluation, one round of
            // device-side resolution is enqueued behind the launch (settle the first inconclusive verdict
            // in place, go on sweeping from there) before the host looks at anything.  It costs three
            // near-empty launches when it turns out not to be needed.
            // As many rounds as the previous solve had full evaluations (tests come close on several sweeps
            // in a row where the spectrum of the moves is flat), four at most; a round that finds nothing to
            // settle is three near-empty launches.
            // By default (ONMF_OPT_DEVICE_RESOLVE < 0) only where it was measured to pay: r <= 32, where a round
            // settles two sweeps, and behind a solve that stopped early - such a solve needs a full evaluation
            // at its stop whatever the probes are like (no upper bound is tight enough to end the loop), while
            // one that ran all its sweeps only needed one because the lower bounds were loose, which the
            // probe refresh behind it repairs.
            const bool dev_on = c->device_resolve > 0 ? screen_resolve_covers(r)
                                                      : (c->device_resolve < 0 && r <= 32 && c->screen_last_stopped_early);
            const int dev_rounds = (start == 0 && persistent && dev_on && c->screen_last_undecided)
                                       ? std::min(4, std::max(1, c->screen_last_evals)) : 0;
            const bool ahead = after && (rb & 3) == 0 && (start > 0 || !c->screen_last_undecided || dev_rounds > 0);
            posted = false;
            if (ahead && behind_sweeps) ONMF_TRY((*behind_sweeps)());
            // a sweep evaluates the test of the one two sweeps back: the last two are still open
            // (the records, and behind them the resolution's own words NEXT and NRES, reach the host through
            // memory the last of these kernels writes)
            if (!c->rec_pinned) {
                HIP_TRY(c, hipHostMalloc((void**)&c->rec_pinned, (size_t)(4 * SCREEN_MAX_SWEEPS + 4) * sizeof(float),
                                         hipHostMallocMapped));
                HIP_TRY(c, hipHostGetDevicePointer((void**)&c->rec_pinned_dev, c->rec_pinned, 0));
            }
            ONMF_TRY(launch_screen_final(c, base, std::max(start, sub_iter - 2), sub_iter - 1, (float)tol,
                                         (float)rank_cap, s, nullptr, cw_rows, c->rec_pinned_dev));
            for (int rd = 0; rd < dev_rounds; ++rd) {
                const int* next = reinterpret_cast<const int*>(base) + ScreenLayout::NEXT;
                ONMF_TRY(launch_screen_resolve(c, base, r, b, buf, sub_iter, (float)tol, s));
                const ScreenDev sc2 = {base, 0, (float)tol, (float)rank_cap, sc.spin_limit, next, sub_iter};
                int groups = 0;
                ONMF_TRY(profile_mark(c, s));
                ONMF_TRY(launch_sweep(c, r, b, Gm, Cm, buf, alpha, 0, 0, sub_iter, (double*)c->stats.p, &groups, s, &sc2));
                ONMF_TRY(profile_mark(c, s));
                ONMF_TRY(launch_screen_final(c, base, 0, sub_iter - 1, (float)tol, (float)rank_cap, s, next, cw_rows,
                                             c->rec_pinned_dev));
            }
            if (ahead) {
                // the host waits for the records only; the follow-up work is already behind them
                if (!c->ev_rec) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_rec, hipEventDisableTiming));
                HIP_TRY(c, hipEventRecord(c->ev_rec, s));
                ONMF_TRY(launch_screen_finalize(c, base, sub_iter, buf, H, rb, s));
                ONMF_TRY((*after)());
                posted = true;
                const bool start_intact = H0 == nullptr || (H0 != buf[0] && H0 != buf[1] && H0 != buf[2]);
                if (defer_redo && start == 0 && start_intact && !second_try) {
                    // not waited for: the next entry point on this context settles it
                    PendingSolve& p = c->pend;
                    p.active = true;
                    p.d = d; p.r = r; p.b = b; p.ldx = ldx; p.sub_iter = sub_iter; p.cw_rows = cw_rows;
                    p.X = X; p.W = W; p.H0 = H0; p.seed = seed; p.alpha = alpha; p.stopping_diff = stopping_diff;
                    p.H = H;
                    for (int q = 0; q < 3; ++q) p.buf[q] = buf[q];
                    p.probes_new = probes_new;
                    p.dev_rounds = dev_rounds;
                    p.s = s;
                    p.redo_after = *defer_redo;
                    if (after_valid) *after_valid = true;
                    return ONMF_OK;
                }
                HIP_TRY(c, hipEventSynchronize(c->ev_rec));
            } else {
                HIP_TRY(c, hipStreamSynchronize(s));
            }
            const float* rec = (const float*)c->rec_pinned;
            if (dev_rounds > 0) {
                const int nres = reinterpret_cast<const int*>(rec)[4 * SCREEN_MAX_SWEEPS + 1];
                if (nres > 0) {
                    c->n_device_resolves += nres;
                    // (rounds, not tests: a round may settle two sweeps)
                    full_evals += reinterpret_cast<const int*>(rec)[4 * SCREEN_MAX_SWEEPS + 3];
                    resolved_on_device = true;
                }
            }
            int next_start = sub_iter;
            bool stop = false;
            chain_clean = true;
            for (int i = start; i < sub_iter; ++i) {
                const int code = (int)rec[4 * i + 2];     // 1 continue, 2 stop, 3 inconclusive
                if (code == 0) {
                    HIP_TRY(c, hipMemcpy(&done, base + ScreenLayout::FLAG, sizeof(int), hipMemcpyDeviceToHost));
                    if (done == SCREEN_FLAG_TIMEOUT) {
                        // the grid was not resident as a whole (CUs taken or masked): one launch per sweep,
                        // which needs no co-residency, for this very solve (if its start can be had again:
                        // drawn from the seed, or still in the caller's H0) and for the next
                        // persistent_backoff solves; the one-launch mode is then tried again
                        ++c->n_sweep_timeouts;
                        c->persistent_backoff = c->persistent_backoff ? std::min(c->persistent_backoff * 2, 1 << 20) : 64;
                        c->persistent_hold = c->persistent_backoff;
                        c->err = "sparse_code: a sweep workgroup gave up waiting for the others (grid not resident "
                                 "as a whole?); solves run with one launch per sweep for a while";
                        const bool start_intact = H0 == nullptr || (H0 != buf[0] && H0 != buf[1] && H0 != buf[2]);
                        if (!second_try && start_intact)
                            return sparse_code_impl(c, d, r, b, X, ldx, W, H0, seed, alpha, sub_iter, stopping_diff,
                                                    coder_mode, H, sweeps_done, dist_trace, s, after, after_valid, true,
                                                    behind_sweeps, nullptr);
                        return onmf_fail(c, ONMF_ERR_TIMEOUT,
                                         "sparse_code: a sweep workgroup gave up waiting for the others and the "
                                         "solve's start is gone; later solves use one launch per sweep for a while");
                    }
                    return onmf_fail(c, ONMF_ERR_HIP, "sparse_code: sweep %d left no verdict", i);
                }
                done = i + 1;
                if (code == 1) {
                    if (dist_trace) dist_trace[i] = rec[4 * i + 0];
                    if (i > 0) min_margin = std::min(min_margin, (double)rec[4 * i + 0] / tol);
                    continue;
                }
                if (code == 2) {
                    if (dist_trace) dist_trace[i] = rec[4 * i + 1];
                    stop = true;
                    break;
                }
                if (i + 1 == sub_iter) {
                    // The loop ends after this sweep whatever its test says (src/onmf.py:252: i < sub_iter):
                    // nothing depends on the value, so it is not evaluated in full; the trace shows the
                    // lower bound.
                    if (dist_trace) dist_trace[i] = rec[4 * i + 0];
                    continue;
                }
                if (lag3) {
                    // the stopping sweep's input is gone (two sweeps ran ahead): the plain form, from the start
                    c->screen_last_undecided = true;
                    c->screen_refresh_due = true;
                    ++c->n_lag3_redone;
                    return sparse_code_impl(c, d, r, b, X, ldx, W, H0, seed, alpha, sub_iter, stopping_diff, coder_mode, H,
                                            sweeps_done, dist_trace, s, after, after_valid, false, behind_sweeps, nullptr);
                }
                double dist = nan;                        // bounds straddle tol: evaluate in full
                undecided = true;
                ++full_evals;
                chain_clean = false;
                // With several sweeps to go, both halves of the test are evaluated even where the cheap
                // one would do: the exact sigma_max(H) they give makes the bounds of the sweeps that
                // follow tight (H moves little from here on), which saves their host round trips.
                const bool seed_bound = i + 2 < sub_iter;
                double ub_next = -1.0;
                ONMF_TRY(exact_dist(c, r, b, buf[(i + 1) % 3], buf[i % 3], s, &dist,
                                    seed_bound ? -1.0 : (double)rec[4 * i + 3], tol, &ub_next));
                if (dist_trace) dist_trace[i] = (float)dist;
                if (!keep_sweeping(dist, tol)) { stop = true; break; }
                if (i + 2 == sub_iter) {
                    // The sweep after this one is the last, and it has run already: the sweeps were all
                    // enqueued, and an inconclusive verdict only stops those two or more sweeps behind it
                    // (a sweep evaluates the test two sweeps back).  Its own test decides nothing
                    // (src/onmf.py:252: i < sub_iter), so the solve is complete, its result in place.
                    done = sub_iter;
                    if (dist_trace) dist_trace[i + 1] = (float)nan;       // not evaluated
                    next_start = sub_iter;
                    break;
                }
                if (i + 1 < sub_iter) {
                    ONMF_TRY(launch_screen_clear_flag(c, base, i + 1, s, (float)ub_next));
                    // the probes were not good enough to decide: one power step on this very move
                    ONMF_TRY(launch_screen_refresh(c, base, r, b, buf[(i + 1) % 3], buf[i % 3], s));
                }
                next_start = i + 1;
                break;
            }
            if (stop) break;
            start = next_start;
        }
        // the next solve refreshes its probes if these left the bounds undecided or close to it
        c->screen_refresh_due = undecided || resolved_on_device || min_margin < 1.5;
        c->screen_last_undecided = undecided || resolved_on_device;
        c->screen_last_evals = full_evals;
        c->screen_last_stopped_early = done < sub_iter;
        if (cw_rows > 0 && done >= 1) {
            // ... or this one does, behind its last sweep and on that sweep's move: at once when the
            // bounds left a test undecided (or the probes are new), every fourth solve while the
            // tests are merely close
            ++c->screen_since_refresh;
            if (undecided || resolved_on_device || probes_new || (min_margin < 1.5 && c->screen_since_refresh >= 4)) {
                ONMF_TRY(launch_screen_refresh(c, base, r, b, buf[done % 3], buf[(done + 2) % 3], s));
                c->screen_since_refresh = 0;
            }
        }
        if (posted && chain_clean) {     // the device put the result in H and the follow-up work stands
            if (after_valid) *after_valid = true;
            c->sweeps_total += done;
            if (sweeps_done) *sweeps_done = done;
            if (dist_trace)
                for (int i = done; i < sub_iter; ++i) dist_trace[i] = std::numeric_limits<float>::quiet_NaN();
            return ONMF_OK;
        }
    } else {
        double dist = 1.0;      // src/onmf.py:251
        // Exact coordinate descent runs to a fixed point, typically hundreds of sweeps: they go in
        // chunks of several sweeps per launch (8, 16, 32, 64, 64, ...) and the relative change is
        // tested on the last sweep of each chunk - one host round trip per chunk instead of per sweep.
        const bool chunked = exact_cd && sweep_tile_covers(r, b) && c->sweep_kernel != 1;
        int chunk = 8;
        while (done < sub_iter && keep_sweeping(dist, tol)) {
            const int count = chunked ? std::min(chunk, sub_iter - done) : 1;
            chunk = std::min(64, chunk * 2);
            float* cur = buf[(done + count - 1) % 3];
            float* nxt = buf[(done + count) % 3];
            int groups = 0;
            ONMF_TRY(profile_mark(c, s));
            if (within_radius && done == 0)
                ONMF_TRY(launch_sweep_radius_first(c, r, b, Gm, Cm, buf[0], buf[1], alpha, radius, (double*)c->stats.p,
                                                   &groups, s));
            else
                ONMF_TRY(launch_sweep(c, r, b, Gm, Cm, buf, alpha, exact_cd, done, count, (double*)c->stats.p, &groups, s));
            ONMF_TRY(profile_mark(c, s));
            HIP_TRY(c, hipMemcpyAsync(c->pinned, c->stats.p, (size_t)groups * 2 * sizeof(double),
                                      hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            const double* hs = (const double*)c->pinned;
            double dsum = 0.0, hsum = 0.0;
            for (int g = 0; g < groups; ++g) {
                dsum += hs[2 * g];
                hsum += hs[2 * g + 1];
            }
            if (dist_trace)
                for (int i = done; i + 1 < done + count; ++i) dist_trace[i] = (float)nan;   // not evaluated
            done += count - 1;      // (the statistics are those of the chunk's last sweep)
            // Decide `dist > tol` from Frobenius norms when they settle it:
            //   s1 in [fD / sqrt(rank), fD],  s0 in [fH / sqrt(rank), fH]   (fD, fH Frobenius norms)
            bool decided = false;
            double shown = nan;
            const double fD = std::sqrt(dsum), fH = std::sqrt(hsum);
            if (exact_cd) {
                // fixed-point iteration of the convex problem: stop on the relative Frobenius change
                dist = (hsum == 0.0) ? (dsum == 0.0 ? 0.0 : std::numeric_limits<double>::infinity())
                                     : fD / fH;
                decided = true;
                shown = dist;
            } else if (!c->exact_stop) {
                if (hsum == 0.0) {                // s0 == 0: dist is NaN (stop) or +inf (continue)
                    dist = (dsum == 0.0) ? nan : std::numeric_limits<double>::infinity();
                    decided = true;
                    shown = dist;
                } else if (dsum == 0.0) {
                    dist = 0.0;
                    decided = true;
                    shown = 0.0;
                } else if (tol < 0.0) {
                    dist = 1.0;                   // any positive value compares the same way
                    decided = true;
                } else {
                    const double lower = (fD / std::sqrt(rank_cap)) / fH * (1.0 - 1e-4);
                    const double upper = fD / (fH / std::sqrt(rank_cap)) * (1.0 + 1e-4);
                    if (lower > tol) { dist = lower; decided = true; shown = lower; }
                    else if (upper <= tol) { dist = upper; decided = true; shown = upper; }
                }
            }
            if (!decided) {
                ONMF_TRY(exact_dist(c, r, b, nxt, cur, s, &dist));
                shown = dist;
            }
            if (dist_trace) dist_trace[done] = (float)shown;
            ++done;
        }
    }
    if (buf[done % 3] != H)
        HIP_TRY(c, hipMemcpyAsync(H, buf[done % 3], rb * sizeof(float), hipMemcpyDeviceToDevice, s));
    c->sweeps_total += done;
    if (sweeps_done) *sweeps_done = done;
    if (dist_trace)
        for (int i = done; i < sub_iter; ++i) dist_trace[i] = std::numeric_limits<float>::quiet_NaN();
    return ONMF_OK;
}

// The verdict records of a solve that was not waited for (ONMF_OPT_DEFER_VERDICT).  Conclusive all the way:
// the bookkeeping the solve's own end would have done.  Not conclusive (bounds straddle the tolerance in the
// middle of the solve, or a workgroup gave up waiting): the solve is run again the ordinary way - its
// inputs are still there, the caller leaves a deferred call's buffers alone until the next call - and then
// whatever consumed its code.
int onmf_settle_pending(onmf_ctx* c) {
    if (!c->pend.active) return ONMF_OK;
    PendingSolve p = c->pend;
    c->pend.active = false;
    c->pend.redo_after = nullptr;
    HIP_TRY(c, hipEventSynchronize(c->ev_rec));
    const float* rec = c->rec_pinned;
    const double tol = (double)p.stopping_diff;
    double min_margin = 1e300;
    int done = 0;
    bool clean = true;
    for (int i = 0; i < p.sub_iter; ++i) {
        const int code = (int)rec[4 * i + 2];                 // 0 none, 1 continue, 2 stop, 3 inconclusive
        if (code == 0) { clean = false; break; }
        done = i + 1;
        if (code == 1) {
            if (i > 0) min_margin = std::min(min_margin, (double)rec[4 * i + 0] / tol);
            continue;
        }
        if (code == 2) break;
        if (i + 1 == p.sub_iter) continue;                    // the last sweep's test decides nothing
        clean = false;
        break;
    }
    float* base = (float*)c->screen.p;
    if (clean) {
        // (tests settled by the device rounds that were enqueued behind the sweeps read as conclusive here)
        const int* words = reinterpret_cast<const int*>(rec) + 4 * SCREEN_MAX_SWEEPS;
        const int nres = p.dev_rounds > 0 ? words[1] : 0, rounds = p.dev_rounds > 0 ? words[3] : 0;
        c->n_device_resolves += nres;
        c->screen_refresh_due = nres > 0 || min_margin < 1.5;
        c->screen_last_undecided = nres > 0;
        c->screen_last_evals = rounds;
        c->screen_last_stopped_early = done < p.sub_iter;
        if (p.cw_rows > 0 && done >= 1) {
            ++c->screen_since_refresh;
            if (nres > 0 || p.probes_new || (min_margin < 1.5 && c->screen_since_refresh >= 4)) {
                ONMF_TRY(launch_screen_refresh(c, base, p.r, p.b, p.buf[done % 3], p.buf[(done + 2) % 3], p.s));
                c->screen_since_refresh = 0;
            }
        }
        c->sweeps_total += done;
        return ONMF_OK;
    }
    int flag = 0;
    HIP_TRY(c, hipMemcpy(&flag, base + ScreenLayout::FLAG, sizeof(int), hipMemcpyDeviceToHost));
    const bool timed_out = flag == SCREEN_FLAG_TIMEOUT;
    if (timed_out) {
        ++c->n_sweep_timeouts;
        c->persistent_backoff = c->persistent_backoff ? std::min(c->persistent_backoff * 2, 1 << 20) : 64;
        c->persistent_hold = c->persistent_backoff;
    }
    c->screen_last_undecided = true;          // the second run does not post its follow-up work ahead
    c->screen_refresh_due = true;
    ++c->n_deferred_redone;
    int sweeps = 0;
    ONMF_TRY(sparse_code_impl(c, p.d, p.r, p.b, p.X, p.ldx, p.W, p.H0, p.seed, p.alpha, p.sub_iter, p.stopping_diff,
                              ONMF_CODER_DAMPED, p.H, &sweeps, nullptr, p.s, nullptr, nullptr, timed_out));
    return p.redo_after ? p.redo_after() : ONMF_OK;
}

static int aggregate_impl(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* H,
                          double w, const float* A_in, const float* B_in, const float* C_in,
                          float* A_out, float* B_out, float* C_out, hipStream_t s) {
    if (d <= 0 || r <= 0 || b <= 0 || !X || !H || !A_out || !B_out || ldx < b)
        return onmf_fail(c, ONMF_ERR_ARG, "aggregate: bad arguments");
    const float fw = (float)w, keep = (float)(1.0 - w);
    // A_out = (1-w) A_in + w H H^T ; B_out = (1-w) B_in + w H X^T   (src/onmf.py:155-156)
    ONMF_TRY(launch_gemm_nt_pair(c, r, b, H, b, H, b, r, X, ldx, d, fw, A_in ? keep : 0.f, A_in, A_out, r,
                                 B_in ? keep : 0.f, B_in, B_out, d, s));
    if (C_out)   // C_out = (1-w) C_in + w X X^T   (src/onmf.py:158)
        ONMF_TRY(launch_gemm(c, false, false, d, d, b, X, ldx, X, ldx, fw, C_in ? keep : 0.f, C_in, C_out, d, s));
    return ONMF_OK;
}

// After a dictionary pass whose workgroups wait for each other (d > 1024): read its timeout flag.
// If a workgroup gave up (the grid was not resident as a whole), the pass is done again without any
// waiting between workgroups (launch_dict_pass_serial), from W_in if that is still intact (W_out is
// another buffer) or from the transposed copy the panel kernel's head staged.  Synchronises s.
static int dict_check(onmf_ctx* c, int d, int r, const float* W_in, const float* A, const float* B, float* W_out,
                      hipStream_t s) {
    if (!c->dict_coop_last) return ONMF_OK;
    const size_t off = c->dict_flag_off;
    ONMF_TRY(onmf_reserve_pinned(c, 64));
    HIP_TRY(c, hipMemcpyAsync(c->pinned, (char*)c->sync.p + off, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (*(int*)c->pinned == 0) return ONMF_OK;
    ++c->n_dict_timeouts;
    c->dict_coop_last = false;
    const float* src = W_in;
    if (W_in == W_out) {
        // in place: the caller's W went with the discarded result; the head kernel's transposed copy
        // is intact (the panel kernel only reads it)
        const float* wt = c->dict_kernel != 1 ? dict_staged_wt(c, d, r) : nullptr;
        if (!wt)
            return onmf_fail(c, ONMF_ERR_TIMEOUT,
                             "update_dict: a workgroup gave up waiting for the others (not co-resident?) and the "
                             "in-place input is gone");
        ONMF_TRY(launch_transpose(r, d, wt, W_out, s));
        src = W_out;
    }
    c->err = "update_dict: a workgroup gave up waiting for the others (not co-resident?); the pass was redone "
             "atom by atom";
    return launch_dict_pass_serial(c, d, r, src, A, B, W_out, s);
}

extern "C" {

int onmf_sparse_code(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* W,
                     const float* H0, uint64_t seed, float alpha, int sub_iter, float stopping_diff,
                     int coder_mode, float* H, int* sweeps_done, float* dist_trace, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    ONMF_ENTER(c);
    ONMF_TRY(sparse_code_impl(c, d, r, b, X, ldx, W, H0, seed, alpha, sub_iter, stopping_diff,
                              coder_mode, H, sweeps_done, dist_trace, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return ONMF_OK;
}

int onmf_sparse_code_within_radius(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* W,
                                   const float* H0, uint64_t seed, float radius, float alpha, int sub_iter,
                                   float stopping_diff, float* H, int* sweeps_done, float* dist_trace, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (!(radius > 0.f) || !std::isfinite(radius))
        return onmf_fail(c, ONMF_ERR_ARG, "sparse_code_within_radius: the radius must be positive and finite");
    hipStream_t s = (hipStream_t)stream;
    ONMF_ENTER(c);
    ONMF_TRY(sparse_code_impl(c, d, r, b, X, ldx, W, H0, seed, alpha, sub_iter, stopping_diff, ONMF_CODER_DAMPED, H,
                              sweeps_done, dist_trace, s, nullptr, nullptr, false, nullptr, nullptr, radius));
    HIP_TRY(c, hipStreamSynchronize(s));
    return ONMF_OK;
}

int onmf_aggregate(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* H, double w,
                   const float* A_in, const float* B_in, const float* C_in, float* A_out, float* B_out,
                   float* C_out, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    ONMF_ENTER(c);
    return aggregate_impl(c, d, r, b, X, ldx, H, w, A_in, B_in, C_in, A_out, B_out, C_out,
                          (hipStream_t)stream);
}

int onmf_update_dict(onmf_ctx* c, int d, int r, const float* W_in, const float* A, const float* B,
                     float* W_out, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (d <= 0 || r <= 0 || !W_in || !A || !B || !W_out)
        return onmf_fail(c, ONMF_ERR_ARG, "update_dict: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ONMF_ENTER(c);
    ONMF_TRY(launch_dict_pass(c, d, r, W_in, A, B, W_out, s));
    return dict_check(c, d, r, W_in, A, B, W_out, s);
}

int onmf_update_dict_static(onmf_ctx* c, int d, int r, const float* W_in, const float* A, const float* B,
                            float* W_out, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (d <= 0 || r <= 0 || !W_in || !A || !B || !W_out)
        return onmf_fail(c, ONMF_ERR_ARG, "update_dict_static: bad arguments");
    if (d > STATIC_DICT_MAX_D || r > STATIC_DICT_MAX_R || dict_workspace_floats(d, r) > STATIC_SCRATCH_FLOATS)
        return onmf_fail(c, ONMF_ERR_ARG,
                         "update_dict_static: d=%d r=%d exceed the static scratch (d <= %d, r <= %d); use "
                         "onmf_update_dict", d, r, STATIC_DICT_MAX_D, STATIC_DICT_MAX_R);
    if (!c->static_scratch) return onmf_fail(c, ONMF_ERR_HIP, "update_dict_static: no static scratch on this device");
    ONMF_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &capture);
    const bool ordered = capture == hipStreamCaptureStatusNone && g_scratch_event[c->device];
    if (ordered && g_scratch_event_recorded[c->device]) HIP_TRY(c, hipStreamWaitEvent(s, g_scratch_event[c->device], 0));
    // one workgroup does the whole pass (d <= 1024): no waits between workgroups, no flag to read back,
    // nothing to clear - the head kernel rewrites every part of the workspace the pass reads
    const int rc = launch_dict_pass(c, d, r, W_in, A, B, W_out, s, 0, c->static_scratch);
    if (ordered) {
        HIP_TRY(c, hipEventRecord(g_scratch_event[c->device], s));
        g_scratch_event_recorded[c->device] = true;
    }
    return rc;
}

}  // extern "C"

// One step (src/onmf.py:119-167).  The dictionary pass reads (W_in, dA, dB) - the aggregates from
// BEFORE the step (onmf.py:161); the aggregate recursion starts from (gA, gB, gC) with weight w and
// writes (A_out, B_out, C_out).  onmf_step passes the same aggregates to both; onmf_step_partial
// passes none to the recursion (bare sums).  `gate` (optional): an event the dictionary pass and the
// aggregate products wait for - the code solve does not.
static int step_impl(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* W_in,
                     const float* dA, const float* dB, const float* gA, const float* gB, const float* gC,
                     const float* H0, uint64_t seed, double w, float alpha, int sub_iter, float stopping_diff,
                     float* H_out, float* A_out, float* B_out, float* C_out, float* W_out, int* sweeps_done,
                     float* dist_trace, hipStream_t s, hipEvent_t gate, bool defer = false) {
    float* H = H_out;
    if (!H) {
        // hbuf[] are the sweep ping-pong buffers, so the code gets its own scratch
        ONMF_TRY(onmf_reserve(c, c->dbuf, (size_t)r * b * sizeof(float) * 2, "step code"));
        H = (float*)c->dbuf.p + (size_t)r * b;
    }
    // what has to be done again, besides the solve, if the verdicts of a step that was not waited for turn
    // out inconclusive: the aggregate products (by value: it outlives this call)
    const std::function<int()> redo_products = [=]() {
        return aggregate_impl(c, d, r, b, X, ldx, H, w, gA, gB, gC, A_out, B_out, C_out, s);
    };
    const std::function<int()>* redo = defer ? &redo_products : nullptr;
    const bool aliased = (W_out == W_in) || (A_out == dA) || (B_out == dB);
    // (beside a code solve of a few thousand columns there is little to hide the pass behind, and two
    // latency chains on one chip stretch each other: measured serial vs beside, same box, d=441 r=25
    // b=1000 0.497 vs 0.589 ms per step, d=100 r=16 b=1000 0.281 vs 0.300; d=1024 r=64 b=8192 0.469 vs
    // 0.436, d=1200 r=100 b=8192 0.571 vs 0.507)
    const bool worth_a_stream = (double)r * b >= 2e5;
    if (c->overlap_dict && !aliased && worth_a_stream) {
        // The dictionary pass reads only W_in and the OLD aggregates (src/onmf.py:161), so it is
        // independent of this step's code and runs on a second stream (default priority) beside it.
        if (!c->side) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, EV_ORDER_ONLY));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, EV_ORDER_ONLY));
        }
        HIP_TRY(c, hipEventRecord(c->ev_fork, s));
        HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
        if (gate) HIP_TRY(c, hipStreamWaitEvent(c->side, gate, 0));
        // The pass is one workgroup per 1024 rows that claims its CU whole, so it cannot run beside the
        // sweeps (one workgroup per CU, all of them needed at once).  It runs in two parts: the first
        // atoms beside W^T X, the rest beside the aggregate products - posted right behind the sweeps,
        // ahead of the verdict kernels, so that its workgroup is resident before those products fill
        // the chip (arriving after them it waits for one of their workgroups to end: most of their
        // duration).
        // ... unless the sweeps leave CUs free (a batch of a few thousand columns: one small workgroup per 64
        // columns): then the whole pass runs beside the entire step, in one part.
        if (c->cu_count == 0) {
            int cus = 0;
            HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            c->cu_count = cus;
        }
        const int dict_wgs = (d + 1023) / 1024;
        const bool room_beside_sweeps = sweep_tile_covers(r, b) &&
                                        sweep_tile_workgroups(c, r, b, true) + dict_wgs + 8 <= c->cu_count;
        const bool split = dict_pass_splits(c, d, r) && !room_beside_sweeps;
        // (a pass in one part is off the critical path altogether: it is enqueued behind the solve's first
        // kernels - at these sizes the GPU waits for the host's enqueueing, and the pass's five launches in
        // front of them cost the step 40 us)
        bool whole_posted = split;
        const std::function<int()> post_whole = [&]() {
            if (whole_posted) return (int)ONMF_OK;
            whole_posted = true;
            return launch_dict_pass(c, d, r, W_in, dA, dB, W_out, c->side, 0);
        };
        int rc = split ? launch_dict_pass(c, d, r, W_in, dA, dB, W_out, c->side, 1) : ONMF_OK;
        int rc2 = ONMF_OK;
        bool agg_done = false, second_posted = !split;
        const std::function<int()> post_dict = [&]() {
            if (second_posted) return (int)ONMF_OK;
            second_posted = true;
            if (!c->ev_mid) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_mid, EV_ORDER_ONLY));
            HIP_TRY(c, hipEventRecord(c->ev_mid, s));                 // behind the sweeps
            HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_mid, 0));
            return launch_dict_pass(c, d, r, W_in, dA, dB, W_out, c->side, 2);
        };
        const std::function<int()> agg = [&]() {
            ONMF_TRY(post_dict());
            if (gate) HIP_TRY(c, hipStreamWaitEvent(s, gate, 0));
            return aggregate_impl(c, d, r, b, X, ldx, H, w, gA, gB, gC, A_out, B_out, C_out, s);
        };
        if (rc == ONMF_OK)
            rc2 = sparse_code_impl(c, d, r, b, X, ldx, W_in, H0, seed, alpha, sub_iter, stopping_diff,
                                   ONMF_CODER_DAMPED, H, sweeps_done, dist_trace, s, &agg, &agg_done, false, &post_dict,
                                   &post_whole, -1.f, redo);
        if (rc == ONMF_OK && rc2 == ONMF_OK) rc2 = post_whole();       // (a solve that failed early did not get to it)
        if (rc == ONMF_OK && rc2 == ONMF_OK && !agg_done) rc2 = agg();
        hipError_t e1 = hipEventRecord(c->ev_join, c->side);
        // always rejoin, also on an error path, so the side stream never outlives the call's buffers
        hipError_t e2 = hipStreamWaitEvent(s, c->ev_join, 0);
        if (rc != ONMF_OK) return rc;
        if (rc2 != ONMF_OK) return rc2;
        HIP_TRY(c, e1);
        HIP_TRY(c, e2);
        // A long step whose dictionary pass is several cooperating workgroups ends with a wait: the next
        // step's kernels are then enqueued onto an idle chip, the pass's workgroups a few microseconds
        // ahead of the big product's, and find their places together; enqueued behind a running step,
        // all streams are released at once and the pass's workgroups queue behind the product's
        // (measured at d=4096 r=256 b=65536: 3.11 ms per step with the wait, 3.22 without; short steps
        // gain more from the host running ahead: d=1200 r=100 b=8192 0.47 -> 0.39 ms without it).
        if (d > 1024 && 2.0 * r * (double)b * d >= 8e9) HIP_TRY(c, hipStreamSynchronize(s));
        return dict_check(c, d, r, W_in, dA, dB, W_out, s);
    }
    if (!aliased && !c->screen_last_undecided) {
        // One stream.  The dictionary pass and the products are enqueued behind the solve before the host
        // waits for its verdicts (the products run again if the host has to finish the solve: nothing
        // they read is overwritten).  Not when the previous solve needed the host: this one probably does
        // too, and what was enqueued ahead then stands between the host's evaluations and their
        // results (measured at d=441 r=25 b=1000, 7 sweeps on average: 0.498 ms per step without, 0.547 with).
        bool agg_done = false, dict_posted = false;
        const std::function<int()> post_dict = [&]() {
            if (dict_posted) return (int)ONMF_OK;
            dict_posted = true;
            if (gate) HIP_TRY(c, hipStreamWaitEvent(s, gate, 0));
            return launch_dict_pass(c, d, r, W_in, dA, dB, W_out, s);
        };
        const std::function<int()> agg = [&]() {
            ONMF_TRY(post_dict());
            return aggregate_impl(c, d, r, b, X, ldx, H, w, gA, gB, gC, A_out, B_out, C_out, s);
        };
        ONMF_TRY(sparse_code_impl(c, d, r, b, X, ldx, W_in, H0, seed, alpha, sub_iter, stopping_diff,
                                  ONMF_CODER_DAMPED, H, sweeps_done, dist_trace, s, &agg, &agg_done, false, nullptr,
                                  nullptr, -1.f, redo));
        if (!agg_done) ONMF_TRY(agg());
        return dict_check(c, d, r, W_in, dA, dB, W_out, s);
    }
    // serial order keeps every in-place aliasing legal: code (reads W_in) -> dictionary pass with
    // the OLD aggregates (checked before they are overwritten) -> aggregates
    ONMF_TRY(sparse_code_impl(c, d, r, b, X, ldx, W_in, H0, seed, alpha, sub_iter, stopping_diff,
                              ONMF_CODER_DAMPED, H, sweeps_done, dist_trace, s));
    if (gate) HIP_TRY(c, hipStreamWaitEvent(s, gate, 0));
    ONMF_TRY(launch_dict_pass(c, d, r, W_in, dA, dB, W_out, s));
    ONMF_TRY(dict_check(c, d, r, W_in, dA, dB, W_out, s));
    return aggregate_impl(c, d, r, b, X, ldx, H, w, gA, gB, gC, A_out, B_out, C_out, s);
}

extern "C" {

int onmf_step(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* W_in,
              const float* A_in, const float* B_in, const float* C_in, const float* H0, uint64_t seed,
              double w, float alpha, int sub_iter, float stopping_diff, float* H_out, float* A_out,
              float* B_out, float* C_out, float* W_out, int* sweeps_done, float* dist_trace,
              void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (!W_in || !A_in || !B_in || !A_out || !B_out || !W_out)
        return onmf_fail(c, ONMF_ERR_ARG, "step: null state pointer");
    ONMF_ENTER(c);
    const bool defer = c->defer_verdict && !sweeps_done && !dist_trace;
    return step_impl(c, d, r, b, X, ldx, W_in, A_in, B_in, A_in, B_in, C_in, H0, seed, w, alpha, sub_iter,
                     stopping_diff, H_out, A_out, B_out, C_out, W_out, sweeps_done, dist_trace,
                     (hipStream_t)stream, nullptr, defer);
}

int onmf_settle(onmf_ctx* c) {
    if (!c) return ONMF_ERR_ARG;
    ONMF_ENTER(c);           // (settles)
    return ONMF_OK;
}

size_t onmf_partial_floats(int d, int r) { return (((size_t)r * r + 3) & ~(size_t)3) + (size_t)r * d; }

int onmf_step_partial(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* W_in,
                      const float* A_in, const float* B_in, const float* H0, uint64_t seed, float alpha,
                      int sub_iter, float stopping_diff, float* H_out, float* P_out, float* W_out,
                      int* sweeps_done, float* dist_trace, void* state_ready_event, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (!W_in || !A_in || !B_in || !P_out || !W_out)
        return onmf_fail(c, ONMF_ERR_ARG, "step_partial: null state pointer");
    if (d <= 0 || r <= 0) return onmf_fail(c, ONMF_ERR_ARG, "step_partial: bad shape");
    ONMF_ENTER(c);
    float* HHt = P_out;
    float* HXt = P_out + (((size_t)r * r + 3) & ~(size_t)3);
    return step_impl(c, d, r, b, X, ldx, W_in, A_in, B_in, nullptr, nullptr, nullptr, H0, seed, 1.0, alpha, sub_iter,
                     stopping_diff, H_out, HHt, HXt, nullptr, W_out, sweeps_done, dist_trace, (hipStream_t)stream,
                     (hipEvent_t)state_ready_event);
}

int onmf_step_apply(onmf_ctx* c, int d, int r, const float* P, double w, const float* A_in, const float* B_in,
                    float* A_out, float* B_out, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (d <= 0 || r <= 0 || !P || !A_in || !B_in || !A_out || !B_out)
        return onmf_fail(c, ONMF_ERR_ARG, "step_apply: bad arguments");
    ONMF_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    const float fw = (float)w, keep = (float)(1.0 - w);
    const float* HHt = P;
    const float* HXt = P + (((size_t)r * r + 3) & ~(size_t)3);
    if (launch_axpby(A_out, A_in, keep, HHt, fw, (size_t)r * r, s) != ONMF_OK ||
        launch_axpby(B_out, B_in, keep, HXt, fw, (size_t)r * d, s) != ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "step_apply: launch failed");
    return ONMF_OK;
}

int onmf_gather_columns(onmf_ctx* c, int d, int n, const float* X, int ldx, const int64_t* idx, int b,
                        float* Xb, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (d <= 0 || n <= 0 || b <= 0 || !X || !idx || !Xb || ldx < n)
        return onmf_fail(c, ONMF_ERR_ARG, "gather_columns: bad arguments");
    ONMF_ENTER(c);
    if (launch_gather_columns(d, n, X, ldx, (const long long*)idx, b, Xb, (hipStream_t)stream) != ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "gather_columns: launch failed");
    return ONMF_OK;
}

int onmf_code_accumulate(onmf_ctx* c, int r, int n, float* code, const int64_t* idx, int b,
                         const float* H, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (r <= 0 || n <= 0 || b <= 0 || !code || !H || (!idx && b != n))
        return onmf_fail(c, ONMF_ERR_ARG, "code_accumulate: bad arguments");
    ONMF_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    if (!idx) {
        if (launch_axpy(code, H, 1.0f, (size_t)r * n, s) != ONMF_OK)
            return onmf_fail(c, ONMF_ERR_HIP, "code_accumulate: launch failed");
        return ONMF_OK;
    }
    return launch_scatter_add_last(c, r, n, code, (const long long*)idx, b, H, s);
}

int onmf_unfold3(onmf_ctx* c, int n0, int n1, int n2, const float* T, int mode, int transpose,
                 float* out, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (n0 <= 0 || n1 <= 0 || n2 <= 0 || !T || !out || mode < 0 || mode > 2)
        return onmf_fail(c, ONMF_ERR_ARG, "unfold3: bad arguments");
    ONMF_ENTER(c);
    if (launch_unfold3(n0, n1, n2, T, mode, transpose, out, (hipStream_t)stream) != ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "unfold3: launch failed");
    return ONMF_OK;
}

int onmf_sparse_code_each(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* W,
                          const float* H0, uint64_t seed, float alpha, int sub_iter, float stopping_diff,
                          int coder_mode, float* H, int32_t* sweeps_each, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (d <= 0 || r <= 0 || b <= 0 || !X || !W || !H || ldx < b || sub_iter < 0)
        return onmf_fail(c, ONMF_ERR_ARG, "sparse_code_each: bad arguments (d=%d r=%d b=%d ldx=%d)", d, r, b, ldx);
    hipStream_t s = (hipStream_t)stream;
    ONMF_ENTER(c);
    const size_t rb = (size_t)r * b;
    ONMF_TRY(onmf_reserve(c, c->gram, (size_t)r * r * sizeof(float), "Gm"));
    ONMF_TRY(onmf_reserve(c, c->cm, rb * sizeof(float), "Cm"));
    float* Gm = (float*)c->gram.p;
    float* Cm = (float*)c->cm.p;
    // Gm = W^T W, Cm = W^T X (src/onmf.py:242-243), the start (src/onmf.py:245-248), then every sweep
    // of every column in one launch
    ONMF_TRY(launch_gemm(c, true, true, r, r, d, W, r, W, r, 1.f, 0.f, nullptr, Gm, r, s));
    ONMF_TRY(launch_gemm(c, true, true, r, b, d, W, r, X, ldx, 1.f, 0.f, nullptr, Cm, b, s));
    if (H0) {
        if (H0 != H) HIP_TRY(c, hipMemcpyAsync(H, H0, rb * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else if (launch_fill_uniform(H, rb, seed, s) != ONMF_OK) {
        return onmf_fail(c, ONMF_ERR_HIP, "sparse_code_each: fill launch failed");
    }
    if (sub_iter == 0) {
        if (sweeps_each) HIP_TRY(c, hipMemsetAsync(sweeps_each, 0, (size_t)b * sizeof(int32_t), s));
        return ONMF_OK;
    }
    EachStop es = {stopping_diff, sweeps_each};
    return launch_sweep_each(c, r, b, Gm, Cm, H, alpha, coder_mode == ONMF_CODER_EXACT_CD, sub_iter, es, s);
}

int onmf_sparse_code_lars(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* W, double alpha,
                          int max_iter, float* H, int32_t* info, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (d <= 0 || r <= 0 || b <= 0 || !X || !W || !H || ldx < b || max_iter < 0 || !(alpha >= 0.0) || r > 256)
        return onmf_fail(c, ONMF_ERR_ARG, "sparse_code_lars: bad arguments (d=%d r=%d b=%d ldx=%d alpha=%g)", d, r, b,
                         ldx, alpha);
    hipStream_t s = (hipStream_t)stream;
    ONMF_ENTER(c);
    ONMF_TRY(onmf_reserve(c, c->gram64, (size_t)r * r * sizeof(double), "Gm (double)"));
    ONMF_TRY(onmf_reserve(c, c->cm64, (size_t)r * b * sizeof(double), "Cm (double)"));
    double* Gd = (double*)c->gram64.p;
    double* Cd = (double*)c->cm64.p;
    // the Gram matrix and the correlations the library is handed (_dict_learning.py: gram = W^T W, cov = W^T X)
    ONMF_TRY(launch_prod_f64_pair(c, r, d, W, r, X, ldx, b, Cd, b, W, r, r, Gd, r, s));     // [W^T X | W^T W]
    return launch_lars_pos(c, r, b, Gd, Cd, b, alpha, (double)d, max_iter, H, b, info, s);
}

int onmf_ntf_step(onmf_ctx* c, int d, int r, int b, const float* X, int ldx, const float* W_in, const float* A_in,
                  const float* B_in, double w, double alpha, int max_iter, float* H_out, float* A_out, float* B_out,
                  float* W_out, int32_t* info, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (d <= 0 || r <= 0 || b <= 0 || !X || !W_in || !A_in || !B_in || !A_out || !B_out || !W_out || ldx < b ||
        max_iter < 0 || !(alpha >= 0.0) || r > 256)
        return onmf_fail(c, ONMF_ERR_ARG, "ntf_step: bad arguments (d=%d r=%d b=%d ldx=%d alpha=%g)", d, r, b, ldx, alpha);
    hipStream_t s = (hipStream_t)stream;
    ONMF_ENTER(c);
    float* H = H_out;
    if (!H) {
        ONMF_TRY(onmf_reserve(c, c->dbuf, (size_t)r * b * sizeof(float) * 2, "step code"));
        H = (float*)c->dbuf.p + (size_t)r * b;
    }
    ONMF_TRY(onmf_reserve(c, c->gram64, (size_t)r * r * sizeof(double), "Gm (double)"));
    ONMF_TRY(onmf_reserve(c, c->cm64, (size_t)r * b * sizeof(double), "Cm (double)"));
    double* Gd = (double*)c->gram64.p;
    double* Cd = (double*)c->cm64.p;
    // the dictionary pass reads the aggregates from before the step (src/ontf.py:151) and nothing of the code:
    // it runs on the side stream beside the coder when its output is not one of the step's inputs
    const bool beside = c->overlap_dict && W_out != W_in && A_out != A_in && B_out != B_in;
    if (beside) {
        if (!c->side) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, EV_ORDER_ONLY));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, EV_ORDER_ONLY));
        }
        HIP_TRY(c, hipEventRecord(c->ev_fork, s));
        HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    }
    int rc_dict = beside ? launch_dict_pass(c, d, r, W_in, A_in, B_in, W_out, c->side, 0) : ONMF_OK;
    int rc = launch_prod_f64_pair(c, r, d, W_in, r, X, ldx, b, Cd, b, W_in, r, r, Gd, r, s);   // [W^T X | W^T W]
    if (rc == ONMF_OK) rc = launch_lars_pos(c, r, b, Gd, Cd, b, alpha, (double)d, max_iter, H, b, info, s);
    if (rc == ONMF_OK && !beside) rc = launch_dict_pass(c, d, r, W_in, A_in, B_in, W_out, s, 0);
    if (rc == ONMF_OK) rc = aggregate_impl(c, d, r, b, X, ldx, H, w, A_in, B_in, nullptr, A_out, B_out, nullptr, s);
    if (beside) {       // always rejoin, also on an error path
        hipError_t e1 = hipEventRecord(c->ev_join, c->side);
        hipError_t e2 = hipStreamWaitEvent(s, c->ev_join, 0);
        if (rc == ONMF_OK && rc_dict == ONMF_OK) {
            HIP_TRY(c, e1);
            HIP_TRY(c, e2);
        }
    }
    ONMF_TRY(rc_dict);
    ONMF_TRY(rc);
    return dict_check(c, d, r, W_in, A_in, B_in, W_out, s);
}

int onmf_patch_product(onmf_ctx* c, int d, int r, int b, const float* W, const float* H, float* R, int ldr,
                       void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (d <= 0 || r <= 0 || b <= 0 || !W || !H || !R || ldr < b)
        return onmf_fail(c, ONMF_ERR_ARG, "patch_product: bad arguments");
    ONMF_ENTER(c);
    return launch_gemm(c, false, true, d, b, r, W, r, H, b, 1.f, 0.f, nullptr, R, ldr, (hipStream_t)stream);
}

static int patch_args(onmf_ctx* c, const char* what, int h, int w, int ch, int k, int grid_rows, int grid_cols,
                      int stride) {
    if (h <= 0 || w <= 0 || ch <= 0 || k <= 0 || k > h || k > w)
        return onmf_fail(c, ONMF_ERR_ARG, "%s: bad image / patch size (h=%d w=%d ch=%d k=%d)", what, h, w, ch, k);
    if ((double)k * k * ch > 65535.0) return onmf_fail(c, ONMF_ERR_ARG, "%s: patch too large", what);
    if (grid_rows > 0 || grid_cols > 0) {
        if (stride <= 0 || grid_rows <= 0 || grid_cols <= 0 || (long long)(grid_rows - 1) * stride + k > h ||
            (long long)(grid_cols - 1) * stride + k > w)
            return onmf_fail(c, ONMF_ERR_ARG, "%s: the %d x %d grid (stride %d) leaves the image", what, grid_rows,
                             grid_cols, stride);
    }
    return ONMF_OK;
}

int onmf_extract_patches(onmf_ctx* c, const float* img, int h, int w, int ch, int k, const int32_t* corners,
                         int b, int grid_rows, int grid_cols, int stride, float* X, int ldx, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (!img || !X || b <= 0 || ldx < b) return onmf_fail(c, ONMF_ERR_ARG, "extract_patches: bad arguments");
    if (corners) grid_rows = grid_cols = 0;
    ONMF_TRY(patch_args(c, "extract_patches", h, w, ch, k, grid_rows, grid_cols, stride));
    if (!corners && (long long)grid_rows * grid_cols != b)
        return onmf_fail(c, ONMF_ERR_ARG, "extract_patches: b=%d is not the grid's %d x %d", b, grid_rows, grid_cols);
    ONMF_ENTER(c);
    if (launch_patch_extract(img, h, w, ch, k, (const int*)corners, b, corners ? 1 : grid_cols, corners ? 1 : stride,
                             X, ldx, (hipStream_t)stream) != ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "extract_patches: launch failed");
    return ONMF_OK;
}

int onmf_fold_patches(onmf_ctx* c, const float* R, int ldr, int h, int w, int ch, int k, int grid_rows,
                      int grid_cols, int stride, float* out, float* count, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (!R || !out || grid_rows <= 0 || grid_cols <= 0 || (long long)grid_rows * grid_cols > ldr)
        return onmf_fail(c, ONMF_ERR_ARG, "fold_patches: bad arguments");
    ONMF_TRY(patch_args(c, "fold_patches", h, w, ch, k, grid_rows, grid_cols, stride));
    ONMF_ENTER(c);
    if (launch_patch_fold(R, ldr, h, w, ch, k, grid_rows, grid_cols, stride, out, count, (hipStream_t)stream) !=
        ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "fold_patches: launch failed");
    return ONMF_OK;
}

int onmf_surrogate_loss(onmf_ctx* c, int d, int r, const float* W, const float* A, const float* B,
                        const float* Cagg, double* loss, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (d <= 0 || r <= 0 || !W || !A || !B || !loss) return onmf_fail(c, ONMF_ERR_ARG, "surrogate_loss: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ONMF_ENTER(c);
    constexpr int BLOCKS = 64;
    ONMF_TRY(onmf_reserve(c, c->qt, (size_t)r * d * sizeof(float), "W A"));
    ONMF_TRY(onmf_reserve(c, c->stats, BLOCKS * sizeof(double), "loss partials"));
    ONMF_TRY(onmf_reserve_pinned(c, BLOCKS * sizeof(double)));
    float* Q = (float*)c->qt.p;
    ONMF_TRY(launch_gemm(c, false, true, d, r, r, W, r, A, r, 1.f, 0.f, nullptr, Q, r, s));
    if (launch_surrogate_partials(d, r, Q, W, B, Cagg, (double*)c->stats.p, BLOCKS, s) != ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "surrogate_loss: launch failed");
    HIP_TRY(c, hipMemcpyAsync(c->pinned, c->stats.p, BLOCKS * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    double total = 0.0;
    for (int i = 0; i < BLOCKS; ++i) total += ((const double*)c->pinned)[i];
    *loss = total;
    return ONMF_OK;
}

static int ndl_args(onmf_ctx* c, const char* what, int n_nodes, const int32_t* row_ptr, const int32_t* nbr, int k,
                    const int32_t* motif, int n_chains, const int32_t* emb) {
    if (n_nodes <= 0 || !row_ptr || !nbr || !motif || !emb || n_chains <= 0)
        return onmf_fail(c, ONMF_ERR_ARG, "%s: bad arguments", what);
    if (k < 1 || k > 64) return onmf_fail(c, ONMF_ERR_ARG, "%s: motif size k=%d is not in 1..64", what, k);
    return ONMF_OK;
}

int onmf_ndl_init_embeddings(onmf_ctx* c, int n_nodes, const int32_t* row_ptr, const int32_t* nbr, int k,
                             const int32_t* motif, int n_chains, const int32_t* pivots, uint64_t seed,
                             int32_t* emb, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    ONMF_TRY(ndl_args(c, "ndl_init_embeddings", n_nodes, row_ptr, nbr, k, motif, n_chains, emb));
    ONMF_ENTER(c);
    if (launch_ndl_chains(n_nodes, row_ptr, nbr, k, motif, 2, n_chains, 0, emb, pivots, seed, 0, nullptr, 0, nullptr,
                          (hipStream_t)stream) != ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "ndl_init_embeddings: launch failed");
    return ONMF_OK;
}

int onmf_ndl_sample_patches(onmf_ctx* c, int n_nodes, const int32_t* row_ptr, const int32_t* nbr, int k,
                            const int32_t* motif, int chain_kind, int n_chains, int steps, int32_t* emb,
                            uint64_t seed, uint64_t step_offset, float* X, int ldx, int32_t* emb_trace,
                            void* stream) {
    if (!c) return ONMF_ERR_ARG;
    ONMF_TRY(ndl_args(c, "ndl_sample_patches", n_nodes, row_ptr, nbr, k, motif, n_chains, emb));
    if (chain_kind != ONMF_NDL_GLAUBER && chain_kind != ONMF_NDL_PIVOT)
        return onmf_fail(c, ONMF_ERR_ARG, "ndl_sample_patches: unknown chain kind %d", chain_kind);
    if (steps <= 0 || !X || (long long)n_chains * steps > ldx)
        return onmf_fail(c, ONMF_ERR_ARG, "ndl_sample_patches: %d chains x %d steps do not fit ldx=%d", n_chains, steps,
                         ldx);
    ONMF_ENTER(c);
    if (launch_ndl_chains(n_nodes, row_ptr, nbr, k, motif, chain_kind, n_chains, steps, emb, nullptr, seed, step_offset,
                          X, ldx, emb_trace, (hipStream_t)stream) != ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "ndl_sample_patches: launch failed");
    return ONMF_OK;
}

int onmf_ndl_accumulate_edges(onmf_ctx* c, int n_nodes, int k, int n_patches, const int32_t* emb_trace, const float* R,
                              int ldr, float* sum, float* count, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (n_nodes <= 0 || k < 1 || k > 64 || n_patches <= 0 || !emb_trace || !R || !sum || !count || ldr < n_patches)
        return onmf_fail(c, ONMF_ERR_ARG, "ndl_accumulate_edges: bad arguments");
    ONMF_ENTER(c);
    if (launch_ndl_edge_accumulate(n_nodes, k, n_patches, emb_trace, R, ldr, sum, count, (hipStream_t)stream) != ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "ndl_accumulate_edges: launch failed");
    return ONMF_OK;
}

int onmf_numpy_set_jump_table(onmf_ctx* c, const uint32_t* table_host, int rows, long long chunk_words) {
    if (!c) return ONMF_ERR_ARG;
    if (rows < 0 || rows > 4096 || (rows > 0 && !table_host) || chunk_words != numpy_rand_chunk_words())
        return onmf_fail(c, ONMF_ERR_ARG, "numpy_set_jump_table: %d rows for chunks of %lld words (the kernels' "
                         "chunk is %lld words)", rows, chunk_words, numpy_rand_chunk_words());
    ONMF_ENTER(c);
    return numpy_rand_setup(c, table_host, rows, nullptr);
}

int onmf_numpy_rand(onmf_ctx* c, const uint32_t* key, int pos, size_t n, float* out, uint32_t* key_out, int* pos_out,
                    void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (!key || !key_out || !pos_out || pos < 0 || pos > 624 || (n > 0 && !out))
        return onmf_fail(c, ONMF_ERR_ARG, "numpy_rand: bad arguments (pos=%d)", pos);
    ONMF_ENTER(c);
    if (c->mt_rows < 0) ONMF_TRY(numpy_rand_setup(c, nullptr, 0, (hipStream_t)stream));
    return launch_numpy_rand(c, key, pos, n, out, key_out, pos_out, (hipStream_t)stream);
}

int onmf_fill_uniform(onmf_ctx* c, float* out, size_t n, uint64_t seed, void* stream) {
    if (!c) return ONMF_ERR_ARG;
    if (!out && n) return onmf_fail(c, ONMF_ERR_ARG, "fill_uniform: null output");
    ONMF_ENTER(c);
    if (launch_fill_uniform(out, n, seed, (hipStream_t)stream) != ONMF_OK)
        return onmf_fail(c, ONMF_ERR_HIP, "fill_uniform: launch failed");
    return ONMF_OK;
}

}  // extern "C"

#ifdef ONMF_TRACE_WG
// Diagnostic build only: workgroup 0's time marks of the last screened solve (16 per sweep, 100 MHz ticks).
extern "C" int onmf_debug_sweep_marks(onmf_ctx* c, unsigned long long* host, int max_marks) {
    if (!c || !c->screen.p) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const int n = std::min(max_marks, 16 * SCREEN_MAX_SWEEPS);
    const char* src = (const char*)c->screen.p + (((size_t)ScreenLayout::TOTAL * sizeof(float) + 63) & ~(size_t)63);
    if (hipMemcpy(host, src, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
#endif
