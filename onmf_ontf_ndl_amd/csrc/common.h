// Shared declarations for the gfx950 ONMF library (internal; the public C-ABI is include/onmf_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../include/onmf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// context: owns scratch only (slabs for split-K partials, the W^T X panel, per-sweep statistics)
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// A step whose stopping-test verdicts the host has not looked at yet (ONMF_OPT_DEFER_VERDICT): everything
// was enqueued on the assumption that the device's verdicts are conclusive; the next entry point on the
// context reads the records and, where they are not, runs the solve and what consumed its code again.
struct PendingSolve {
    bool active = false;
    int d = 0, r = 0, b = 0, ldx = 0, sub_iter = 0, cw_rows = 0;
    const float *X = nullptr, *W = nullptr, *H0 = nullptr;
    uint64_t seed = 0;
    float alpha = 0.f, stopping_diff = 0.f;
    float* H = nullptr;
    float* buf[3] = {nullptr, nullptr, nullptr};
    bool probes_new = false;
    int dev_rounds = 0;                     // rounds of device-side evaluation enqueued behind the sweeps
    hipStream_t s = nullptr;
    std::function<int()> redo_after;        // what consumed the code (the step's aggregate products), by value
};

struct onmf_ctx {
    int device = 0;
    std::string err;
    DevBuf slab;        // split-K partial products
    DevBuf cm;          // Cm = W^T X  (r x b)
    DevBuf gram;        // Gm = W^T W  (r x r)
    DevBuf hbuf[2];     // ping-pong code buffers (r x b)
    DevBuf stats;       // per-workgroup sweep statistics (double pairs)
    DevBuf small;       // r x r scratch x 2 (stop-test Grams) + dictionary-pass scratch
    DevBuf qt;          // dictionary pass: Qt = A^T W^T  (r x d)
    DevBuf sync;        // dictionary pass: cross-workgroup slots + counters
    DevBuf idx;         // scatter scratch
    std::vector<double> hstats;
    std::vector<float> hgram;
    void* pinned = nullptr;   // small pinned host staging
    size_t pinned_bytes = 0;
    // instrumentation: device time of the last dominant-kernel launches (filled on request)
    int profile = 0;
    int exact_stop = 0;
    int sweep_kernel = 0;
    std::vector<hipEvent_t> ev;     // pairs of events bracketing sweep launches
    int ev_used = 0;
    DevBuf dbuf;                    // H - H_prev for the exact stopping test (r x b)
    DevBuf gstack;                  // small r: [H - H_prev ; H_prev] stacked (2r x b), one product for both Grams
    DevBuf rslab;                   // device-side verdict resolution: per-workgroup partial Gram matrices
    int device_resolve = -1;        // settle inconclusive verdicts on the device: < 0 where it pays (see api.hip), 0 never, > 0 for every r <= 64
    long long n_device_resolves = 0;
    DevBuf screen;                  // stopping-test screen state (ScreenLayout)
    DevBuf slab_side;               // split-K slabs of products issued on the side stream
    hipStream_t side = nullptr;     // dictionary pass runs here, beside the code solve
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_rec = nullptr, ev_mid = nullptr;
    hipStream_t prep = nullptr;     // the solve's small preparations run here, beside the W^T X product
    // host memory the device can write: [0] dictionary passes redone on the device after a timeout
    int* host_events = nullptr;
    int* host_events_dev = nullptr;
    hipEvent_t ev_prep_fork = nullptr, ev_prep_join = nullptr;
    DevBuf slab_prep;               // split-K slabs of products issued on the preparation stream
    int overlap_dict = 1;
    int lag3 = 1;               // control-wave kernels may act on a test three sweeps later (ONMF_OPT_VERDICT_LAG3)
    long long n_lag3_redone = 0;    // solves run again because such a test came out inconclusive
    int gemm_tail = 0;          // finer grain at the end of the big W^T X launch: 100 * percent of tiles + pieces (0: off)
    int screen_stop = 1;
    int screen_r = -1;              // r of the probes currently held in `screen`
    bool screen_refresh_due = true; // the next screened solve runs a probe power step
    int screen_since_refresh = 0;   // solves since the probes' last power step (ranks that refresh behind the solve)
    bool screen_last_undecided = false;   // the last screened solve needed the host for a verdict
    int screen_last_evals = 0;            // ... and this many of its tests were evaluated in full
    bool screen_last_stopped_early = false;   // the last screened solve ended before its last sweep
    long long spin_limit = 200000000LL;   // bound of the sweeps' waits for each other, 100 MHz ticks (2 s)
    int persistent_sweeps = 1;      // all sweeps of a screened solve in one launch when the grid fits the chip
    int cu_count = 0;
    DevBuf ddelta;                  // dictionary pass: atom deltas (d x r), one-workgroup kernel
    int dict_kernel = 0;            // 0 = by shape, 1 = cooperative workgroups, 2 = one workgroup
    size_t dict_flag_off = 0;       // byte offset of that flag in `sync`
    bool dict_coop_last = false;    // the last dictionary pass used the cooperative kernel (its timeout flag is live)
    // one-launch solves after a timeout: how many solves run with one launch per sweep before the
    // one-launch mode is tried again (doubles at every timeout), and how many of them are left
    int persistent_backoff = 0;
    int persistent_hold = 0;
    long long n_sweep_timeouts = 0;       // solves whose workgroups gave up waiting for each other
    long long n_dict_timeouts = 0;        // dictionary passes whose workgroups gave up waiting for each other
    long long n_fallback_solves = 0;      // solves run with one launch per sweep because of an earlier timeout
    float* static_scratch = nullptr;      // this device's copy of the library's static scratch array
    // positive-LARS coder: W^T W and W^T X accumulated in double, per-wave inverses that do not fit in LDS
    DevBuf gram64, cm64, lars_ws, lars_meta, lars_overflow;
    bool lars_hint_ok = false;            // a LARS call of this context has published its slot count
    int lars_slots = 0;                   // ONMF_OPT_LARS_SLOTS
    // verdicts looked at by the next call instead of this one
    int defer_verdict = 0;                // ONMF_OPT_DEFER_VERDICT
    PendingSolve pend;
    float* rec_pinned = nullptr;          // the verdict records in host memory the device writes (screen_final_kernel)
    float* rec_pinned_dev = nullptr;      // the same memory as the device addresses it
    long long sweeps_total = 0;           // sweeps of all settled solves
    long long n_deferred_redone = 0;      // deferred solves that had to be run again
    // NumPy's MT19937 stream on the device (mt19937.hip): jump table, head words, chunk blocks, final block
    DevBuf mt;
    int mt_rows = -1;                     // rows of the jump table held (-1: buffers not set up)
};

// Makes the context's device current for the duration of an entry point and puts the caller's back
// on the way out (the caller may be driving another GPU from the same thread).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else if (err != hipSuccess) prev = -1;
        if (prev == device) prev = -1;            // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
int onmf_settle_pending(onmf_ctx* c);
#define ONMF_ENTER(ctx)                                                                          \
    DeviceGuard _device_guard((ctx)->device);                                                    \
    if (_device_guard.err != hipSuccess)                                                         \
        return onmf_fail((ctx), ONMF_ERR_HIP, "selecting device %d failed: %s", (ctx)->device,   \
                         hipGetErrorString(_device_guard.err));                                  \
    if ((ctx)->pend.active) {                                                                    \
        const int _settle_rc = onmf_settle_pending(ctx);                                         \
        if (_settle_rc != ONMF_OK) return _settle_rc;                                            \
    }

int onmf_fail(onmf_ctx* c, int code, const char* fmt, ...);
int onmf_reserve(onmf_ctx* c, DevBuf& b, size_t bytes, const char* what);
int onmf_reserve_pinned(onmf_ctx* c, size_t bytes);

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return onmf_fail((ctx), ONMF_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                             hipGetErrorString(_e), __FILE__, __LINE__);                        \
    } while (0)

#define ONMF_TRY(expr)                                                                          \
    do {                                                                                        \
        int _rc = (expr);                                                                       \
        if (_rc != ONMF_OK) return _rc;                                                         \
    } while (0)

// A kernel that wants more than 64 KB of dynamic LDS must be told so once per device: the attribute
// belongs to the current device's copy of the kernel, so the "already done" note is kept per device
// (one static LdsLimit per kernel).  Relaxed atomics: two contexts on two threads may both set the
// attribute, which is harmless.
struct LdsLimit {
    static constexpr int MAX_DEV = 64;
    std::atomic<size_t> bytes[MAX_DEV];
    LdsLimit() { for (auto& b : bytes) b.store(0, std::memory_order_relaxed); }
};
inline int onmf_raise_lds_limit(onmf_ctx* c, LdsLimit& lim, const void* kernel, size_t bytes) {
    const bool slot = c->device >= 0 && c->device < LdsLimit::MAX_DEV;
    if (slot && lim.bytes[c->device].load(std::memory_order_relaxed) >= bytes) return ONMF_OK;
    HIP_TRY(c, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (slot) lim.bytes[c->device].store(bytes, std::memory_order_relaxed);
    return ONMF_OK;
}

// ---------------------------------------------------------------------------------------------
// kernels' host launchers (defined in the .hip files)
// ---------------------------------------------------------------------------------------------
// C = beta*Cin + alpha * op(A) op(B).  a_km: A(m,k)=A[k*lda+m] else A[m*lda+k];
// b_kn: B(k,n)=B[k*ldb+n] else B[n*ldb+k].  Cin may be null when beta == 0; Cin may alias C.
int launch_gemm(onmf_ctx* c, bool a_km, bool b_kn, int M, int N, int K, const float* A, int lda,
                const float* B, int ldb, float alpha, float beta, const float* Cin, float* C,
                int ldc, hipStream_t s, DevBuf* slab = nullptr, int tail_spec = 0);

// Two products that share the left operand, A B1^T and A B2^T (all operands contraction-fast), in one
// launch where the shapes allow; see gemm.hip.
// (launch_gemm_pair: any operand layout - both B operands in the same one - and the slab set to use)
int launch_gemm_pair(onmf_ctx* c, bool a_km, bool b_kn, int M, int K, const float* A, int lda, const float* B1,
                     int ldb1, int N1, const float* B2, int ldb2, int N2, float alpha, float beta1, const float* Cin1,
                     float* C1, int ldc1, float beta2, const float* Cin2, float* C2, int ldc2, hipStream_t s,
                     DevBuf* slab);
int launch_gemm_nt_pair(onmf_ctx* c, int M, int K, const float* A, int lda, const float* B1, int ldb1, int N1,
                        const float* B2, int ldb2, int N2, float alpha, float beta1, const float* Cin1, float* C1,
                        int ldc1, float beta2, const float* Cin2, float* C2, int ldc2, hipStream_t s);

// What a probing sweep launch needs of the stopping-test screen (screen.hip).
struct ScreenDev {
    float* base;         // ScreenLayout
    int v_first;         // first sweep whose test is still open (earlier ones were settled before the launch)
    float tol;           // stopping_diff
    float rank_cap;      // min(r, b)
    long long spin_limit;    // bound (100 MHz ticks) on waiting for the other workgroups' sums
    const int* start_dev;    // not null: the launch's first sweep is read from here (and is its v_first), ...
    int k_end;               // ... its last sweep is k_end - 1
    int lag;                 // kernels with a control wave: 3 = act on a test three sweeps later (see sweep_tile.h); else two
};
// whether the sweeps of a screened solve at this shape come from the kernel with a control wave
bool sweep_has_control_wave_at(onmf_ctx* c, int r, int b);

// Per-column stopping (many independent one-column solves in one launch).
struct EachStop {
    float tol;           // a column goes on while ||h_new - h_old|| / ||h_old|| > tol
    int* sweeps;         // device int[b] or null: sweeps each column ran
};

// Gauss-Seidel sweeps k_first .. k_first + k_count - 1 over all rows: sweep k reads hb[k % 3] (only
// the first of a launch actually loads it) and writes hb[(k + 1) % 3].  stats (device, 2 doubles per
// wave) receives sum((new - old)^2) and sum(old^2) partials of the LAST sweep run; *n_groups = number
// of pairs.  With `screen` the sweeps also leave the stopping test's by-products and end themselves
// when it says so (k_count > 1 needs `screen` and a grid that is resident as a whole).
int launch_sweep_tile_ct1(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                          float alpha, int exact_cd, int k_first, int k_count, double* stats,
                          int* n_groups, hipStream_t s, const ScreenDev* screen);
int launch_sweep_tile_ct2(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                          float alpha, int exact_cd, int k_first, int k_count, double* stats,
                          int* n_groups, hipStream_t s, const ScreenDev* screen);
int launch_sweep(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                 float alpha, int exact_cd, int k_first, int k_count, double* stats, int* n_groups,
                 hipStream_t s, const ScreenDev* screen = nullptr);
// Sweep 0 of a solve within a radius (src/onmf.py:260-263), by the column kernel.
int launch_sweep_radius_first(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, const float* Hin,
                              float* Hout, float alpha, float radius, double* stats, int* n_groups,
                              hipStream_t s);
// Sweeps 0 .. sub_iter-1 in one launch, every column with its own stopping test; H in place.
int launch_sweep_each(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* H, float alpha,
                      int exact_cd, int sub_iter, EachStop es, hipStream_t s);
int launch_sweep_tile_each(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* H, float alpha,
                           int exact_cd, int sub_iter, EachStop es, hipStream_t s);
int launch_sweep_tile_each4(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* H, float alpha,
                            int exact_cd, int sub_iter, EachStop es, hipStream_t s);
int launch_sweep_tile_ct2_small(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                                float alpha, int exact_cd, int k_first, int k_count, double* stats,
                                int* n_groups, hipStream_t s, const ScreenDev* screen);
int launch_sweep_tile_ct2_medium(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                                float alpha, int exact_cd, int k_first, int k_count, double* stats,
                                int* n_groups, hipStream_t s, const ScreenDev* screen);
bool sweep_tile_small_covers(int r);
// workgroups of a sweep launch (probing or not) for this shape: 256 columns each, or 64 each for the
// small-workgroup probing kernel (small r, few columns)
int sweep_tile_workgroups(onmf_ctx* c, int r, int b, bool probing);
bool sweep_tile_covers(int r, int b);

// Image <-> patch matrix (patches.hip).  corners: device int[2*b] (row, column) pairs, or null for the
// regular grid of grid_cols patches per row, `stride` pixels apart.
int launch_patch_extract(const float* img, int h, int w, int ch, int k, const int* corners, int b, int grid_cols,
                         int stride, float* X, int ldx, hipStream_t s);
int launch_patch_fold(const float* R, int ldr, int h, int w, int ch, int k, int grid_rows, int grid_cols, int stride,
                      float* out, float* count, hipStream_t s);

// Motif-sampling chains (ndl.hip): mode 0 Glauber, 1 pivot, 2 initial embedding from pivots.
int launch_ndl_chains(int n, const int* row_ptr, const int* nbr, int k, const int* motif, int mode, int n_chains,
                      int steps, int* emb, const int* pivots, uint64_t seed, uint64_t step0, float* X, int ldx,
                      int* emb_trace, hipStream_t s);
int launch_ndl_edge_accumulate(int n, int k, int T, const int* emb_trace, const float* R, int ldr, float* sum,
                               float* count, hipStream_t s);

// Philox4x32-10 counter-based generator (one round; ten of them with the keys stepped give 4 x 32 random bits
// per counter): the device's own uniform draws.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// U[0,1) floats 4q .. 4q+3 of the stream keyed by `seed` into out (n floats in all)
__device__ __forceinline__ void fill_uniform_quad(float* __restrict__ out, size_t n, size_t q, uint64_t seed) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int rd = 0; rd < 10; ++rd) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)(c[e] >> 8) * (1.0f / 16777216.0f);   // [0,1)
    const size_t base = q * 4;
    if (base + 3 < n && (((uintptr_t)(out + base)) & 15) == 0) {
        *reinterpret_cast<float4*>(out + base) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (base + e < n) out[base + e] = v[e];
    }
}

// Stopping-test screen: state layout and kernels.
// first of the eight words of onmf_ctx::host_events that the LARS coder's last kernel writes: most slots a column
// used, r, columns recoded so far, columns that needed more than 16 / 32 / 48 / 64 slots, columns coded
constexpr int ONMF_LARS_HOST_WORDS = 4;
constexpr int SCREEN_MAX_SWEEPS = 64;
constexpr int SCREEN_REPL = 8;       // replicas of the sweep kernel's accumulators (atomic contention)
constexpr int SCREEN_ZREPL = 4;      // replicas of the power-step accumulator
constexpr int SCREEN_YREPL = 2;      // replicas of the sweep kernel's Collatz-Wielandt accumulator
constexpr int SCREEN_CW_ROWS = 128;  // largest (padded) r whose sweeps carry that accumulator
// rows of that accumulator for rank r (0: the sweeps of this rank do not carry it)
inline int screen_cw_rows(int r) { const int rp = (r + 15) / 16 * 16; return rp <= SCREEN_CW_ROWS ? rp : 0; }
struct ScreenLayout {           // offsets in floats into ctx->screen
    static constexpr int UBLK = 0;                                 // 16 * 256: probe blocks
    static constexpr int CWV = 4096;                               // 256: positive vector of the Collatz-Wielandt bound
    static constexpr int ZACC = CWV + 256;                         // ZREPL x 256 x 16
    // one accumulator set per sweep of the solve (never reused: no clearing races)
    static constexpr int SETS = ZACC + SCREEN_ZREPL * 4096;
    static constexpr int SET_P2 = 0;                               // REPL x 16 probe energies
    static constexpr int SET_SUMS = SCREEN_REPL * 16;              // REPL x 2 doubles
    // YREPL x 128 doubles: y = H (H^T v) of the code the sweep leaves (r <= 128: Collatz-Wielandt bound
    // of its sigma_max as a by-product of the sweep, see sweep_tile.h)
    static constexpr int SET_Y = SET_SUMS + SCREEN_REPL * 4;
    static constexpr int SETSZ = SET_Y + SCREEN_YREPL * 2 * SCREEN_CW_ROWS;
    static constexpr int CNT = SETS + SCREEN_MAX_SWEEPS * SETSZ;   // ints: workgroups done with sweep k
    static constexpr int UBH = CNT + SCREEN_MAX_SWEEPS;            // MAX + 1: bound of sigma_max(H before sweep k)
    static constexpr int FLAG = UBH + SCREEN_MAX_SWEEPS + 1;       // int: 0 = keep sweeping
    static constexpr int AVEC = FLAG + 1;          // 16: last Ritz coefficient vector (warm start)
    static constexpr int REC = AVEC + 16;          // SCREEN_MAX_SWEEPS x 4 floats
    // device-side resolution of an inconclusive verdict (screen_resolve_kernel): ints
    static constexpr int NEXT = REC + 4 * SCREEN_MAX_SWEEPS;       // first sweep of the launch enqueued behind it
    static constexpr int NRES = NEXT + 1;                          // verdicts resolved on the device in this solve
    static constexpr int RCNT = NEXT + 2;                          // its workgroups done with their partial Grams
    static constexpr int RNDS = NEXT + 3;                          // launches of it that settled something in this solve
    static constexpr int CWY = NEXT + 4;                           // 256 doubles: H H^T v
    static constexpr int TOTAL = CWY + 512;
};
static_assert(ScreenLayout::SETS % 2 == 0 && ScreenLayout::SETSZ % 2 == 0 && ScreenLayout::SET_SUMS % 2 == 0 &&
                  ScreenLayout::CWY % 2 == 0,
              "double alignment");
constexpr int SCREEN_FLAG_TIMEOUT = 4;     // a workgroup gave up waiting for the others
// spin limit that makes every bounded wait between workgroups give up at its first look, whether or
// not what it waits for is there already (ONMF_OPT_SPIN_LIMIT_MS = -2): exercises the recovery paths
constexpr long long SPIN_FORCE_TIMEOUT = -2LL;

#if defined(__HIPCC__)
// The three-way verdict of one sweep from its by-products (see screen.hip).  Deterministic: every
// caller that feeds it the same numbers gets the same answer.  0 continue, 1 stop, 2 inconclusive.
__device__ __forceinline__ int screen_verdict(float lb2, double dsum, double hsum, float ubp, float tol,
                                              float rank_cap, float& lower_f, float& upper_f, float& ubh_f,
                                              float& ubnext_f) {
    const double fD = sqrt(dsum), fH = sqrt(hsum);
    double ubH = fH * (1.0 + 1e-5);                       // sigma_max <= Frobenius norm
    if (ubp > 0.f && (double)ubp < ubH) ubH = ubp;
    int verdict;
    double lower = 0.0, upper = 1e300;
    if (hsum == 0.0) {            // dist = 0/0 (NaN: loop ends) or x/0 (+inf: continues)
        verdict = (dsum == 0.0) ? 1 : 0;
        lower = upper = (dsum == 0.0) ? 0.0 : 1e300;
    } else if (dsum == 0.0) {     // dist = 0
        verdict = (0.0 > (double)tol) ? 0 : 1;
        lower = upper = 0.0;
    } else {
        const double lbs = fmax(sqrt((double)lb2) * (1.0 - 2e-4), fD / sqrt((double)rank_cap) * (1.0 - 1e-4));
        lower = lbs / ubH;
        upper = fD * (1.0 + 1e-4) / (fH / sqrt((double)rank_cap));
        verdict = (lower > (double)tol) ? 0 : ((upper <= (double)tol) ? 1 : 2);
    }
    lower_f = (float)lower;
    upper_f = (float)fmin(upper, 3.0e38);
    ubh_f = (float)ubH;
    // sigma_max(H_new) <= sigma_max(H_old) + sigma_max(H_new - H_old) <= ubH + ||H_new - H_old||_F
    ubnext_f = (float)fmin((ubH + fD) * (1.0 + 1e-5), 3.0e38);
    return verdict;
}

// Everything a workgroup reads of what ANOTHER workgroup of the same launch wrote goes through agent-scope
// atomics (relaxed): the accumulator sets, the completion counts, the flag, and the bounds below.  The
// kernels rely on no ordering beyond "an atomic whose value has come back has been performed": a workgroup
// waits for the values its own atomic adds return before it bumps the count, so whoever sees the count
// complete finds all sums performed.  Plain loads are used only for data that no workgroup of the launch
// writes (the probe blocks, the positive vector).
__device__ __forceinline__ float screen_load_bound(const float* base, int k) {
    return __hip_atomic_load(base + ScreenLayout::UBH + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same for sweep j, from its accumulator set; called by all 64 lanes of one wave.  The set is
// read with agent-scope atomic loads: within a launch the sums come from workgroups on other XCDs,
// whose stores a plain load could miss in this XCD's L2.  `ub` carries the bound of
// sigma_max(H before sweep j) in and the bound for sweep j + 1 out.  With `record` the caller is the
// one workgroup that writes the verdict down.  Returns the verdict, wave-uniform.
// `seeded` >= 0: the caller has already read base[UBH + j + 1] (it may do so while it still waits for
// the sweep's count: one memory round trip less on the way to the verdict); < 0: read it here.
// `cw_rows` > 0: the sweep also left y = H_new (H_new^T v) in its set (rows 0 .. cw_rows - 1): with the
// positive vector v (CWV), max_k y_k / v_k bounds sigma_max(H_new)^2 from above (Collatz-Wielandt; H_new
// is entrywise nonnegative) - H_new is what sweep j + 1 starts from, so the bound replaces the inherited
// one (sigma_max(H_old) + ||H_new - H_old||_F, which gives away a few per cent per sweep) when tighter.
__device__ __forceinline__ int screen_eval_wave(float* base, int j, float& ub, float tol, float rank_cap,
                                                bool record, float seeded_in = -1.0f, int cw_rows = 0) {
    const int lane = threadIdx.x & 63;
    float* set = base + ScreenLayout::SETS + j * ScreenLayout::SETSZ;
    float cw_ratio = 0.f;
    for (int row = lane; row < cw_rows; row += 64) {
        double y = 0.0;
        for (int rp = 0; rp < SCREEN_YREPL; ++rp)
            y += __hip_atomic_load(reinterpret_cast<double*>(set + ScreenLayout::SET_Y) + rp * SCREEN_CW_ROWS + row,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float v = base[ScreenLayout::CWV + row];
        cw_ratio = fmaxf(cw_ratio, (float)(y / (double)v) * (1.0f + 1e-6f));
    }
    // all loads of the wave go out together (one memory round trip): lanes 0-15 the probe energies,
    // lanes 16-31 the Frobenius sums
    float e = 0.f;
    double sv = 0.0;
    if (lane < 16) {
        for (int rp = 0; rp < SCREEN_REPL; ++rp)
            e += __hip_atomic_load(set + ScreenLayout::SET_P2 + rp * 16 + lane, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    } else if (lane < 16 + 2 * SCREEN_REPL) {
        // lane 16 + 2 rp + q holds sums[2 rp + q] (q = 0: sum of squared moves, 1: sum of squares of H)
        sv = __hip_atomic_load(reinterpret_cast<double*>(set + ScreenLayout::SET_SUMS) + (lane - 16),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) e = fmaxf(e, __shfl_xor(e, off));
    double dsum = 0.0, hsum = 0.0;
    for (int rp = 0; rp < SCREEN_REPL; ++rp) {                // replica order as the sums were laid out
        dsum += __shfl(sv, 16 + 2 * rp, 64);
        hsum += __shfl(sv, 17 + 2 * rp, 64);
    }
    if (cw_rows > 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cw_ratio = fmaxf(cw_ratio, __shfl_xor(cw_ratio, off));
    }
    int verdict = 0;
    float ubn = ub;
    if (lane == 0) {
        float lower, upper, ubh;
        verdict = screen_verdict(e, dsum, hsum, ub, tol, rank_cap, lower, upper, ubh, ubn);
        // a bound for the next sweep that was put there before the launch (Collatz-Wielandt, or the
        // host's exact value) stands if it is the tighter one; every workgroup reads the same value
        // or the minimum already taken with it, so all arrive at the same number
        const float seeded = seeded_in >= 0.f ? seeded_in : screen_load_bound(base, j + 1);
        if (seeded > 0.f && seeded < ubn) ubn = seeded;
        // (1 + 1e-4 on the square: fp32 rounding of the sums behind y)
        if (cw_ratio > 0.f && cw_ratio < 3.0e38f) {
            const float cwb = sqrtf(cw_ratio * (1.0f + 1e-4f)) * (1.0f + 1e-6f);
            if (cwb < ubn) ubn = cwb;
        }
        if (record) {
            float* rec = base + ScreenLayout::REC + 4 * j;
            rec[0] = lower;
            rec[1] = upper;
            rec[2] = (float)(verdict + 1);                    // 0 = no verdict yet
            rec[3] = ubh;
            __hip_atomic_store(base + ScreenLayout::UBH + j + 1, ubn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (verdict != 0)
                __hip_atomic_store(reinterpret_cast<int*>(base) + ScreenLayout::FLAG, verdict, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    verdict = __builtin_amdgcn_readfirstlane(verdict);
    ub = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ubn)));
    return verdict;
}
#endif

// n_sets: accumulator sets to clear (the solve's sweeps)
int launch_screen_init(onmf_ctx* c, float* base, int r, hipStream_t s, int n_sets, float* fill_out = nullptr,
                       size_t fill_n = 0, uint64_t fill_seed = 0);
// the verdicts that no sweep of the launch evaluated: sweeps k_from .. k_last, in order
// (start_dev not null: k_from = max(*start_dev, k_last - 1), nothing if *start_dev > k_last)
// cw_rows: screen_cw_rows(r) when the sweeps were the tile kernel's (they carry the Collatz-Wielandt sums);
// the kernel then also takes one power step of the positive vector on the last sweep's sums
int launch_screen_final(onmf_ctx* c, float* base, int k_from, int k_last, float tol, float rank_cap,
                        hipStream_t s, const int* start_dev = nullptr, int cw_rows = 0, float* rec_host = nullptr);
// r <= 64: settle the first inconclusive verdict in the records on the device - both Gram matrices, their
// top eigenvalues, the reference's comparison - and prepare the launch enqueued behind it (first sweep in
// ScreenLayout::NEXT, flag and accumulator sets reopened).  Does nothing if there is none to settle.
int launch_screen_resolve(onmf_ctx* c, float* base, int r, int b, float* const* hb, int sub_iter, float tol,
                          hipStream_t s);
bool screen_resolve_covers(int r);
// reopen the loop after the host settled sweep k - 1: flag = 0; ub = an upper bound of
// sigma_max(H before sweep k) if the host knows one, else -1
int launch_screen_clear_flag(onmf_ctx* c, float* base, int k, hipStream_t s, float ub = -1.0f);
// device-side: copy hb[done % 3] to dst, if that is another buffer, where done comes from the records
int launch_screen_finalize(onmf_ctx* c, const float* base, int sub_iter, float* const* hb, float* dst,
                           size_t n, hipStream_t s);
// one power step of the probe subspace on a sample of columns, then the probe update
int launch_screen_refresh(onmf_ctx* c, float* base, int r, int b, const float* Hnew, const float* Hold,
                          hipStream_t s);
// Collatz-Wielandt upper bound of sigma_max(H) for an entrywise nonnegative H = the code before sweep k:
// stored as that sweep's bound, and one power step of the vector it uses
int launch_screen_cw(onmf_ctx* c, float* base, int r, int b, const float* H, int k, hipStream_t s);

// part 0: the whole pass.  Parts 1 and 2 (dict_pass_splits): the first panels of atoms / the rest,
// as two launches with the same arguments; W_out is complete after part 2.
// `ws` (optional): caller-provided workspace of dict_workspace_floats(d, r) floats, 256-byte aligned,
// for the one-workgroup form (d <= 1024); the context's own buffers are then left alone and nothing
// is allocated.
int launch_dict_pass(onmf_ctx* c, int d, int r, const float* Win, const float* A, const float* B,
                     float* Wout, hipStream_t s, int part = 0, float* ws = nullptr);
bool dict_pass_splits(const onmf_ctx* c, int d, int r);
size_t dict_workspace_floats(int d, int r);
// The pass without any waiting between workgroups: two small launches per atom.  Slow (r round trips
// through the launch queue) but needs no co-residency: what a pass is redone with after its
// workgroups gave up waiting for each other.  Wout may be Win.
int launch_dict_pass_serial(onmf_ctx* c, int d, int r, const float* Win, const float* A, const float* B,
                            float* Wout, hipStream_t s);
// W^T as the last panel-kernel pass staged it (valid after launch_dict_pass on the panel path)
const float* dict_staged_wt(const onmf_ctx* c, int d, int r);

// Out (M x N, double) = A^T B for float operands stored contraction-major (A[k*lda + m], B[k*ldb + n])
int launch_prod_f64(onmf_ctx* c, int M, int N, int K, const float* A, int lda, const float* B, int ldb, double* Out,
                    int ldo, hipStream_t s);
int launch_prod_f64_pair(onmf_ctx* c, int M, int K, const float* A, int lda, const float* B1, int ldb1, int N1,
                         double* Out1, int ldo1, const float* B2, int ldb2, int N2, double* Out2, int ldo2,
                         hipStream_t s);
// scikit-learn's positive LARS-lasso path per column (lars.hip): G (r x r), Cd (r x b, leading dimension
// ldc) in double; H (r x b, leading dimension ldh) out; info: device int[4*b] or null
int launch_lars_pos(onmf_ctx* c, int r, int b, const double* G, const double* Cd, int ldc, double alpha,
                    double n_samples, int max_iter, float* H, int ldh, int* info, hipStream_t s);

// np.random.rand(n) continued on the device from NumPy's (key, pos); see mt19937.hip.  Synchronises s.
int launch_numpy_rand(onmf_ctx* c, const uint32_t* key_host, int pos, size_t n, float* out, uint32_t* key_out_host,
                      int* pos_out, hipStream_t s);
int numpy_rand_setup(onmf_ctx* c, const uint32_t* table_host, int rows, hipStream_t s);
long long numpy_rand_chunk_words();
int launch_fill_uniform(float* out, size_t n, uint64_t seed, hipStream_t s);
int launch_gather_columns(int d, int n, const float* X, int ldx, const long long* idx, int b,
                          float* Xb, hipStream_t s);
int launch_scatter_add_last(onmf_ctx* c, int r, int n, float* code, const long long* idx, int b,
                            const float* H, hipStream_t s);
int launch_axpy(float* y, const float* x, float a, size_t n, hipStream_t s);
int launch_sub(float* out, const float* a, const float* b, size_t n, hipStream_t s);
int launch_sub_stack(float* out, const float* a, const float* b, size_t n, hipStream_t s);
int launch_axpby(float* out, const float* old, float keep, const float* part, float w, size_t n, hipStream_t s);
int launch_unfold3(int n0, int n1, int n2, const float* T, int mode, int transpose, float* out,
                   hipStream_t s);
int launch_transpose(int rows, int cols, const float* in, float* out, hipStream_t s);
int launch_surrogate_partials(int d, int r, const float* Q, const float* W, const float* B, const float* Cagg,
                              double* partial, int blocks, hipStream_t s);

// host: largest eigenvalue of a symmetric PSD n x n matrix (row-major float), in double
double host_lambda_max(const float* S, int n);
// the same, returning early with any Ritz value >= enough (a lower bound of the eigenvalue) when enough > 0
double host_lambda_max_enough(const float* S, int n, double enough);
