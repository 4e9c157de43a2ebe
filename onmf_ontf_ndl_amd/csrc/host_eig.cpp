// Host side of the stopping test: largest eigenvalue of a symmetric positive semi-definite matrix
// (sigma_max(M)^2 of src/onmf.py:265 is this value for M M^T).  Lanczos with full re-orthogonalisation,
// the tridiagonal's top eigenvalue by Sturm-sequence bisection; all in double.  Plain C++ (no GPU code).
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

double host_lambda_max(const float* S, int n);
double host_lambda_max_enough(const float* S, int n, double enough);

namespace {

// number of eigenvalues of the symmetric tridiagonal (a, b) of order m that are < x (Sturm count)
int tridiag_below(const double* a, const double* b, int m, double x) {
    int cnt = 0;
    double q = a[0] - x;
    if (q < 0) ++cnt;
    for (int i = 1; i < m; ++i) {
        if (q == 0.0) q = 1e-300;
        q = a[i] - x - b[i - 1] * b[i - 1] / q;
        if (q < 0) ++cnt;
    }
    return cnt;
}

// top eigenvalue by bisection; `floor` is a known lower bound of it (or -inf)
double tridiag_top(const double* a, const double* b, int m, double floor) {
    if (m == 1) return a[0];
    double lo = std::numeric_limits<double>::infinity(), hi = -lo;
    for (int i = 0; i < m; ++i) {
        const double off = (i > 0 ? std::fabs(b[i - 1]) : 0.0) + (i + 1 < m ? std::fabs(b[i]) : 0.0);
        lo = std::min(lo, a[i] - off);
        hi = std::max(hi, a[i] + off);
    }
    if (floor > lo && floor <= hi) {
        lo = floor;
        // the value moves little between evaluations: look just above the floor first
        const double near = floor + 1e-9 * std::fabs(floor);
        if (near < hi && tridiag_below(a, b, m, near) >= m) hi = near;
    }
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (!(mid > lo && mid < hi)) break;                 // the interval is down to neighbouring doubles
        if (tridiag_below(a, b, m, mid) >= m) hi = mid; else lo = mid;
    }
    return 0.5 * (lo + hi);
}

double dot_plain(const double* x, const double* y, int n) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = 0;
    for (; i + 3 < n; i += 4) {
        s0 += x[i] * y[i];
        s1 += x[i + 1] * y[i + 1];
        s2 += x[i + 2] * y[i + 2];
        s3 += x[i + 3] * y[i + 3];
    }
    for (; i < n; ++i) s0 += x[i] * y[i];
    return (s0 + s1) + (s2 + s3);
}

void axpy_plain(double* w, double c, const double* v, int n) {
    for (int i = 0; i < n; ++i) w[i] -= c * v[i];
}

#if defined(__x86_64__)
__attribute__((target("avx2,fma"))) double dot_avx2(const double* x, const double* y, int n) {
    __m256d s0 = _mm256_setzero_pd(), s1 = _mm256_setzero_pd();
    int i = 0;
    for (; i + 7 < n; i += 8) {
        s0 = _mm256_fmadd_pd(_mm256_loadu_pd(x + i), _mm256_loadu_pd(y + i), s0);
        s1 = _mm256_fmadd_pd(_mm256_loadu_pd(x + i + 4), _mm256_loadu_pd(y + i + 4), s1);
    }
    for (; i + 3 < n; i += 4) s0 = _mm256_fmadd_pd(_mm256_loadu_pd(x + i), _mm256_loadu_pd(y + i), s0);
    s0 = _mm256_add_pd(s0, s1);
    double t[4];
    _mm256_storeu_pd(t, s0);
    double s = (t[0] + t[1]) + (t[2] + t[3]);
    for (; i < n; ++i) s += x[i] * y[i];
    return s;
}

__attribute__((target("avx2,fma"))) void axpy_avx2(double* w, double c, const double* v, int n) {
    const __m256d cc = _mm256_set1_pd(c);
    int i = 0;
    for (; i + 3 < n; i += 4)
        _mm256_storeu_pd(w + i, _mm256_fnmadd_pd(cc, _mm256_loadu_pd(v + i), _mm256_loadu_pd(w + i)));
    for (; i < n; ++i) w[i] -= c * v[i];
}
#endif

// out = (double)in, returns the sum of squares (several partial sums: not one long dependent chain)
double widen_plain(const float* in, double* out, size_t n) {
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 7 < n; i += 8)
        for (int k = 0; k < 8; ++k) {
            const double v = (double)in[i + k];
            out[i + k] = v;
            s[k] += v * v;
        }
    for (; i < n; ++i) {
        const double v = (double)in[i];
        out[i] = v;
        s[0] += v * v;
    }
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

#if defined(__x86_64__)
__attribute__((target("avx2,fma"))) double widen_avx2(const float* in, double* out, size_t n) {
    __m256d s0 = _mm256_setzero_pd(), s1 = _mm256_setzero_pd();
    size_t i = 0;
    for (; i + 7 < n; i += 8) {
        const __m256d a = _mm256_cvtps_pd(_mm_loadu_ps(in + i)), b = _mm256_cvtps_pd(_mm_loadu_ps(in + i + 4));
        _mm256_storeu_pd(out + i, a);
        _mm256_storeu_pd(out + i + 4, b);
        s0 = _mm256_fmadd_pd(a, a, s0);
        s1 = _mm256_fmadd_pd(b, b, s1);
    }
    s0 = _mm256_add_pd(s0, s1);
    double t[4];
    _mm256_storeu_pd(t, s0);
    double s = (t[0] + t[1]) + (t[2] + t[3]);
    for (; i < n; ++i) {
        const double v = (double)in[i];
        out[i] = v;
        s += v * v;
    }
    return s;
}
#endif

typedef double (*dot_fn)(const double*, const double*, int);
typedef double (*widen_fn)(const float*, double*, size_t);
typedef void (*axpy_fn)(double*, double, const double*, int);

struct Kernels {
    dot_fn dot = dot_plain;
    axpy_fn axpy = axpy_plain;
    widen_fn widen = widen_plain;
    Kernels() {
#if defined(__x86_64__)
        __builtin_cpu_init();
        if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) {
            dot = dot_avx2;
            axpy = axpy_avx2;
            widen = widen_avx2;
        }
#endif
    }
};

// `enough` > 0: return as soon as a Ritz value (always a lower bound of the top eigenvalue) reaches it.
// The top Ritz value only grows from step to step.  From step 4 on it is evaluated at every second
// step (bisection on the small tridiagonal, from the previous value upwards) and the run ends when two
// evaluations in a row agree to 1e-15 relative, i.e. it has not moved for four steps.  The Gram matrices
// of this path have a well separated top eigenvalue (H H^T of a nonnegative H: second eigenvalue a few
// percent of the first; the moves' Gram: about a tenth), so that happens after 8-14 steps, not n; a
// flat spectrum simply runs on, to n steps (exact) at the latest.  Small matrices (n <= 24) run all n
// steps: that costs less than the evaluations on the way.
double lambda_max_impl(const float* S, int n, double enough) {
    static const Kernels kern;
    if (n <= 0) return 0.0;
    if (n == 1) return (double)S[0];
    const int mmax = n;
    // work space kept per thread: no allocation after the first call at a size
    static thread_local std::vector<double> Sd, V, w, al, be;
    Sd.resize((size_t)n * n);
    V.resize((size_t)(mmax + 1) * n);
    w.resize(n);
    al.resize(mmax + 1);
    be.resize(mmax + 1);
    const double fro = kern.widen(S, Sd.data(), (size_t)n * n);
    if (!(fro > 0.0)) return (fro == 0.0) ? 0.0 : std::numeric_limits<double>::quiet_NaN();
    // deterministic start vector with a component along every coordinate
    double nv = 0.0;
    for (int i = 0; i < n; ++i) {
        V[i] = 1.0 + 0.5 * std::sin(12.9898 * (i + 1));
        nv += V[i] * V[i];
    }
    nv = std::sqrt(nv);
    for (int i = 0; i < n; ++i) V[i] /= nv;
    double prev = -1.0;
    const double ninf = -std::numeric_limits<double>::infinity();
    for (int j = 0; j < mmax; ++j) {
        const double* vj = &V[(size_t)j * n];
        for (int i = 0; i < n; ++i) w[i] = kern.dot(&Sd[(size_t)i * n], vj, n);
        al[j] = kern.dot(w.data(), vj, n);
        for (int pass = 0; pass < 2; ++pass) {
            for (int l = 0; l <= j; ++l) {
                const double* vl = &V[(size_t)l * n];
                kern.axpy(w.data(), kern.dot(w.data(), vl, n), vl, n);
            }
        }
        const double bn = std::sqrt(kern.dot(w.data(), w.data(), n));
        const int m = j + 1;
        if (enough > 0.0 && tridiag_below(al.data(), be.data(), m, enough) < m) {
            // a Ritz value >= enough exists: report it
            return tridiag_top(al.data(), be.data(), m, enough);
        }
        if (bn <= 1e-14 * std::sqrt(fro) || m == mmax)       // invariant subspace, or all n steps: exact
            return tridiag_top(al.data(), be.data(), m, prev > 0.0 ? prev : ninf);
        if (n > 24 && m >= 4 && (m & 1) == 0) {
            const double theta = tridiag_top(al.data(), be.data(), m, prev > 0.0 ? prev : ninf);
            if (prev > 0.0 && theta - prev <= 1e-15 * theta) return theta;
            prev = theta;
        }
        be[j] = bn;
        double* vn = &V[(size_t)(j + 1) * n];
        const double inv = 1.0 / bn;
        for (int i = 0; i < n; ++i) vn[i] = w[i] * inv;
    }
    return tridiag_top(al.data(), be.data(), mmax, prev);
}

}  // namespace

double host_lambda_max(const float* S, int n) { return lambda_max_impl(S, n, 0.0); }
double host_lambda_max_enough(const float* S, int n, double enough) { return lambda_max_impl(S, n, enough); }
