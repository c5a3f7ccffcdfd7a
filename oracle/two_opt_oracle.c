/*
 * oracle/two_opt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of ML4TSPBench's batched best-improvement 2-opt
 * (reference: search/two_opt.py:5-51 `batched_two_opt_torch`, and the textually
 * identical loop in ml4tsp-library/ml4tsp/nar/local_search/two_opt.py:18-67).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or call this file.  The product path (ml4tspbench_amd/, libtsp2opt_hip.so)
 * never does.
 *
 * Parity status: PINNED.  the fixtures under tests/golden/ were produced by importing the
 * reference's search/two_opt.py in the build container (oracle/gen_golden.py);
 * tests/test_oracle_golden.py checks this file against every one of them
 * bit-for-bit.
 *
 * Arithmetic contract that is reproduced (search/two_opt.py:22-42):
 *   dist(a,b)  = sqrt((ax-bx)*(ax-bx) + (ay-by)*(ay-by))      separate sub, mul, add, sqrt; no FMA
 *   delta(i,j) = ((dist(t[i],t[j]) + dist(t[i+1],t[j+1])) - dist(t[i],t[i+1])) - dist(t[j],t[j+1])
 *   valid      = j >= i+2 (torch.triu(diagonal=2)), every other entry is 0.0
 *   argmin     = first flat index i*N+j attaining the minimum (masked zeros included,
 *                so "nothing strictly negative" gives index 0 -> i=j=0 -> empty reversal)
 *   apply      = reverse t[i+1..j]                                   (two_opt.py:42)
 *   stop       = min over the WHOLE batch >= -1e-6 (two_opt.py:35,40) or iteration cap (:47)
 * Compile with -ffp-contract=off (see oracle/Makefile); dist() is symmetric bit-for-bit
 * because (a-b)*(a-b) == (b-a)*(b-a), which lets one row of distances serve two rows
 * of deltas.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_VERSION 1

/* ---- one sweep, generated for double and float ------------------------------------- */
#define DEFINE_SWEEP(NAME, T, SQRT)                                                        \
/* px,py: coordinates in tour-position order, n+1 entries (entry n == entry 0).          \
 * cur,nxt: scratch rows of n+2 entries.  Returns the reference's                          \
 * torch.min(valid_change) in *min_out (0.0 when nothing is negative, NaN when a valid   \
 * entry is NaN) and the per-tour flat argmin in *idx_out. */                              \
static void NAME(const T* px, const T* py, int n, T* e, T* cur, T* nxt,                    \
                 T* min_out, int64_t* idx_out)                                             \
{                                                                                          \
    T best = (T)0.0;                                                                       \
    int64_t best_idx = 0;                                                                  \
    int has_nan = 0;                                                                       \
    for (int k = 0; k < n; ++k) {                                                          \
        T dx = px[k] - px[k + 1], dy = py[k] - py[k + 1];                                  \
        e[k] = SQRT(dx * dx + dy * dy);                                                    \
    }                                                                                      \
    /* cur[j] = dist(pos 0, pos j) for j = 2..n */                                         \
    for (int j = 2; j <= n; ++j) {                                                         \
        T dx = px[0] - px[j], dy = py[0] - py[j];                                          \
        cur[j] = SQRT(dx * dx + dy * dy);                                                  \
    }                                                                                      \
    for (int i = 0; i + 2 <= n - 1; ++i) {                                                 \
        /* nxt[j] = dist(pos i+1, pos j) for j = i+3..n */                                 \
        const T ax = px[i + 1], ay = py[i + 1];                                            \
        for (int j = i + 3; j <= n; ++j) {                                                 \
            T dx = ax - px[j], dy = ay - py[j];                                            \
            nxt[j] = SQRT(dx * dx + dy * dy);                                              \
        }                                                                                  \
        const T ei = e[i];                                                                 \
        for (int j = i + 2; j <= n - 1; ++j) {                                             \
            T d = ((cur[j] + nxt[j + 1]) - ei) - e[j];                                     \
            if (d < best) { best = d; best_idx = (int64_t)i * n + j; }                     \
            else if (d != d) has_nan = 1;                                                  \
        }                                                                                  \
        T* t = cur; cur = nxt; nxt = t;                                                    \
    }                                                                                      \
    *min_out = has_nan ? (T)NAN : best;                                                    \
    *idx_out = best_idx;                                                                   \
}

static inline float sqrt_f32(float x) { return sqrtf(x); }
DEFINE_SWEEP(sweep_f64, double, sqrt)
DEFINE_SWEEP(sweep_f32, float, sqrt_f32)

static void reverse_i64(int64_t* t, int64_t lo, int64_t hi)
{
    while (lo < hi) { int64_t a = t[lo]; t[lo] = t[hi]; t[hi] = a; ++lo; --hi; }
}

/* ---- one group of `p` tours that share one coordinate set ---------------------------- */
/* Returns the reference's `iterator` (number of applied iterations).  sweeps_out gets the
 * number of full sweeps executed (iterator+1 when it converged, == iterator when capped). */
#define DEFINE_GROUP(NAME, T, SWEEP)                                                       \
static int NAME(const T* pts /* n x 2, node order */, int64_t* tours /* p x (n+1) */,      \
                int p, int n, int max_iterations, int32_t* sweeps_out)                     \
{                                                                                          \
    int iterator = 0, sweeps = 0;                                                          \
    if (n < 1 || p < 1) { if (sweeps_out) *sweeps_out = 0; return 0; }                     \
    T* buf = (T*)malloc(sizeof(T) * (size_t)(5 * (n + 2)));                                \
    T *px = buf, *py = buf + (n + 2), *e = buf + 2 * (n + 2),                              \
      *cur = buf + 3 * (n + 2), *nxt = buf + 4 * (n + 2);                                  \
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)p);                          \
    for (;;) {                                                                             \
        T gmin = (T)0.0; int gnan = 0;                                                     \
        for (int b = 0; b < p; ++b) {                                                      \
            int64_t* t = tours + (size_t)b * (n + 1);                                      \
            for (int k = 0; k <= n; ++k) { px[k] = pts[2 * t[k]]; py[k] = pts[2 * t[k] + 1]; } \
            T m; SWEEP(px, py, n, e, cur, nxt, &m, &idx[b]);                               \
            if (m != m) gnan = 1; else if (m < gmin) gmin = m;                             \
        }                                                                                  \
        ++sweeps;                                                                          \
        /* two_opt.py:40  `if min_change < -1e-6` (NaN compares false -> break) */        \
        if (!gnan && (double)gmin < -1e-6) {                                               \
            for (int b = 0; b < p; ++b) {                                                  \
                int64_t i = idx[b] / n, j = idx[b] % n;                                    \
                reverse_i64(tours + (size_t)b * (n + 1), i + 1, j);                        \
            }                                                                              \
            ++iterator;                                                                    \
        } else break;                                                                      \
        if (iterator >= max_iterations) break;                                             \
    }                                                                                      \
    free(buf); free(idx);                                                                  \
    if (sweeps_out) *sweeps_out = sweeps;                                                  \
    return iterator;                                                                       \
}

DEFINE_GROUP(group_f64, double, sweep_f64)
DEFINE_GROUP(group_f32, float, sweep_f32)

/*
 * oracle_two_opt -- host pointers throughout.
 *   points            (num_instances, n, 2) row-major, float (dtype 0) or double (dtype 1)
 *   tours             (num_tours, n+1) int64, updated in place
 *   tours_per_instance tour k uses instance k / tours_per_instance (base.py:61 mapping)
 *   group_size        1 = independent stop per tour (the plugin, two_opt.py:33);
 *                     P = consecutive P tours stop together (search/two_opt.py:35,40)
 *   iters_out         per tour: the `iterator` of the call that contained it
 *   sweeps_out        per tour: full sweeps executed by that call (may be NULL)
 *   threads           OpenMP threads over groups (<=1: serial)
 * Returns 0, or a negative code for bad arguments.
 */
int oracle_two_opt(const void* points, int64_t* tours, int32_t* iters_out, int32_t* sweeps_out,
                   int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t group_size,
                   int32_t max_iterations, int32_t dtype, int32_t threads)
{
    if (!points || !tours || num_tours < 0 || n < 1) return -1;
    if (tours_per_instance < 1 || group_size < 1) return -2;
    if (tours_per_instance % group_size != 0 || num_tours % group_size != 0) return -3;
    if (dtype != 0 && dtype != 1) return -4;
    const int64_t groups = num_tours / group_size;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int64_t g = 0; g < groups; ++g) {
        const int64_t first = g * group_size;
        const int64_t inst = first / tours_per_instance;
        int64_t* t = tours + (size_t)first * (n + 1);
        int32_t sweeps = 0;
        int it;
        if (dtype == 1)
            it = group_f64((const double*)points + (size_t)inst * n * 2, t, group_size, n,
                           max_iterations, &sweeps);
        else
            it = group_f32((const float*)points + (size_t)inst * n * 2, t, group_size, n,
                           max_iterations, &sweeps);
        for (int b = 0; b < group_size; ++b) {
            if (iters_out) iters_out[first + b] = it;
            if (sweeps_out) sweeps_out[first + b] = sweeps;
        }
    }
    return 0;
}

/* Closed-tour length in double from the given dtype's coordinates (utils/evaluator.py:5-13
 * evaluates sum of Euclidean edge lengths). */
int oracle_tour_lengths(const void* points, const int64_t* tours, double* lengths_out,
                        int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t dtype)
{
    if (!points || !tours || !lengths_out || n < 1 || tours_per_instance < 1) return -1;
    for (int64_t k = 0; k < num_tours; ++k) {
        const int64_t inst = k / tours_per_instance;
        const int64_t* t = tours + (size_t)k * (n + 1);
        double s = 0.0;
        for (int q = 0; q < n; ++q) {
            double ax, ay, bx, by;
            if (dtype == 1) {
                const double* p = (const double*)points + (size_t)inst * n * 2;
                ax = p[2 * t[q]]; ay = p[2 * t[q] + 1]; bx = p[2 * t[q + 1]]; by = p[2 * t[q + 1] + 1];
            } else {
                const float* p = (const float*)points + (size_t)inst * n * 2;
                ax = p[2 * t[q]]; ay = p[2 * t[q] + 1]; bx = p[2 * t[q + 1]]; by = p[2 * t[q + 1] + 1];
            }
            s += sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by));
        }
        lengths_out[k] = s;
    }
    return 0;
}

int oracle_version(void) { return ORACLE_VERSION; }

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
