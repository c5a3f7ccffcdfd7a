/*
 * oracle/ls_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's other move operators:
 *   two_opt_cost / get_two_opt_tour / swap      search/local_search.py:9-53
 *   relocate_cost / get_relocate_tour / relocate search/local_search.py:56-105, search/relocate.py:5-38
 *   local_search                                 search/local_search.py:125-134
 * (the same functions appear again in ml4tsp-library/ml4tsp/nar/local_search/guided_ls.py:90-209).
 * Only tests/ may link or call this file; the product path never does.
 *
 * Parity status: PINNED by tests/golden/ls_*.npz, produced by importing the reference's
 * search/relocate.py and search/local_search.py (oracle/gen_golden_ls.py).
 *
 * Contract that is reproduced (all arithmetic in double, like scipy.spatial.distance_matrix of
 * float64 points, whose entries are sqrt(|dx|^2 + |dy|^2)):
 *   w(a,b)        = dist(a,b)                          plain
 *                 = dist(a,b) + k * penalty[a][b]      guided (edge_weight_guided, local_search.py:200)
 *   two-opt cost  = ((w(a,c) + w(b,d)) + w(a,b)) - w(c,d),  a,b,c,d = t[i], t[i-1], t[j], t[j-1], i < j
 *                   (the reference ADDS w(a,b), local_search.py:23; reproduced as written)
 *   relocate cost = ((((-w(a,b) - w(b,c)) + w(a,c)) - w(d,e)) + w(d,b)) + w(b,e),
 *                   a,b,c = t[i-1], t[i], t[i+1]; d,e = t[j], t[j+1] if i < j else t[j-1], t[j]
 *   a candidate replaces the best one iff delta < best_delta and not numpy.isclose(0, delta),
 *   i.e. not (|delta| <= 1e-8 + 1e-5 * |delta|); scan order: i ascending, then j ascending over
 *   positions 1 .. n-1 (two-opt: pairs i < j with j - i >= 2; relocate: i != j and i - j != 1;
 *   with fixed_i: j ascending, |i - j| >= 2 for two-opt, j != i for relocate), first best wins.
 *   apply: two-opt reverses positions lo .. hi-1 (swap(), local_search.py:9-14); relocate pops position
 *   i and inserts at position j.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const double* P;        /* n x 2 */
    const double* pen;      /* n x n or NULL */
    double k;
    int n;
} Ctx;

static double wgt(const Ctx* c, int a, int b)
{
    const double dx = fabs(c->P[2 * b] - c->P[2 * a]), dy = fabs(c->P[2 * b + 1] - c->P[2 * a + 1]);
    const double d = sqrt(dx * dx + dy * dy);
    if (!c->pen) return d;
    return d + c->k * c->pen[(size_t)a * c->n + b];
}

static int is_close0(double delta) { return fabs(delta) <= 1e-8 + 1e-5 * fabs(delta); }

static double two_opt_cost(const Ctx* c, const int64_t* t, int i, int j)
{
    if (i == j) return 0.0;
    if (j < i) { int x = i; i = j; j = x; }
    const int a = (int)t[i], b = (int)t[i - 1], cc = (int)t[j], d = (int)t[j - 1];
    return ((wgt(c, a, cc) + wgt(c, b, d)) + wgt(c, a, b)) - wgt(c, cc, d);
}

static double relocate_cost(const Ctx* c, const int64_t* t, int i, int j)
{
    if (i == j) return 0.0;
    const int a = (int)t[i - 1], b = (int)t[i], cc = (int)t[i + 1];
    int d, e;
    if (i < j) { d = (int)t[j]; e = (int)t[j + 1]; } else { d = (int)t[j - 1]; e = (int)t[j]; }
    return ((((-wgt(c, a, b) - wgt(c, b, cc)) + wgt(c, a, cc)) - wgt(c, d, e)) + wgt(c, d, b)) + wgt(c, b, e);
}

/* one operator call: op 0 = get_two_opt_tour, 1 = get_relocate_tour; fixed_i < 0: all moves.
 * Applies the best move to t (n+1 entries) when there is one; returns its delta (0.0: none). */
static double ls_step(const Ctx* c, int64_t* t, int op, int fixed_i)
{
    const int n = c->n;
    double best = 0.0;
    int bi = -1, bj = -1;
    for (int i = 1; i <= n - 1; ++i) {
        if (fixed_i >= 0 && i != fixed_i) continue;
        for (int j = 1; j <= n - 1; ++j) {
            double delta;
            if (op == 0) {
                if (fixed_i < 0 && j <= i) continue;            /* combinations: i < j */
                if (abs(i - j) < 2) continue;
                delta = two_opt_cost(c, t, i, j);
            } else {
                if (i == j) continue;
                if (fixed_i < 0 && i - j == 1) continue;        /* permutations, minus relocate i -> i-1 */
                delta = relocate_cost(c, t, i, j);
            }
            if (delta < best && !is_close0(delta)) { best = delta; bi = i; bj = j; }
        }
    }
    if (bi < 0) return 0.0;
    if (op == 0) {
        int lo = bi < bj ? bi : bj, hi = bi < bj ? bj : bi;     /* reverse positions lo .. hi-1 */
        for (int a = lo, b = hi - 1; a < b; ++a, --b) { const int64_t x = t[a]; t[a] = t[b]; t[b] = x; }
    } else {
        const int64_t node = t[bi];
        if (bi < bj) memmove(t + bi, t + bi + 1, sizeof(int64_t) * (size_t)(bj - bi));
        else memmove(t + bj + 1, t + bj, sizeof(int64_t) * (size_t)(bi - bj));
        t[bj] = node;
    }
    return best;
}

/* ---- exported ---------------------------------------------------------------------------------- */
/* one operator call per tour.  points (num_instances, n, 2) double; tours (num_tours, n+1) int64 in
 * place; fixed_i (num_tours) or NULL; guide_k (num_tours) + penalty (num_tours, n, n) or NULL */
int oracle_ls_step(const double* points, int64_t* tours, double* delta_out, int64_t num_tours, int32_t n,
                   int32_t tpi, int32_t op, const int32_t* fixed_i, const double* guide_k, const double* penalty)
{
    if (n < 2 || tpi < 1 || (op != 0 && op != 1)) return -1;
    for (int64_t k = 0; k < num_tours; ++k) {
        Ctx c = {points + (size_t)(k / tpi) * (size_t)n * 2, penalty ? penalty + (size_t)k * n * n : NULL,
                 guide_k ? guide_k[k] : 0.0, n};
        const double d = ls_step(&c, tours + (size_t)k * (size_t)(n + 1), op, fixed_i ? fixed_i[k] : -1);
        if (delta_out) delta_out[k] = d;
    }
    return 0;
}

/* local_search (local_search.py:125-134): alternate the two operators until a whole round improves
 * nothing; rounds_out[k] = rounds executed (the last one without improvement included), capped */
int oracle_local_search(const double* points, int64_t* tours, int32_t* rounds_out, int64_t num_tours,
                        int32_t n, int32_t tpi, int32_t max_rounds)
{
    if (n < 2 || tpi < 1) return -1;
#pragma omp parallel for schedule(dynamic)
    for (int64_t k = 0; k < num_tours; ++k) {
        Ctx c = {points + (size_t)(k / tpi) * (size_t)n * 2, NULL, 0.0, n};
        int64_t* t = tours + (size_t)k * (size_t)(n + 1);
        int rounds = 0, improved = 1;
        while (improved && rounds < max_rounds) {
            improved = 0;
            for (int op = 0; op < 2; ++op)
                if (ls_step(&c, t, op, -1) < 0.0) improved = 1;
            ++rounds;
        }
        if (rounds_out) rounds_out[k] = rounds;
    }
    return 0;
}
