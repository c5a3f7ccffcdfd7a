/*
 * oracle/greedy_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's greedy edge-merge decoder:
 *   search/greedy_search.py:19-74 (tsp_greedy: symmetrise, merge, walk) and
 *   search/cython_merge/sources/cython_merge.pyx:19-120 (merge_cython, find_begin/find_end).
 * Parity status: PINNED against tests/golden/greedy_*.npz (made by oracle/gen_golden_greedy.py with
 * the reference's own tsp_greedy + its Cython merge compiled from source).  Where two different
 * edges have exactly the same score the reference's order is numpy's unstable argsort order
 * (undefined); this restatement breaks such ties by the lower flat index i*N+j: parity for tied
 * scores is unpinned.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { double key; int32_t idx; } edge_t;

static int cmp_edge(const void* a, const void* b)
{
    const edge_t* x = (const edge_t*)a; const edge_t* y = (const edge_t*)b;
    /* ascending key, NaN last (numpy sort order), ties by flat index */
    const int xn = x->key != x->key, yn = y->key != y->key;
    if (xn != yn) return xn - yn;
    if (!xn) { if (x->key < y->key) return -1; if (x->key > y->key) return 1; }
    return (x->idx > y->idx) - (x->idx < y->idx);
}

static int find_root(int32_t* r, int i)          /* cython_merge.pyx:107-120 */
{
    int b = r[i];
    if (b != i) { b = find_root(r, b); r[i] = b; }
    return b;
}

/* one heat map: S (n x n doubles, already symmetrised in the heat map's dtype), points (n x 2) */
static void greedy_one(const double* pts, const double* S, int n, int64_t* tour)
{
    edge_t* e = (edge_t*)malloc(sizeof(edge_t) * (size_t)n * n);
    int32_t* rb = (int32_t*)malloc(sizeof(int32_t) * n), *re = (int32_t*)malloc(sizeof(int32_t) * n);
    int32_t* nb = (int32_t*)malloc(sizeof(int32_t) * 2 * n);     /* adjacency: two slots per node */
    for (int i = 0; i < n; ++i) {
        rb[i] = re[i] = i; nb[2 * i] = nb[2 * i + 1] = -1;
        for (int j = 0; j < n; ++j) {
            const double dx = pts[2 * i] - pts[2 * j], dy = pts[2 * i + 1] - pts[2 * j + 1];
            const double dist = sqrt(dx * dx + dy * dy);          /* np.linalg.norm(axis=-1) */
            e[i * n + j].key = -S[i * n + j] / dist;              /* cython_merge.pyx:37 */
            e[i * n + j].idx = i * n + j;
        }
    }
    qsort(e, (size_t)n * n, sizeof(edge_t), cmp_edge);
    int merges = 0;
    for (size_t k = 0; k < (size_t)n * n && merges < n - 1; ++k) {
        const int i = e[k].idx / n, j = e[k].idx % n;
        const int bi = find_root(rb, i), ei = find_root(re, i), bj = find_root(rb, j), ej = find_root(re, j);
        if (bi == bj) continue;                                   /* same partial route */
        if (i != bi && i != ei) continue;                         /* i is interior */
        if (j != bj && j != ej) continue;
        nb[2 * i + (nb[2 * i] >= 0)] = j;
        nb[2 * j + (nb[2 * j] >= 0)] = i;
        ++merges;
        if (i == bi && j == ej) { rb[bi] = bj; re[ej] = ei; }
        else if (i == ei && j == bj) { rb[bj] = bi; re[ei] = ej; }
        else if (i == bi && j == bj) { rb[bi] = ej; rb[bj] = ej; rb[ej] = ej; re[ej] = ei; re[bj] = ei; }
        else if (i == ei && j == ej) { re[ei] = bj; rb[bj] = bi; rb[ej] = bi; re[ej] = bj; re[bj] = bj; }
    }
    const int fb = find_root(rb, 0), fe = find_root(re, 0);       /* closing edge */
    if (fb != fe || n == 1) {
        nb[2 * fe + (nb[2 * fe] >= 0)] = fb;
        nb[2 * fb + (nb[2 * fb] >= 0)] = fe;
    }
    /* walk (greedy_search.py:65-73): from 0, always the neighbour that is not the previous node,
     * the larger index when both qualify */
    tour[0] = 0;
    for (int k = 1; k <= n; ++k) {
        const int cur = (int)tour[k - 1], prev = k > 1 ? (int)tour[k - 2] : -1;
        int best = -1;
        for (int s = 0; s < 2; ++s) { const int c = nb[2 * cur + s]; if (c >= 0 && c != prev && c > best) best = c; }
        tour[k] = best;
    }
    free(e); free(rb); free(re); free(nb);
}

/* heat: (p, n, n) float (dtype 0) or double (dtype 1); tours out: (p, n+1) int64.
 * S = heat + heat^T is formed in the heat map's dtype like numpy does (greedy_search.py:36-38). */
int oracle_greedy(const double* points, const void* heat, int64_t* tours, int32_t p, int32_t n, int32_t dtype)
{
    if (!points || !heat || !tours || n < 2 || p < 1) return -1;
    double* S = (double*)malloc(sizeof(double) * (size_t)n * n);
    for (int s = 0; s < p; ++s) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                if (dtype == 0) {
                    const float* h = (const float*)heat + (size_t)s * n * n;
                    const float v = h[i * n + j] + h[j * n + i];
                    S[i * n + j] = (double)v;
                } else {
                    const double* h = (const double*)heat + (size_t)s * n * n;
                    S[i * n + j] = h[i * n + j] + h[j * n + i];
                }
            }
        greedy_one(points, S, n, tours + (size_t)s * (n + 1));
    }
    free(S);
    return 0;
}
