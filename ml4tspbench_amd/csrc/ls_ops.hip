// ml4tspbench_amd/csrc/ls_ops.hip -- the reference's other move operators on gfx950
//
// Replaces (reference, read-only): get_two_opt_tour / get_relocate_tour and their helpers
// (search/local_search.py:9-105, search/relocate.py:5-38; the same functions again in
// ml4tsp-library/ml4tsp/nar/local_search/guided_ls.py:90-197) and the local_search loop that alternates
// them (search/local_search.py:125-134).  The reference evaluates every candidate move in a Python
// double loop over a scipy distance matrix; here one workgroup owns one tour (tour and coordinates in
// LDS), its 256 lanes evaluate the candidate moves of whole rows i with distances recomputed on the fly,
// and the best move is the lexicographic minimum of (delta, i*n + j) -- the reference's "first strict
// improvement in scan order".  Everything is exact double arithmetic in the reference's operation order
// (no filter: these operators are O(N^2) per call and are called O(N) times, not the hot path):
//   w(a,b)        = sqrt(dx*dx + dy*dy)                       scipy.spatial.distance_matrix
//                 = that + k * penalty[a][b]                  guided weights (local_search.py:200)
//   two-opt cost  = ((w(a,c) + w(b,d)) + w(a,b)) - w(c,d)     local_search.py:17-24, as written there
//   relocate cost = ((((-w(a,b) - w(b,c)) + w(a,c)) - w(d,e)) + w(d,b)) + w(b,e)   relocate.py:5-15
//   accepted iff delta < best and not numpy.isclose(0, delta)  (|delta| <= 1e-8 + 1e-5 |delta|)
// Compile with -ffp-contract=off.

#include <hip/hip_runtime.h>

#include <cstdint>

#include "tsp2opt.h"
#include "host_util.hpp"
#include "wave_reduce.hpp"

#pragma clang fp contract(off)

namespace t2o {

struct LsParams {
    const double* points;
    int64_t* tours;
    double* delta_out;
    int32_t* rounds_out;
    int32_t* status;
    const int32_t* fixed_i;
    const double* guide_k;
    const double* penalty;
    int64_t num_tours;
    int n, tpi;
    int op;                 // 0 two-opt, 1 relocate, 2 local_search loop
    int max_rounds;
    int xlds;               // coordinates in LDS
    int flat;               // edge-weight tables in LDS: candidates of a full scan are spread over all lanes
};

struct LsCtx {
    const uint16_t* t;      // tour positions (LDS)
    const double* X;        // coordinates, node order (LDS or global)
    const double* pen;      // penalty matrix of this tour or null
    double k;
    int n;
    double* ewf;            // flat scans: ewf[p] = w(t[p-1], t[p]), ewr[p] = w(t[p], t[p-1]), p = 1 .. n, and
    double* ewr;            // head[i] = (-w(a,b) - w(b,c)) + w(a,c) of relocate's node i; null: rows per lane
    double* head;
};

template <bool G>
__device__ __forceinline__ double wgt(const LsCtx& c, int a, int b)
{
    const double dx = c.X[2 * b] - c.X[2 * a], dy = c.X[2 * b + 1] - c.X[2 * a + 1];
    const double sx = dx * dx, sy = dy * dy;
    const double d = __builtin_sqrt(sx + sy);
    if constexpr (!G) return d;
    else {
        const double kp = c.k * c.pen[(size_t)a * c.n + b];
        return d + kp;
    }
}

__device__ __forceinline__ bool is_close0(double delta)
{
    const double r = 1e-5 * __builtin_fabs(delta);
    return __builtin_fabs(delta) <= 1e-8 + r;
}

// best move of operator `op` (0 two-opt, 1 relocate); fixed < 0: all moves.  Returns in every thread the
// delta (0.0: none) and key = i*n + j.  red: LDS scratch of 2 * (nt/64) doubles.
template <bool G>
__device__ void ls_best(const LsCtx& c, int op, int fixed, int tid, int nt, double* red, double& e, unsigned& key)
{
    const int n = c.n;
    e = 0.0;
    key = 0xffffffffu;
    auto consider = [&](double delta, int i, int j) {
        if (delta < e && !is_close0(delta)) { e = delta; key = (unsigned)i * (unsigned)n + (unsigned)j; }
    };
    auto row_two_opt = [&](int i, int j0, int jstep) {
        // i fixed; for j > i: (lo, hi) = (i, j); for j < i (fixed_i only): (j, i)
        for (int j = j0; j <= n - 1; j += jstep) {
            const int d = i - j;
            if (d < 2 && d > -2) continue;
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const int a = c.t[lo], b = c.t[lo - 1], cc = c.t[hi], dd = c.t[hi - 1];
            const double delta = ((wgt<G>(c, a, cc) + wgt<G>(c, b, dd)) + wgt<G>(c, a, b)) - wgt<G>(c, cc, dd);
            consider(delta, i, j);
        }
    };
    auto row_relocate = [&](int i, int j0, int jstep, bool skip_prev) {
        const int a = c.t[i - 1], b = c.t[i], cc = c.t[i + 1];
        const double head = (-wgt<G>(c, a, b) - wgt<G>(c, b, cc)) + wgt<G>(c, a, cc);
        for (int j = j0; j <= n - 1; j += jstep) {
            if (j == i || (skip_prev && i - j == 1)) continue;
            const int d = i < j ? c.t[j] : c.t[j - 1], ee = i < j ? c.t[j + 1] : c.t[j];
            const double delta = ((head - wgt<G>(c, d, ee)) + wgt<G>(c, d, b)) + wgt<G>(c, b, ee);
            consider(delta, i, j);
        }
    };
    if (fixed < 0 && c.ewf) {
        // Full scan, flat: every candidate costs two square roots -- the weights of the tour's own edges (and
        // relocate's three per node) are tabulated once per scan, bit-identical to recomputing them -- and the
        // candidates are spread evenly over the lanes.  The reference's "first strict improvement in scan
        // order" is the lexicographic minimum of (delta, i n + j) over the admissible candidates.
        const int m = n - 1;                            // positions 1 .. m take part
        auto consider_any = [&](double delta, int i, int j) {
            const unsigned kk = (unsigned)i * (unsigned)n + (unsigned)j;
            if (delta < 0.0 && !is_close0(delta) && better(delta, kk, e, key)) { e = delta; key = kk; }
        };
        __syncthreads();                                // the tour may just have been rewritten
        if (op == 0) {
            for (int q = 1 + tid; q <= n; q += nt) c.ewr[q] = wgt<G>(c, c.t[q], c.t[q - 1]);
            __syncthreads();
            // pairs u < v of M = m - 1 items (i = u + 1, j = v + 2): rows r and M-2-r of the triangle share a
            // line of M cells
            const int M = m - 1, lines = M > 1 ? M >> 1 : 0, cells = lines * M;
            const int mid = (M & 1) ? -1 : (M - 2) >> 1;
            for (int f = tid; f < cells; f += nt) {
                const int r = f / M, cc0 = f - r * M;
                int u = r, v = r + 1 + cc0;
                if (cc0 >= M - 1 - r) {
                    if (r == mid) continue;
                    u = M - 2 - r;
                    v = u + 1 + (cc0 - (M - 1 - r));
                }
                const int lo = u + 1, hi = v + 2;
                const int a = c.t[lo], b = c.t[lo - 1], cc = c.t[hi], dd = c.t[hi - 1];
                consider_any(((wgt<G>(c, a, cc) + wgt<G>(c, b, dd)) + c.ewr[lo]) - c.ewr[hi], lo, hi);
            }
        } else {
            for (int q = 1 + tid; q <= n; q += nt) c.ewf[q] = wgt<G>(c, c.t[q - 1], c.t[q]);
            __syncthreads();
            for (int i = 1 + tid; i <= m; i += nt)
                c.head[i] = (-c.ewf[i] - c.ewf[i + 1]) + wgt<G>(c, c.t[i - 1], c.t[i + 1]);
            __syncthreads();
            for (int f = tid; f < m * m; f += nt) {
                const int i = f / m + 1, j = f - (i - 1) * m + 1;
                if (j == i || i - j == 1) continue;
                const int b = c.t[i];
                const int d = i < j ? c.t[j] : c.t[j - 1], ee = i < j ? c.t[j + 1] : c.t[j];
                const double wde = i < j ? c.ewf[j + 1] : c.ewf[j];
                consider_any(((c.head[i] - wde) + wgt<G>(c, d, b)) + wgt<G>(c, b, ee), i, j);
            }
        }
    } else if (fixed >= 0) {                    // one row, the lanes share its columns (ascending per lane)
        if (fixed >= 1 && fixed <= n - 1) {
            if (op == 0) row_two_opt(fixed, 1 + tid, nt);
            else row_relocate(fixed, 1 + tid, nt, false);
        }
    } else {                                    // a lane takes whole rows, ascending
        for (int i = 1 + tid; i <= n - 1; i += nt) {
            if (op == 0) row_two_opt(i, i + 2, 1);
            else row_relocate(i, 1, 1, true);
        }
    }
    wave_best(e, key);
    const int nw = nt >> 6, wv = tid >> 6;
    if (nw > 1) {
        unsigned* rk = reinterpret_cast<unsigned*>(red + nw);
        __syncthreads();
        if ((tid & 63) == 0) { red[wv] = e; rk[wv] = key; }
        __syncthreads();
        e = red[0];
        key = rk[0];
        for (int w = 1; w < nw; ++w)
            if (better(red[w], rk[w], e, key)) { e = red[w]; key = rk[w]; }
    }
}

// apply move (i, j) of operator op to the tour in LDS (all threads; ends with a barrier)
__device__ void ls_apply(uint16_t* t, int n, int op, unsigned key, int tid, int nt)
{
    const int i = (int)(key / (unsigned)n), j = (int)(key - (unsigned)i * (unsigned)n);
    __syncthreads();
    if (op == 0) {                              // swap(): reverse positions lo .. hi-1
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int h = (hi - lo) >> 1;
        for (int k = tid; k < h; k += nt) {
            const uint16_t x = t[lo + k], y = t[hi - 1 - k];
            t[lo + k] = y;
            t[hi - 1 - k] = x;
        }
    } else {                                    // relocate(): pop position i, insert at position j
        const uint16_t node = t[i];
        const int lo = i < j ? i : j, hi = i < j ? j : i, len = hi - lo + 1;
        // every lane reads its source entries, then all write (the ranges overlap)
        // (chunks of nt positions, walked away from the side the sources lie on: a chunk never reads what
        //  an earlier chunk wrote)
        for (int k0 = 0; k0 < len; k0 += nt) {
            const int p = i < j ? lo + k0 + tid : hi - k0 - tid;
            uint16_t v = 0;
            const bool mine = p >= lo && p <= hi;
            if (mine) v = p == j ? node : (i < j ? t[p + 1] : t[p - 1]);
            __syncthreads();
            if (mine) t[p] = v;
            __syncthreads();
        }
    }
    __syncthreads();
}

template <bool G>
__global__ void __launch_bounds__(256) k_ls(LsParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = p.n, tid = threadIdx.x, nt = blockDim.x;
    const int64_t tk = blockIdx.x;
    if (tk >= p.num_tours) return;
    double* red = reinterpret_cast<double*>(smem);                  // 16 doubles
    int* flag = reinterpret_cast<int*>(smem + 128);
    uint16_t* t = reinterpret_cast<uint16_t*>(smem + 144);
    size_t off = 144 + ((sizeof(uint16_t) * (size_t)(n + 1) + 15) & ~(size_t)15);
    double* xl = reinterpret_cast<double*>(smem + off);
    const double* X = p.points + (size_t)(tk / p.tpi) * (size_t)n * 2;
    int64_t* gt = p.tours + (size_t)tk * (size_t)(n + 1);
    if (tid == 0) *flag = 0;
    __syncthreads();
    int bad = 0;
    for (int k = tid; k <= n; k += nt) {
        const int64_t v = gt[k];
        if (v < 0 || v >= n) { bad = 1; t[k] = 0; } else t[k] = (uint16_t)v;
    }
    if (p.xlds)
        for (int k = tid; k < 2 * n; k += nt) xl[k] = X[k];
    if (bad) atomicOr(flag, 1);
    __syncthreads();
    if (*flag) {                                // an entry outside [0, n): tour left untouched
        if (tid == 0) {
            if (p.status) p.status[tk] = TSP2OPT_ST_BAD_INDEX;
            if (p.delta_out) p.delta_out[tk] = 0.0;
            if (p.rounds_out) p.rounds_out[tk] = 0;
        }
        return;
    }
    LsCtx c;
    c.t = t;
    c.X = p.xlds ? xl : X;
    c.pen = G ? p.penalty + (size_t)tk * (size_t)n * (size_t)n : nullptr;
    c.k = G ? p.guide_k[tk] : 0.0;
    c.n = n;
    c.ewf = c.ewr = c.head = nullptr;
    if (p.flat) {
        double* tab = xl + 2 * (size_t)n;
        c.ewf = tab;
        c.ewr = tab + (n + 2);
        c.head = tab + 2 * (size_t)(n + 2);
    }
    double e;
    unsigned key;
    int rounds = 0;
    bool changed = false;
    if (p.op == 2) {                            // local_search(): both operators until a round improves nothing
        bool improved = true;
        while (improved && rounds < p.max_rounds) {
            improved = false;
            for (int op = 0; op < 2; ++op) {
                ls_best<G>(c, op, -1, tid, nt, red, e, key);
                if (e < 0.0) {
                    ls_apply(t, n, op, key, tid, nt);
                    improved = true;
                    changed = true;
                }
            }
            ++rounds;
        }
        e = 0.0;
    } else {
        ls_best<G>(c, p.op, p.fixed_i ? p.fixed_i[tk] : -1, tid, nt, red, e, key);
        if (e < 0.0) {
            ls_apply(t, n, p.op, key, tid, nt);
            changed = true;
        }
    }
    if (changed)
        for (int k = tid; k <= n; k += nt) gt[k] = (int64_t)t[k];
    if (tid == 0) {
        if (p.status) p.status[tk] = (p.op == 2 && rounds >= p.max_rounds) ? TSP2OPT_ST_CAPPED : 0;
        if (p.delta_out) p.delta_out[tk] = e;
        if (p.rounds_out) p.rounds_out[tk] = rounds;
    }
}


// ------------------------------------------------------------------------------------------
// guided local search: whole outer rounds on the device (search/local_search.py:176-231, tsp_gls)
// ------------------------------------------------------------------------------------------
// One workgroup per tour keeps the tour and the coordinates in LDS for the whole call and runs, per outer
// round, the reference's perturbation loop -- utility argmax over the tour's edges (:192-202), penalty update
// (:206-208), the two operators with the penalised edge's end nodes as fixed rows under the guided weights
// (:212-223) until `perturbation_moves` moves were made -- then local_search under the plain weights (:226) and
// the cost comparison that keeps the best tour (:227-229).  Penalties, regrets and the best tour live in device
// memory between calls; the host only decides how many rounds to run (a round count, or one round per launch
// against the wall clock).  `init`: the part in front of the loop (:178-182): k = 0.1 * cost / n, a first
// local_search, best = that.
struct GlsParams {
    const double* points;
    int64_t* tours;             // current tour of every search, in / out
    int64_t* best_tours;        // best tour so far, in / out
    double* best_cost;
    double* guide_k;
    double* penalty;            // (T, n, n)
    const double* regret;       // (T / tph, n, n): max(-heat, 0)
    int32_t* status;
    int64_t num_tours;
    int n, tpi, tph;
    int pm;                     // perturbation_moves
    int rounds;
    int init;
    int xlds, flat;
};

// closed-tour cost exactly as TSPEvaluator.evaluate sums it (utils/evaluator.py:9-13): edge by edge in tour
// order, in double; thread 0 sums, every thread gets the value
__device__ double gls_cost(const LsCtx& c, double* red, int tid)
{
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int q = 0; q < c.n; ++q) s += wgt<false>(c, c.t[q], c.t[q + 1]);
        red[0] = s;
    }
    __syncthreads();
    const double s = red[0];
    __syncthreads();
    return s;
}

__device__ void gls_local_search(const LsCtx& c, uint16_t* t, int tid, int nt, double* red)
{
    bool improved = true;
    while (improved) {
        improved = false;
        for (int op = 0; op < 2; ++op) {
            double e;
            unsigned key;
            ls_best<false>(c, op, -1, tid, nt, red, e, key);
            if (e < 0.0) {
                ls_apply(t, c.n, op, key, tid, nt);
                improved = true;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_gls(GlsParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = p.n, tid = threadIdx.x, nt = blockDim.x;
    const int64_t tk = blockIdx.x;
    if (tk >= p.num_tours) return;
    double* red = reinterpret_cast<double*>(smem);                  // 16 doubles
    int* flag = reinterpret_cast<int*>(smem + 128);                 // [0] bad input, [1] position found
    uint16_t* t = reinterpret_cast<uint16_t*>(smem + 144);
    size_t off = 144 + ((sizeof(uint16_t) * (size_t)(n + 1) + 15) & ~(size_t)15);
    double* xl = reinterpret_cast<double*>(smem + off);
    const double* X = p.points + (size_t)(tk / p.tpi) * (size_t)n * 2;
    int64_t* gt = p.tours + (size_t)tk * (size_t)(n + 1);
    int64_t* gb = p.best_tours + (size_t)tk * (size_t)(n + 1);
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    __syncthreads();
    int bad = 0;
    for (int q = tid; q <= n; q += nt) {
        const int64_t v = gt[q];
        if (v < 0 || v >= n) { bad = 1; t[q] = 0; } else t[q] = (uint16_t)v;
    }
    if (p.xlds)
        for (int q = tid; q < 2 * n; q += nt) xl[q] = X[q];
    if (bad) atomicOr(&flag[0], 1);
    __syncthreads();
    if (flag[0]) {                              // an entry outside [0, n): tour left untouched
        if (tid == 0 && p.status) p.status[tk] = TSP2OPT_ST_BAD_INDEX;
        return;
    }
    double* pen = p.penalty + (size_t)tk * (size_t)n * (size_t)n;
    const double* reg = p.regret + (size_t)(tk / p.tph) * (size_t)n * (size_t)n;
    LsCtx cp;                                   // plain weights (local_search, costs)
    cp.t = t;
    cp.X = p.xlds ? xl : X;
    cp.pen = nullptr;
    cp.k = 0.0;
    cp.n = n;
    cp.ewf = cp.ewr = cp.head = nullptr;
    if (p.flat) {
        double* tab = xl + 2 * (size_t)n;
        cp.ewf = tab;
        cp.ewr = tab + (n + 2);
        cp.head = tab + 2 * (size_t)(n + 2);
    }
    double k, best;
    if (p.init) {
        const double c0 = gls_cost(cp, red, tid);
        k = 0.1 * c0 / (double)n;                                   // local_search.py:179
        gls_local_search(cp, t, tid, nt, red);                      // :180
        best = gls_cost(cp, red, tid);                              // :181-182
        __syncthreads();
        for (int q = tid; q <= n; q += nt) gb[q] = (int64_t)t[q];
    } else {
        k = p.guide_k[tk];
        best = p.best_cost[tk];
    }
    LsCtx cg = cp;                              // guided weights: dist + k * penalty (:209)
    cg.pen = pen;
    cg.k = k;
    cg.ewf = cg.ewr = cg.head = nullptr;        // fixed rows only: no tables
    unsigned* rk = reinterpret_cast<unsigned*>(red + 8);
    int st = 0;
    for (int r = 0; r < p.rounds; ++r) {
        int moves = 0, guard = 0;
        while (moves < p.pm) {
            // the tour edge of maximum utility regret / (1 + penalty), the first one among equals (:192-202)
            double bu = -1.0;
            unsigned bp = 0xffffffffu;
            for (int q = tid; q < n; q += nt) {
                const int a = t[q], b = t[q + 1];
                const size_t e = (size_t)(a < b ? a : b) * (size_t)n + (size_t)(a < b ? b : a);
                const double u = reg[e] / (1.0 + pen[e]);
                if (u > bu) { bu = u; bp = (unsigned)q; }           // ascending q per lane: strict > keeps the first
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const double ou = __shfl_xor(bu, o);
                const unsigned op_ = (unsigned)__shfl_xor((int)bp, o);
                if (ou > bu || (ou == bu && op_ < bp)) { bu = ou; bp = op_; }
            }
            __syncthreads();
            if ((tid & 63) == 0) { red[tid >> 6] = bu; rk[tid >> 6] = bp; }
            __syncthreads();
            bu = red[0];
            bp = rk[0];
            for (int w = 1; w < (nt >> 6); ++w)
                if (red[w] > bu || (red[w] == bu && rk[w] < bp)) { bu = red[w]; bp = rk[w]; }
            const int ea = t[bp], eb = t[bp + 1];
            const int src = ea < eb ? ea : eb, dst = ea < eb ? eb : ea;
            __syncthreads();
            if (tid == 0) {                                         // :206-208
                pen[(size_t)src * n + dst] += 1.0;
                pen[(size_t)dst * n + src] += 1.0;
                __threadfence();
            }
            __syncthreads();
            for (int which = 0; which < 2; ++which) {               // :212-223
                const int node = which == 0 ? src : dst;
                if (node == 0) continue;                            // not the depot
                for (int q = 1 + tid; q < n; q += nt)
                    if (t[q] == node) flag[1] = q;                  // cur_tour.index(n): its one position
                __syncthreads();
                const int i = flag[1];
                for (int op = 0; op < 2; ++op) {                    // the same row i for both operators
                    double e;
                    unsigned key;
                    ls_best<true>(cg, op, i, tid, nt, red, e, key);
                    if (e < 0.0) {
                        ls_apply(t, n, op, key, tid, nt);
                        ++moves;
                    }
                }
                __syncthreads();
            }
            if (++guard > 100000) { st = TSP2OPT_ST_CAPPED; break; }    // the reference would spin for ever
        }
        gls_local_search(cp, t, tid, nt, red);                      // :226
        const double c = gls_cost(cp, red, tid);                    // :227
        if (c < best) {                                             // :228-229
            best = c;
            for (int q = tid; q <= n; q += nt) gb[q] = (int64_t)t[q];
        }
    }
    __syncthreads();
    for (int q = tid; q <= n; q += nt) gt[q] = (int64_t)t[q];
    if (tid == 0) {
        p.guide_k[tk] = k;
        p.best_cost[tk] = best;
        if (p.status) p.status[tk] = st;
    }
}

int ls_launch(LsParams p, int device, hipStream_t stream)
{
    const size_t base = 144 + ((sizeof(uint16_t) * (size_t)(p.n + 1) + 15) & ~(size_t)15);
    size_t lds = base;
    p.xlds = 0;
    if (base + 16 * (size_t)p.n <= 160 * 1024) { p.xlds = 1; lds += 16 * (size_t)p.n; }
    p.flat = 0;
    if (p.xlds && lds + 24 * (size_t)(p.n + 2) <= 64 * 1024) { p.flat = 1; lds += 24 * (size_t)(p.n + 2); }
    if (lds > 160 * 1024) return fail(TSP2OPT_E_UNSUPPORTED, "n too large for the local-search operators");
    const bool guided = p.guide_k != nullptr;
    auto kern = guided ? k_ls<true> : k_ls<false>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)p.num_tours), dim3(256), lds, stream, p);
    HIP_TRY(hipGetLastError());
    (void)device;
    return 0;
}

int ls_check(const double* points, const int64_t* tours, int64_t num_tours, int32_t n, int32_t tpi)
{
    if (num_tours < 0 || n < 2 || tpi < 1) return fail(TSP2OPT_E_ARG, "bad size");
    if (n > TSP2OPT_MAX_N) return fail(TSP2OPT_E_UNSUPPORTED, "n > 65535 is not supported");
    if (num_tours > 0x7fffffffll) return fail(TSP2OPT_E_UNSUPPORTED, "too many tours");
    if (num_tours > 0 && (!points || !tours)) return fail(TSP2OPT_E_ARG, "null points/tours");
    if ((reinterpret_cast<uintptr_t>(points) % 8) || (reinterpret_cast<uintptr_t>(tours) % 8))
        return fail(TSP2OPT_E_ARG, "misaligned pointer (arrays must be aligned to their element size)");
    return 0;
}

}  // namespace t2o

using namespace t2o;

extern "C" int tsp2opt_ls_step(const double* points, int64_t* tours, double* delta_out, int32_t* status_out,
                               int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t op,
                               const int32_t* fixed_i, const double* guide_k, const double* penalty,
                               int32_t device, void* hip_stream)
{
    int rc = ls_check(points, tours, num_tours, n, tours_per_instance);
    if (rc) return rc;
    if (op != 0 && op != 1) return fail(TSP2OPT_E_ARG, "op must be 0 (two-opt) or 1 (relocate)");
    if ((guide_k == nullptr) != (penalty == nullptr)) return fail(TSP2OPT_E_ARG, "guide_k and penalty go together");
    if (num_tours == 0) return 0;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    LsParams p{};
    p.points = points; p.tours = tours; p.delta_out = delta_out; p.status = status_out;
    p.fixed_i = fixed_i; p.guide_k = guide_k; p.penalty = penalty;
    p.num_tours = num_tours; p.n = n; p.tpi = tours_per_instance; p.op = op; p.max_rounds = 1;
    return ls_launch(p, device, static_cast<hipStream_t>(hip_stream));
}

extern "C" int tsp2opt_local_search(const double* points, int64_t* tours, int32_t* rounds_out, int32_t* status_out,
                                    int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t max_rounds,
                                    int32_t device, void* hip_stream)
{
    int rc = ls_check(points, tours, num_tours, n, tours_per_instance);
    if (rc) return rc;
    if (max_rounds < 1) return fail(TSP2OPT_E_ARG, "max_rounds < 1");
    if (num_tours == 0) return 0;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    LsParams p{};
    p.points = points; p.tours = tours; p.rounds_out = rounds_out; p.status = status_out;
    p.num_tours = num_tours; p.n = n; p.tpi = tours_per_instance; p.op = 2; p.max_rounds = max_rounds;
    return ls_launch(p, device, static_cast<hipStream_t>(hip_stream));
}

extern "C" int tsp2opt_gls_rounds(const double* points, int64_t* tours, int64_t* best_tours, double* best_cost,
                                  double* guide_k, double* penalty, const double* regret, int32_t* status_out,
                                  int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t tours_per_map,
                                  int32_t perturbation_moves, int32_t rounds, int32_t init, int32_t device,
                                  void* hip_stream)
{
    int rc = ls_check(points, tours, num_tours, n, tours_per_instance);
    if (rc) return rc;
    if (tours_per_map < 1 || perturbation_moves < 0 || rounds < 0) return fail(TSP2OPT_E_ARG, "bad argument");
    if (num_tours == 0) return 0;
    if (!best_tours || !best_cost || !guide_k || !penalty || !regret) return fail(TSP2OPT_E_ARG, "null pointer");
    if ((reinterpret_cast<uintptr_t>(best_tours) % 8) || (reinterpret_cast<uintptr_t>(best_cost) % 8) ||
        (reinterpret_cast<uintptr_t>(guide_k) % 8) || (reinterpret_cast<uintptr_t>(penalty) % 8) ||
        (reinterpret_cast<uintptr_t>(regret) % 8) || (reinterpret_cast<uintptr_t>(status_out) % 4))
        return fail(TSP2OPT_E_ARG, "misaligned pointer (arrays must be aligned to their element size)");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    GlsParams p{};
    p.points = points; p.tours = tours; p.best_tours = best_tours; p.best_cost = best_cost; p.guide_k = guide_k;
    p.penalty = penalty; p.regret = regret; p.status = status_out; p.num_tours = num_tours; p.n = n;
    p.tpi = tours_per_instance; p.tph = tours_per_map; p.pm = perturbation_moves; p.rounds = rounds;
    p.init = init ? 1 : 0;
    const size_t base = 144 + ((sizeof(uint16_t) * (size_t)(n + 1) + 15) & ~(size_t)15);
    size_t lds = base;
    if (base + 16 * (size_t)n <= 160 * 1024) { p.xlds = 1; lds += 16 * (size_t)n; }
    if (p.xlds && lds + 24 * (size_t)(n + 2) <= 64 * 1024) { p.flat = 1; lds += 24 * (size_t)(n + 2); }
    if (lds > 160 * 1024) return fail(TSP2OPT_E_UNSUPPORTED, "n too large for the local-search operators");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gls), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_gls, dim3((unsigned)num_tours), dim3(256), lds, stream, p);
    HIP_TRY(hipGetLastError());
    return 0;
}
