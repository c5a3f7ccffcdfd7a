// ml4tspbench_amd/csrc/tsp2opt.hip -- gfx950 kernels + C ABI of libtsp2opt_hip.so
//
// Batched best-improvement 2-opt local search, the whole improvement loop on the device.
// Replaces (reference, read-only): search/two_opt.py:21-48 and
// ml4tsp-library/ml4tsp/nar/local_search/two_opt.py:34-65 (+ the per-tour Python loop of
// ml4tsp-library/ml4tsp/nar/local_search/base.py:58-64).  Design: DESIGN.md.
//
// One workgroup owns one tour.  Per iteration it must find the reference's argmin over all
// position pairs (i, j >= i+2) of
//     delta(i,j) = ((D(i,j) + D(i+1,j+1)) - D(i,i+1)) - D(j,j+1),   D = Euclidean distance
// evaluated in the reference's dtype with IEEE-correct +,-,*,sqrt and no FMA, ties broken by the
// lowest flat index i*N+j, then reverse tour[i+1..j].
//
// Hot loop ("filter sweep"): the pair triangle is re-indexed as CYCLIC DIAGONALS (r, (r+d) mod N),
// d = 2..N/2, which is a rectangle (no triangle waste) and along which D(r+1,c+1) of one pair is
// D(r,c) of the next, so each pair costs ONE distance.  A lane owns two groups of 4 adjacent
// diagonals ("streams" A and B, GP groups apart) and walks rows in macro steps of 4 (32 pairs,
// 32 sqrt).  Every value of the step is an (A, B) register pair, so the subtractions, squares and
// sums issue as v_pk_add/v_pk_mul/v_pk_fma_f32 (measured on MI355X: 25 SIMD-cycles per 64 pairs
// against 37 for the scalar form, tools/valu_chain_microbench.hip).  Tour-ordered, centred and
// scaled fp32 coordinates sit in LDS as three 4-way interleaved arrays (x, y, edge length) whose
// entries are already the (A, B) pairs: entry i holds the value of window group i and of group
// i+GP, so one conflict-free ds_read_b64 per operand feeds the packed math with no register moves;
// row operands are broadcast reads of the same arrays.  The fp32 value is only a FILTER with a proven error bound eps: every lane keeps
// its best macro-step a1 and runner-up a2; after a workgroup min m, only macro steps with
// a1 <= m + 2 eps can hold the true argmin and are re-evaluated EXACTLY (reference dtype, reference
// operation order) by wave 0; if some lane's runner-up is also inside the window the workgroup
// falls back to a full exact sweep for that iteration.  The result is bit-identical to evaluating
// every pair exactly, at fp32 VALU cost.
//
// MFMA is deliberately not used: gather + sqrt + compare work, no dense contraction.
//
// Compile with -ffp-contract=off (see build.py); the exact path relies on separate mul and add.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "tsp2opt.h"
#include "host_util.hpp"
#include "wave_reduce.hpp"

#pragma clang fp contract(off)

namespace t2o {

// ------------------------------------------------------------------------------------------
// constants shared by host and device
// ------------------------------------------------------------------------------------------
constexpr int kMaxCand = 64;          // exact re-evaluation list (macro steps) per iteration
constexpr int kMaxWaves = 16;         // 1024 threads
constexpr int kModeIndep = 0;         // run from iteration 0 until the tour's own stop test
constexpr int kModeForced = 1;        // continue to the group's iteration count (coupled mode)
constexpr unsigned kNoMove = 0u;      // packed (i<<16|j) == 0 <=> flat argmin index 0 <=> no move
constexpr unsigned kFullSweep = 0xffffffffu;   // "last move" value that asks for a full filter sweep

struct Geo {                          // launch geometry of the filter sweep (host: make_plan)
    int n;                            // nodes
    int nt;                           // threads per workgroup
    int qclass;                       // 0: exact-only kernel; else sub-array stride Q
    int GP;                           // lane groups; group gp owns cyclic diagonals 2+8gp .. 9+8gp
    int CS, wc;                       // workgroups sharing the tour (1 = none) and this workgroup's index
    int GPl;                          // lane groups walked by this workgroup: gp = wc + CS*gpl, gpl < GPl
    int G;                            // = 2 * GP groups of 4 diagonals (cyclic diagonals 2..4G+1)
    int NS;                           // row segments
    int MS;                           // macro steps (4 rows) per segment, even: MS/2 tiles of 8 rows
    int xlds;                         // exact coordinates cached in LDS
    int W;                            // window entries kept consistent (multiple of 8, <= 8*Q)
    int inc;                          // tile-minimum cache + incremental sweeps
    int lcap;                         // capacity of the changed-tile list
    float eps2;                       // 2*eps of the filter, in scaled units
};

struct Params {
    const void* points;
    int64_t* tours;
    int32_t* iters;                   // INDEP: out.  FORCED: in (own count) / unchanged
    int32_t* work_sweeps;             // sweeps actually executed (accumulated)
    int32_t* status;
    const int32_t* target;            // FORCED: per-group iteration count
    int32_t* group_flag;              // FORCED: set when a tour is still active at the target
    unsigned* queue;                  // tour queue head
    int64_t num_tours;
    int tpi;
    int group_size;
    int max_iter;
    int mode;
    long long time_limit_ticks;       // wall_clock64 ticks, 0 = none
    unsigned long long* stamps;       // optional phase timers (tsp2opt_debug_phase_cycles), else null
    int dbg_sweeps;                   // > 0: diagnostic, run only this many filter sweeps per tour
    unsigned* csync;                  // CS > 1: per cluster `cstride` dwords: [0] arrivals, [1] abort, [16..] slots
    int cstride;
    int num_clusters;
    Geo g;
};

struct Prof {                         // per-thread phase timers, only live when Params::stamps != null
    bool on;
    unsigned long long sweep, select, apply;
    unsigned amb = 0, cands = 0;
    unsigned long long candcyc = 0;
#ifdef TSP2OPT_FINE
    unsigned long long tl = 0, fine[16] = {};
#endif
};

// Diagnostic build only (-DTSP2OPT_FINE, libtsp2opt_hip_fine.so): FINE(pf, k) charges the shader-clock
// cycles since the previous mark to slot k; the slots are summed into workspace stamps [8, 24).
#ifdef TSP2OPT_FINE
#define FINE_START(pf) do { if ((pf).on) (pf).tl = clock64(); } while (0)
#define FINE(pf, k) do { if ((pf).on) { const unsigned long long t_ = clock64(); (pf).fine[k] += t_ - (pf).tl; (pf).tl = t_; } } while (0)
#else
#define FINE_START(pf) do { } while (0)
#define FINE(pf, k) do { } while (0)
#endif

struct Ctl {                          // LDS control block (first bytes of dynamic LDS)
    double redd[kMaxWaves * 4];       // block reductions
    double pE[kMaxWaves];
    unsigned pIdx[kMaxWaves];
    float wmin[kMaxWaves];
    int pAmb[kMaxWaves];
    unsigned cand[kMaxCand];
    unsigned nlist;                   // changed tiles queued for this iteration
    int ambiguous;
    int tour_id;
    int flags;
    double bestE;
    unsigned bestIdx;
    unsigned pad;
    double cx, cy, sc;
};

// ------------------------------------------------------------------------------------------
// exact arithmetic in the reference's dtype (search/two_opt.py:27-32): separate sub, mul, add,
// correctly rounded sqrt.  fp32 sqrt goes through the correctly rounded fp64 sqrt: rounding a
// 53-bit correctly rounded root of a 24-bit value to 24 bits is itself correctly rounded.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double sqrt_rn(double x) { return __builtin_sqrt(x); }
__device__ __forceinline__ float sqrt_rn(float x) { return (float)__builtin_sqrt((double)x); }

template <typename T>
__device__ __forceinline__ T dist_exact(T ax, T ay, T bx, T by)
{
    T dx = ax - bx, dy = ay - by;
    T sx = dx * dx;
    T sy = dy * dy;
    return sqrt_rn(sx + sy);
}

// fp32 filter distance: FMA and the raw 1-ulp v_sqrt_f32 are fine here, the error is budgeted.
__device__ __forceinline__ float dist32(float ax, float ay, float bx, float by)
{
    float dx = ax - bx, dy = ay - by;
    return __builtin_amdgcn_sqrtf(__builtin_fmaf(dy, dy, dx * dx));
}

typedef float v2f __attribute__((ext_vector_type(2)));   // (stream A, stream B)

__device__ __forceinline__ v2f dist32x2(float ax, float ay, v2f bx, v2f by)
{
    const v2f dx = (v2f){ax, ax} - bx, dy = (v2f){ay, ay} - by;
    const v2f s = __builtin_elementwise_fma(dy, dy, dx * dx);
    return (v2f){__builtin_amdgcn_sqrtf(s.x), __builtin_amdgcn_sqrtf(s.y)};
}

// ------------------------------------------------------------------------------------------
// LDS view of one tour
// ------------------------------------------------------------------------------------------
template <typename T>
struct View {
    Ctl* ctl;
    uint16_t* tour;        // n+1 positions
    // filter copy, 8*Q entries each; slot(w) = (w&7)*Q + (w>>3), w = window (cyclic position) index;
    // entry slot(w) = (value at w, value at w + 4): the two streams of a lane in one 8-byte read
    v2f* Xq;               // x
    v2f* Yq;               // y
    v2f* Eq;               // edge lengths e[w] = |P[w] P[w+1]|
    float* Mc;             // filter minimum of every tile (8 rows x 8 diagonals), or null
    uint16_t* list;        // ids of the tiles the last reversal touched
    const T* X;            // exact coordinates, node order (LDS copy or global)
};

template <int Q>
__device__ __forceinline__ int slot(int w) { return (w & 7) * Q + (w >> 3); }

// exact delta of position pair (i, j), i + 2 <= j <= n-1, in the reference's operation order
template <typename T>
__device__ __forceinline__ double exact_delta(const View<T>& v, int i, int j)
{
    const int ti = v.tour[i], ti1 = v.tour[i + 1], tj = v.tour[j], tj1 = v.tour[j + 1];
    const T xi = v.X[2 * ti], yi = v.X[2 * ti + 1];
    const T xi1 = v.X[2 * ti1], yi1 = v.X[2 * ti1 + 1];
    const T xj = v.X[2 * tj], yj = v.X[2 * tj + 1];
    const T xj1 = v.X[2 * tj1], yj1 = v.X[2 * tj1 + 1];
    const T a_ij = dist_exact(xi, yi, xj, yj);
    const T a_i1j1 = dist_exact(xi1, yi1, xj1, yj1);
    const T a_ii1 = dist_exact(xi, yi, xi1, yi1);
    const T a_jj1 = dist_exact(xj, yj, xj1, yj1);
    const T d = ((a_ij + a_i1j1) - a_ii1) - a_jj1;      // two_opt.py:32, left to right
    return (double)d;
}

// Full exact sweep by the whole workgroup (fallback of the filter, and the only sweep of the
// exact-only kernel).  Thread-local scan order is ascending (i, j), so strict '<' keeps the
// lowest flat index; the cross-thread reduction is lexicographic on (delta, index).
template <typename T>
__device__ void exact_sweep(const View<T>& v, int n, int tid, int nt, double& e, unsigned& idx)
{
    e = 0.0;
    idx = kNoMove;
    for (int i = tid; i + 2 <= n - 1; i += nt) {
        const int ti = v.tour[i], ti1 = v.tour[i + 1];
        const T xi = v.X[2 * ti], yi = v.X[2 * ti + 1];
        const T xi1 = v.X[2 * ti1], yi1 = v.X[2 * ti1 + 1];
        const T a_ii1 = dist_exact(xi, yi, xi1, yi1);
        int tj = v.tour[i + 2];
        T xj = v.X[2 * tj], yj = v.X[2 * tj + 1];
        for (int j = i + 2; j <= n - 1; ++j) {
            const int tj1 = v.tour[j + 1];
            const T xj1 = v.X[2 * tj1], yj1 = v.X[2 * tj1 + 1];
            const T a_ij = dist_exact(xi, yi, xj, yj);
            const T a_i1j1 = dist_exact(xi1, yi1, xj1, yj1);
            const T a_jj1 = dist_exact(xj, yj, xj1, yj1);
            const double d = (double)(((a_ij + a_i1j1) - a_ii1) - a_jj1);
            if (d < e) { e = d; idx = ((unsigned)i << 16) | (unsigned)j; }
            xj = xj1; yj = yj1;
        }
    }
}

// workgroup-wide lexicographic min of (e, idx); result valid in every thread
__device__ void block_best(Ctl* ctl, int tid, int nt, double& e, unsigned& idx)
{
    wave_best(e, idx);
    const int nw = nt >> 6;
    if (nw == 1) return;
    if ((tid & 63) == 0) { ctl->pE[tid >> 6] = e; ctl->pIdx[tid >> 6] = idx; }
    __syncthreads();
    e = ctl->pE[0];
    idx = ctl->pIdx[0];
    for (int w = 1; w < nw; ++w) {
        const double oe = ctl->pE[w];
        const unsigned oi = ctl->pIdx[w];
        if (better(oe, oi, e, idx)) { e = oe; idx = oi; }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// filter sweep
// ------------------------------------------------------------------------------------------
// Window geometry of one lane group gp: stream A walks diagonals 2+8gp .. 5+8gp, stream B the next
// four, so the B operand of a column is the A operand 4 positions later = the .y half of the same
// LDS entry.  Rows advance 4 per macro step; two macro steps (row parities PAR = 0, 1 of an 8-row
// block rq) make one TILE of 8 rows x 8 diagonals = 64 pairs, the unit of the minimum cache.
//
// load_block<PAR>(b): per stream C[u] = P[8b + 4 PAR + 2 + u], E[u] = e[8b + 4 PAR + 1 + u]
template <int Q, int PAR>
__device__ __forceinline__ void load_block(const v2f* Xq, const v2f* Yq, const v2f* Eq, int b,
                                           v2f (&CX)[4], v2f (&CY)[4], v2f (&E)[4])
{
    if constexpr (PAR == 0) {
        CX[0] = Xq[2 * Q + b]; CX[1] = Xq[3 * Q + b]; CX[2] = Xq[4 * Q + b]; CX[3] = Xq[5 * Q + b];
        CY[0] = Yq[2 * Q + b]; CY[1] = Yq[3 * Q + b]; CY[2] = Yq[4 * Q + b]; CY[3] = Yq[5 * Q + b];
        E[0] = Eq[Q + b]; E[1] = Eq[2 * Q + b]; E[2] = Eq[3 * Q + b]; E[3] = Eq[4 * Q + b];
    } else {
        CX[0] = Xq[6 * Q + b]; CX[1] = Xq[7 * Q + b]; CX[2] = Xq[b + 1]; CX[3] = Xq[Q + b + 1];
        CY[0] = Yq[6 * Q + b]; CY[1] = Yq[7 * Q + b]; CY[2] = Yq[b + 1]; CY[3] = Yq[Q + b + 1];
        E[0] = Eq[5 * Q + b]; E[1] = Eq[6 * Q + b]; E[2] = Eq[7 * Q + b]; E[3] = Eq[b + 1];
    }
}

// One macro step: rows r0..r0+3, r0 = 8*rq + 4*PAR, x the lane's 2 x 4 diagonals, 32 pairs.
// cb = rq + gp.  Per stream, with c0 = r0 + 2 + 8*gp:
//   Cold[u] = P[c0+u], Cnew[u] = P[c0+4+u], Eold[u] = e[c0-1+u], Enew[u] = e[c0+3+u],
//   Dc[q] = D(r0, c0+q) on entry and D(r0+4, c0+4+q) on exit.
template <int Q, int PAR>
__device__ __forceinline__ float macro_step(const v2f* Xq, const v2f* Yq, const v2f* Eq,
                                            int rq, int cb,
                                            const v2f (&CXo)[4], const v2f (&CYo)[4], v2f (&CXn)[4],
                                            v2f (&CYn)[4], const v2f (&Eo)[4], v2f (&En)[4],
                                            v2f (&Dc)[4])
{
    load_block<Q, 1 - PAR>(Xq, Yq, Eq, cb + PAR, CXn, CYn, En);
    float rx[4], ry[4], er[4];
    if constexpr (PAR == 0) {
        rx[0] = Xq[Q + rq].x;     ry[0] = Yq[Q + rq].x;             // P[r0+1]
        rx[1] = Xq[2 * Q + rq].x; ry[1] = Yq[2 * Q + rq].x;
        rx[2] = Xq[3 * Q + rq].x; ry[2] = Yq[3 * Q + rq].x;
        rx[3] = Xq[4 * Q + rq].x; ry[3] = Yq[4 * Q + rq].x;         // P[r0+4]
        er[0] = Eq[rq].x;                                           // e[r0]
        er[1] = Eq[Q + rq].x;
        er[2] = Eq[2 * Q + rq].x;
        er[3] = Eq[3 * Q + rq].x;
    } else {
        rx[0] = Xq[5 * Q + rq].x; ry[0] = Yq[5 * Q + rq].x;
        rx[1] = Xq[6 * Q + rq].x; ry[1] = Yq[6 * Q + rq].x;
        rx[2] = Xq[7 * Q + rq].x; ry[2] = Yq[7 * Q + rq].x;
        rx[3] = Xq[rq + 1].x;     ry[3] = Yq[rq + 1].x;
        er[0] = Eq[4 * Q + rq].x;
        er[1] = Eq[5 * Q + rq].x;
        er[2] = Eq[6 * Q + rq].x;
        er[3] = Eq[7 * Q + rq].x;
    }
    float mm = INFINITY;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        v2f t[4], Dn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int vp = a + q + 1;                   // column point c0 + vp
            Dn[q] = dist32x2(rx[a], ry[a], vp < 4 ? CXo[vp] : CXn[vp - 4],
                             vp < 4 ? CYo[vp] : CYn[vp - 4]);       // D(r0+a+1, c0+a+q+1)
            const int ve = a + q;                       // column edge c0 + ve
            t[q] = (Dc[q] + Dn[q]) - (ve < 3 ? Eo[ve + 1] : En[ve - 3]);
        }
        float m = __builtin_fminf(__builtin_fminf(t[0].x, t[0].y), t[1].x);
        m = __builtin_fminf(__builtin_fminf(m, t[1].y), t[2].x);
        m = __builtin_fminf(__builtin_fminf(m, t[2].y), t[3].x);
        m = __builtin_fminf(m, t[3].y) - er[a];
        mm = __builtin_fminf(mm, m);
#pragma unroll
        for (int q = 0; q < 4; ++q) Dc[q] = Dn[q];
    }
    return mm;
}

__device__ __forceinline__ void track(float mm, unsigned id, float& a1, float& a2, unsigned& chunk)
{
    a2 = __builtin_amdgcn_fmed3f(a1, mm, a2);   // runner-up (a1 <= a2 invariant)
    chunk = mm < a1 ? id : chunk;
    a1 = fminf(a1, mm);
}

// Tile ids: id = (rq << 8) | gpl, rq = 8-row block of the whole window (segment s of unit
// u = s * GPl + gpl covers rq = s*MS/2 .. (s+1)*MS/2 - 1), gpl = this workgroup's lane group (< 256,
// global lane group gp = wc + CS * gpl); the cached minimum of the tile sits at Mc[rq * GPl + gpl].
// Full sweep: every lane walks its own unit; the tile minima go to the cache when there is one.
template <int Q, bool CL>
__device__ __forceinline__ void filter_sweep(const v2f* Xq, const v2f* Yq, const v2f* Eq, float* Mc,
                                             const Geo& g, int tid, float& a1, float& a2,
                                             unsigned& chunk)
{
    a1 = INFINITY;
    a2 = INFINITY;
    chunk = 0;
    const int units = g.GPl * g.NS, MS2 = g.MS >> 1;
    for (int u = tid; u < units; u += g.nt) {
        const int gpl = u % g.GPl, s = u / g.GPl;
        const int gp = CL ? g.wc + g.CS * gpl : gpl;
        if constexpr (CL) {
            if (gp >= g.GP) continue;                       // padding of the last workgroups
        }
        int rq = s * MS2;
        int cb = rq + gp;
        v2f CX0[4], CY0[4], CX1[4], CY1[4], E0[4], E1[4], Dc[4];
        load_block<Q, 0>(Xq, Yq, Eq, cb, CX0, CY0, E0);
        const float ax = Xq[rq].x, ay = Yq[rq].x;           // P[r0]
#pragma unroll
        for (int q = 0; q < 4; ++q) Dc[q] = dist32x2(ax, ay, CX0[q], CY0[q]);
        unsigned id = ((unsigned)rq << 8) | (unsigned)gpl;
        float* mc = Mc ? Mc + rq * g.GPl + gpl : nullptr;
        for (int m2 = 0; m2 < MS2; ++m2) {
            float mm = macro_step<Q, 0>(Xq, Yq, Eq, rq, cb, CX0, CY0, CX1, CY1, E0, E1, Dc);
            mm = __builtin_fminf(mm, macro_step<Q, 1>(Xq, Yq, Eq, rq, cb, CX1, CY1, CX0, CY0, E1, E0, Dc));
            if (mc) { *mc = mm; mc += g.GPl; }
            track(mm, id, a1, a2, chunk);
            ++rq; ++cb; id += 256u;
        }
    }
}

// filter minimum of one tile, from scratch (incremental sweeps)
template <int Q>
__device__ __forceinline__ float tile_min(const v2f* Xq, const v2f* Yq, const v2f* Eq, int rq, int gp)
{
    const int cb = rq + gp;
    v2f CX0[4], CY0[4], CX1[4], CY1[4], E0[4], E1[4], Dc[4];
    load_block<Q, 0>(Xq, Yq, Eq, cb, CX0, CY0, E0);
    const float ax = Xq[rq].x, ay = Yq[rq].x;
#pragma unroll
    for (int q = 0; q < 4; ++q) Dc[q] = dist32x2(ax, ay, CX0[q], CY0[q]);
    float mm = macro_step<Q, 0>(Xq, Yq, Eq, rq, cb, CX0, CY0, CX1, CY1, E0, E1, Dc);
    return __builtin_fminf(mm, macro_step<Q, 1>(Xq, Yq, Eq, rq, cb, CX1, CY1, CX0, CY0, E1, E0, Dc));
}

// bits [a, b] of a 64-bit mask, clipped to [0, M-1]
__device__ __forceinline__ unsigned long long bit_range(int a, int b, int M)
{
    a = a < 0 ? 0 : a;
    b = b > M - 1 ? M - 1 : b;
    if (a > b) return 0ull;
    return (~0ull >> (63 - b)) & (~0ull << a);
}

struct Unit {               // the lane's own unit, fixed for the whole kernel
    int gp;                 // lane group (of the whole tour), or -1 when the lane owns no unit
    int gpl;                // its index among this workgroup's lane groups
    int rq0;                // first 8-row block of its row segment
};

// Filter sweep of one iteration.  `last` = the reversal applied since the previous sweep (packed
// (i, j)) or kFullSweep.  With the tile cache, only tiles that read a changed window position
// (coordinates i+1..j, edges i..j, any cyclic copy) are recomputed: every lane marks its own
// touched tiles in closed form, the ids are compacted into a list (ballot per row block, one LDS
// atomic per wave), the list is processed evenly by all lanes, then every lane reduces its own
// cached tile minima exactly as the full sweep would.
template <int Q, bool CL, typename T>
__device__ __forceinline__ void sweep(const View<T>& v, const Geo& g, int tid, const Unit& un,
                                      unsigned last, float& a1, float& a2, unsigned& chunk, Prof& pf)
{
    Ctl* ctl = v.ctl;
    bool full = !g.inc || last == kFullSweep;
    FINE_START(pf);
    if (!full) {
        const int MS2 = g.MS >> 1;
        const int i = (int)(last >> 16), j = (int)(last & 0xffffu);
        unsigned long long mask = 0ull;
        if (un.gp >= 0) {
            for (int lo = i, hi = j; lo < g.W; lo += g.n, hi += g.n) {
                mask |= bit_range((lo >> 3) - un.rq0, (hi >> 3) - un.rq0, MS2);                              // rows
                mask |= bit_range(((lo - 9) >> 3) - un.gp - un.rq0, ((hi - 2) >> 3) - un.gp - un.rq0, MS2);  // columns
            }
        }
        // compaction: wave total -> one atomic -> per-lane positions by ballot rank
        unsigned wtotal = 0;
        for (int m2 = 0; m2 < MS2; ++m2) wtotal += (unsigned)__popcll(__ballot((mask >> m2) & 1ull));
        unsigned base = 0;
        if (wtotal) {
            if ((tid & 63) == 0) base = atomicAdd(&ctl->nlist, wtotal);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            for (int m2 = 0; m2 < MS2; ++m2) {
                const unsigned long long b = __ballot((mask >> m2) & 1ull);
                if ((mask >> m2) & 1ull) {
                    const unsigned pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
                    if (pos < (unsigned)g.lcap)
                        v.list[pos] = (uint16_t)(((unsigned)(un.rq0 + m2) << 8) | (unsigned)un.gpl);
                }
                base += (unsigned)__popcll(b);
            }
        }
        FINE(pf, 0);                                        // marking + compaction
        __syncthreads();
        const unsigned nl = ctl->nlist;
        if (nl > (unsigned)g.lcap) {
            full = true;                                    // most of the tour moved: plain sweep
        } else {
            __builtin_amdgcn_s_setprio(0);
            for (unsigned t = (unsigned)tid; t < nl; t += (unsigned)g.nt) {
                const unsigned id = v.list[t];
                const int rq = (int)(id >> 8), gpl = (int)(id & 255u);
                v.Mc[rq * g.GPl + gpl] = tile_min<Q>(v.Xq, v.Yq, v.Eq, rq, g.wc + g.CS * gpl);
            }
            __builtin_amdgcn_s_setprio(3);
        }
        FINE(pf, 1);                                        // barrier + changed tiles
        __syncthreads();
        if (tid == 0) ctl->nlist = 0;
        if (!full) {
            a1 = INFINITY;
            a2 = INFINITY;
            chunk = 0;
            if (un.gp >= 0) {
                unsigned id = ((unsigned)un.rq0 << 8) | (unsigned)un.gpl;
                const float* mc = v.Mc + un.rq0 * g.GPl + un.gpl;
                for (int m2 = 0; m2 < MS2; ++m2) {
                    track(*mc, id, a1, a2, chunk);
                    id += 256u;
                    mc += g.GPl;
                }
            }
            FINE(pf, 2);                                    // barrier + reduction of the cached minima
            return;
        }
    }
    __builtin_amdgcn_s_setprio(0);
    filter_sweep<Q, CL>(v.Xq, v.Yq, v.Eq, g.inc ? v.Mc : nullptr, g, tid, a1, a2, chunk);
    __builtin_amdgcn_s_setprio(3);
    FINE(pf, 3);                                            // full sweep
}

// window position -> tour position; window positions are below 3n (make_plan guarantees W < 3n)
__device__ __forceinline__ int wrap3(int w, int n)
{
    w -= w >= n ? n : 0;
    w -= w >= n ? n : 0;
    return w;
}

// window positions (row, column) of pair `pr` (0..63: row 0..7, diagonal 0..7) of tile `id`
__device__ __forceinline__ void tile_pair(const Geo& g, unsigned id, int pr, int& rw, int& cw)
{
    rw = 8 * (int)(id >> 8) + (pr >> 3);
    cw = rw + 2 + 8 * (g.wc + g.CS * (int)(id & 255u)) + (pr & 7);
}

// exact value of pair `pr` of tile `id`; returns packed index
template <typename T>
__device__ __forceinline__ void eval_chunk_pair(const View<T>& v, const Geo& g, unsigned id, int pr,
                                                double& e, unsigned& idx)
{
    int rw, cw;
    tile_pair(g, id, pr, rw, cw);
    const int r = wrap3(rw, g.n), c = wrap3(cw, g.n);
    const int i = r < c ? r : c, j = r < c ? c : r;
    e = exact_delta(v, i, j);
    idx = ((unsigned)i << 16) | (unsigned)j;
}

// fp32 (filter) value of pair `pr` of tile `id`, recomputed from the LDS window exactly as the
// sweep computes it, with the pair's packed index
template <int Q, typename T>
__device__ __forceinline__ void filter_chunk_pair(const View<T>& v, const Geo& g, unsigned id, int pr,
                                                  float& d, unsigned& idx)
{
    int rw, cw;
    tile_pair(g, id, pr, rw, cw);
    const int sr = slot<Q>(rw), sr1 = slot<Q>(rw + 1), sc = slot<Q>(cw), sc1 = slot<Q>(cw + 1);
    const float d0 = dist32(v.Xq[sr].x, v.Yq[sr].x, v.Xq[sc].x, v.Yq[sc].x);
    const float d1 = dist32(v.Xq[sr1].x, v.Yq[sr1].x, v.Xq[sc1].x, v.Yq[sc1].x);
    d = ((d0 + d1) - v.Eq[sc].x) - v.Eq[sr].x;
    const int r = wrap3(rw, g.n), c = wrap3(cw, g.n);
    const int i = r < c ? r : c, j = r < c ? c : r;
    idx = ((unsigned)i << 16) | (unsigned)j;
}

// One full argmin (the reference's per-tour torch.min / torch.argmin, two_opt.py:35-38).
// Returns in every thread: e = min(0, min valid delta), idx = packed (i,j) or kNoMove.
// `need_exact_sign`: the caller needs the exact sign of a non-improving minimum (coupled-mode
// continuation); otherwise a minimum proven >= -1e-6 by the filter alone ends the search.
// CL (several workgroups per tour): this workgroup's tiles only, always exact (the result is compared
// with the other workgroups' results), no fp32-only shortcut.
// `spd` (thread 0 only): cached filter value of the pair (0, n-1); its operands P[1], P[n-1], e[0],
// e[n-1] change only when the last reversal started at position 1 or ended at position n-1.
template <typename T, int Q, bool CL>
__device__ void find_best(const View<T>& v, const Geo& g, int tid, const Unit& un, bool need_exact_sign,
                          unsigned last, float& spd, double& e, unsigned& idx, Prof& pf)
{
    Ctl* ctl = v.ctl;
    unsigned long long t0 = 0;
    if (pf.on) t0 = clock64();
    if constexpr (Q == 0) {
        exact_sweep(v, g.n, tid, g.nt, e, idx);
        block_best(ctl, tid, g.nt, e, idx);
        return;
    } else {
        float a1, a2;
        unsigned chunk;
        // everything outside the sweep arithmetic is a latency chain that idles most of the
        // workgroup: it runs at high issue priority (set inside sweep()) so that the other
        // workgroups' sweeps on this SIMD do not stretch it
        sweep<Q, CL>(v, g, tid, un, last, a1, a2, chunk, pf);
        if (pf.on) { const unsigned long long t1 = clock64(); pf.sweep += t1 - t0; t0 = t1; }
        const int nw = g.nt >> 6, lane = tid & 63, wv = tid >> 6;
        const float wm = wave_min(a1);
        float m = wm;
        if (nw > 1) {                                       // one-wave workgroups need no LDS round trip
            if (lane == 0) ctl->wmin[wv] = wm;
            __syncthreads();
            m = ctl->wmin[0];
            for (int w = 1; w < nw; ++w) m = fminf(m, ctl->wmin[w]);
        }
        float thr = m + g.eps2;
        if constexpr (CL) {
            // This workgroup sees only its own tiles, whose minimum may be far above the tour's.  Only
            // moves below -1e-6 are ever applied (independent termination), so pairs whose filter value
            // proves delta >= -1e-6 need no exact evaluation: without this clamp a workgroup whose tiles
            // hold no improving move would chase near-ties among its non-improving minima.
            thr = fminf(thr, (float)(-1.0e-6 * ctl->sc) + 0.625f * g.eps2);
        }
        const unsigned long long cmask0 = __ballot(a1 <= thr);
        int amb = __ballot(a2 <= thr) != 0ull || __popcll(cmask0) > 8;
        FINE(pf, 4);                                        // workgroup minimum of the filter

        if constexpr (CL) {
            if (lane == 0) ctl->pAmb[wv] = amb;
            __syncthreads();
            int fl = 0;
            for (int w = 0; w < nw; ++w) fl |= ctl->pAmb[w];
            amb = fl & 1;
        } else {
            // ---- stage 1, fp32 only.  Every pair whose exact delta can be the argmin has a filter
            // value <= m + 2 eps and sits in a candidate macro step.  Re-evaluate those steps pair by
            // pair (lanes 0..31) in fp32: if exactly ONE pair of the whole tour is inside the window it
            // IS the reference's argmin, and if its value is below -1e-6 by more than eps the move is
            // certain without any exact arithmetic.  Likewise a window that lies above -1e-6 + eps
            // proves convergence.  Anything else falls through to the exact stage.
            const float thr_scale = (float)(1.0e-6 * ctl->sc);
            const float eps = 0.625f * g.eps2;                  // eps2 = 2 eps; 25 % extra margin
            float fd = INFINITY;
            unsigned fidx = 0xffffffffu;
            bool multi = false;
            if (!amb) {
                unsigned long long cm = cmask0;
                while (cm) {
                    const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)cm) - 1);
                    cm &= cm - 1;
                    const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)chunk, src);
                    float d;
                    unsigned di;
                    filter_chunk_pair<Q>(v, g, id, lane, d, di);
                    if (d <= thr) {
                        multi = multi || (fidx != 0xffffffffu && fidx != di);
                        if (d < fd || (d == fd && di < fidx)) { fd = d; fidx = di; }
                    }
                }
                if (tid == 0) {     // the pair (0, n-1): cyclically adjacent, not on any walked diagonal
                    const int n = g.n;
                    if (last == kFullSweep || (last >> 16) == 0u || (last & 0xffffu) == (unsigned)(n - 1)) {
                        const int s0 = slot<Q>(0), s1 = slot<Q>(1), sn1 = slot<Q>(n - 1), sn = slot<Q>(n);
                        const float d0 = dist32(v.Xq[s0].x, v.Yq[s0].x, v.Xq[sn1].x, v.Yq[sn1].x);
                        const float d1 = dist32(v.Xq[s1].x, v.Yq[s1].x, v.Xq[sn].x, v.Yq[sn].x);
                        spd = ((d0 + d1) - v.Eq[sn1].x) - v.Eq[s0].x;
                    }
                    const float d = spd;
                    const unsigned di = (unsigned)(n - 1);
                    if (d <= thr) {
                        multi = multi || (fidx != 0xffffffffu && fidx != di);
                        if (d < fd || (d == fd && di < fidx)) { fd = d; fidx = di; }
                    }
                }
            }
            FINE(pf, 5);                                    // fp32 re-evaluation of the candidate tiles
            // wave result: the minimum value and, when every in-window pair of the wave is the same
            // pair, that pair (otherwise the index is not used: `wother` sends the search to stage 2)
            const unsigned long long hm = __ballot(fidx != 0xffffffffu);
            double wd = (double)wave_min(fd);
            unsigned widx = 0xffffffffu;
            if (hm) widx = (unsigned)__builtin_amdgcn_readlane((int)fidx, __ffsll((long long)hm) - 1);
            const bool wother = __ballot(multi || (fidx != 0xffffffffu && fidx != widx)) != 0ull;
            double bd = wd;
            unsigned bi = widx;
            int fl = amb | (wother ? 2 : 0);
            bool unique = true;
            if (nw > 1) {
                if (lane == 0) {
                    ctl->pE[wv] = wd;
                    ctl->pIdx[wv] = widx;
                    ctl->pAmb[wv] = fl;
                }
                __syncthreads();
                bd = ctl->pE[0];
                bi = ctl->pIdx[0];
                fl = ctl->pAmb[0];
            }
            for (int w = 1; w < nw; ++w) {
                const double od = ctl->pE[w];
                const unsigned oi = ctl->pIdx[w];
                fl |= ctl->pAmb[w];
                if (oi != 0xffffffffu && bi != 0xffffffffu && oi != bi) unique = false;
                if (better(od, oi, bd, bi)) { bd = od; bi = oi; }
            }
            amb = fl & 1;
            unique = unique && !(fl & 2) && bi != 0xffffffffu;
            FINE(pf, 6);                                    // wave + workgroup reduction of the fp32 stage
            if (!amb) {
                if (unique && (float)bd < -thr_scale - eps) {           // certain improving move
                    e = ((double)(float)bd + (double)eps) / ctl->sc;     // an upper bound, < -1e-6
                    idx = bi;
                    if (pf.on) pf.select += clock64() - t0;
                    return;
                }
                if (!need_exact_sign && m > -thr_scale + eps && (bi == 0xffffffffu || (float)bd > -thr_scale + eps)) {
                    e = 0.0;                                            // proven: no delta below -1e-6
                    idx = kNoMove;
                    if (pf.on) pf.select += clock64() - t0;
                    return;
                }
            }
        }
        __syncthreads();        // partials are rewritten below
        if (pf.on) { pf.cands += 1; pf.amb += amb ? 1 : 0; }

        // ---- stage 2, exact: each wave re-evaluates its candidate macro steps in the reference's
        // dtype and operation order (lanes 0..31 = the 32 pairs)
        e = 0.0;
        idx = kNoMove;
        if (!amb) {
            unsigned long long cm = cmask0;
            while (cm) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)cm) - 1);
                cm &= cm - 1;
                const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)chunk, src);
                double d;
                unsigned di;
                eval_chunk_pair(v, g, id, lane, d, di);
                if (better(d, di, e, idx)) { e = d; idx = di; }
            }
            if (tid == 0) {
                const double d = exact_delta(v, 0, g.n - 1);
                if (better(d, (unsigned)(g.n - 1), e, idx)) { e = d; idx = (unsigned)(g.n - 1); }
            }
            wave_best(e, idx);
        }
        if (lane == 0) { ctl->pE[wv] = e; ctl->pIdx[wv] = idx; }
        __syncthreads();
        e = ctl->pE[0];
        idx = ctl->pIdx[0];
        for (int w = 1; w < nw; ++w) {
            const double oe = ctl->pE[w];
            const unsigned oi = ctl->pIdx[w];
            if (better(oe, oi, e, idx)) { e = oe; idx = oi; }
        }
        if (amb) {
            // A lane holds two tiles inside the window (or a wave more than 8 candidates): the per-lane
            // best/runner-up no longer names every candidate.  With the tile cache, every candidate is
            // still known -- all tiles whose cached minimum is inside the window -- and is evaluated
            // exactly, one tile per wave at a time; only without the cache (or if even that list
            // overflows: heavy ties, e.g. grids) the whole tour is swept exactly.
            __syncthreads();
            bool listed = false;
            if (g.inc) {
                const int tiles = g.GPl * g.NS * (g.MS >> 1);
                for (int t = tid; t < tiles; t += g.nt) {
                    const int gpl = t % g.GPl;
                    if (v.Mc[t] <= thr && g.wc + g.CS * gpl < g.GP) {
                        const unsigned pos = atomicAdd(&ctl->nlist, 1u);
                        if (pos < (unsigned)g.lcap) v.list[pos] = (uint16_t)(((unsigned)(t / g.GPl) << 8) | (unsigned)gpl);
                    }
                }
                __syncthreads();
                const unsigned nl = ctl->nlist;
                __syncthreads();
                if (tid == 0) ctl->nlist = 0;
                if (nl <= (unsigned)g.lcap) {
                    listed = true;
                    e = 0.0;
                    idx = kNoMove;
                    for (unsigned c = (unsigned)wv; c < nl; c += (unsigned)nw) {
                        double d;
                        unsigned di;
                        eval_chunk_pair(v, g, v.list[c], lane, d, di);
                        if (better(d, di, e, idx)) { e = d; idx = di; }
                    }
                    if (tid == 0) {
                        const double d = exact_delta(v, 0, g.n - 1);
                        if (better(d, (unsigned)(g.n - 1), e, idx)) { e = d; idx = (unsigned)(g.n - 1); }
                    }
                    block_best(ctl, tid, g.nt, e, idx);
                }
            }
            if (!listed) {
                exact_sweep(v, g.n, tid, g.nt, e, idx);
                block_best(ctl, tid, g.nt, e, idx);
            }
        }
        FINE(pf, 7);                                        // exact stage
        if (pf.on) pf.select += clock64() - t0;
    }
}

// write the filter copy of position p: every window index w = p (mod n) is the A value of entry
// slot(w) and the B value of entry slot(w - 4)
template <int Q, typename T>
__device__ __forceinline__ void put_xy(const View<T>& v, const Geo& g, int p, float x, float y)
{
    if (g.W + 4 <= 2 * g.n) {       // (uniform) at most two window copies: w = p and w = p + n
        const int s0 = slot<Q>(p), w1 = p + g.n;
        v.Xq[s0].x = x;
        v.Yq[s0].x = y;
        if (p >= 4) {
            const int sb = slot<Q>(p - 4);
            v.Xq[sb].y = x;
            v.Yq[sb].y = y;
        }
        if (w1 < g.W + 4) {
            const int sb = slot<Q>(w1 - 4);
            v.Xq[sb].y = x;
            v.Yq[sb].y = y;
            if (w1 < g.W) {
                const int s1 = slot<Q>(w1);
                v.Xq[s1].x = x;
                v.Yq[s1].x = y;
            }
        }
        return;
    }
    for (int w = p; w < g.W + 4; w += g.n) {
        if (w < g.W) {
            const int s = slot<Q>(w);
            v.Xq[s].x = x;
            v.Yq[s].x = y;
        }
        if (w >= 4) {
            const int sb = slot<Q>(w - 4);
            v.Xq[sb].y = x;
            v.Yq[sb].y = y;
        }
    }
}

template <int Q, typename T>
__device__ __forceinline__ void put_e(const View<T>& v, const Geo& g, int p, float e)
{
    if (g.W + 4 <= 2 * g.n) {
        const int w1 = p + g.n;
        v.Eq[slot<Q>(p)].x = e;
        if (p >= 4) v.Eq[slot<Q>(p - 4)].y = e;
        if (w1 < g.W + 4) {
            v.Eq[slot<Q>(w1 - 4)].y = e;
            if (w1 < g.W) v.Eq[slot<Q>(w1)].x = e;
        }
        return;
    }
    for (int w = p; w < g.W + 4; w += g.n) {
        if (w < g.W) v.Eq[slot<Q>(w)].x = e;
        if (w >= 4) v.Eq[slot<Q>(w - 4)].y = e;
    }
}

// reverse tour[i+1..j] (two_opt.py:42) and keep the filter window consistent
template <typename T, int Q>
__device__ void apply_move(const View<T>& v, const Geo& g, int tid, unsigned idx, Prof& pf)
{
    const int i = (int)(idx >> 16), j = (int)(idx & 0xffffu);
    FINE_START(pf);

    const int L = j - i;
    const int h = L >> 1;
    for (int k = tid; k < h; k += g.nt) {
        const int p = i + 1 + k, p2 = j - k;
        const uint16_t ta = v.tour[p], tb = v.tour[p2];
        v.tour[p] = tb;
        v.tour[p2] = ta;
        if constexpr (Q != 0) {
            const int sp = slot<Q>(p), sp2 = slot<Q>(p2);
            const float ax = v.Xq[sp].x, ay = v.Yq[sp].x, bx = v.Xq[sp2].x, by = v.Yq[sp2].x;
            put_xy<Q>(v, g, p, bx, by);
            put_xy<Q>(v, g, p2, ax, ay);
            if (k == 0) {   // p = i+1, p2 = j: the two new boundary edges
                put_e<Q>(v, g, i, dist32(v.Xq[slot<Q>(i)].x, v.Yq[slot<Q>(i)].x, bx, by));
                put_e<Q>(v, g, j, dist32(ax, ay, v.Xq[slot<Q>(j + 1)].x, v.Yq[slot<Q>(j + 1)].x));
            }
        }
    }
    FINE(pf, 8);                                // positions + coordinates + boundary edges
    if constexpr (Q != 0) {
        const int he = (L - 1) >> 1;            // interior edges i+1..j-1 are mirrored
        for (int k = tid; k < he; k += g.nt) {
            const int p = i + 1 + k, p2 = j - 1 - k;
            const float a = v.Eq[slot<Q>(p)].x, b = v.Eq[slot<Q>(p2)].x;
            put_e<Q>(v, g, p, b);
            put_e<Q>(v, g, p2, a);
        }
    }
    FINE(pf, 9);                                // interior edges
    __syncthreads();
    FINE(pf, 10);                               // closing barrier
}

// ------------------------------------------------------------------------------------------
// cluster barrier + best-move exchange (several workgroups per tour, both cluster kernels)
// ------------------------------------------------------------------------------------------
constexpr unsigned kSpinLimit = 1u << 26;       // bounded wait (about 10 s): never hang the GPU
constexpr int kSlotDw = 4;                      // delta lo, delta hi, index, flags

// publish this workgroup's (e, idx), wait for the cluster, reduce all slots.  Returns false on
// abort (a workgroup of the cluster is missing: spin limit hit).
__device__ bool cluster_exchange(Ctl* ctl, unsigned* cs, int CS, int wc, unsigned epoch, int tid,
                                 int nt, unsigned flags_in, double& e, unsigned& idx,
                                 unsigned& flags_out)
{
    if (tid == 0) {
        unsigned* slot = cs + 16 + ((epoch & 1u) * (unsigned)CS + (unsigned)wc) * kSlotDw;
        __hip_atomic_store(slot + 0, (unsigned)__double2loint(e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 1, (unsigned)__double2hiint(e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 2, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 3, flags_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(cs, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)CS * (epoch + 1u);
        unsigned spins = 0;
        bool ok = true;
        while (__hip_atomic_load(cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit ||
                __hip_atomic_load(cs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(cs + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ctl->ambiguous = ok ? 0 : 1;        // reused as the abort flag
    }
    __syncthreads();
    const bool ok = ctl->ambiguous == 0;
    // slot w is read by thread w (agent-scope loads: they bypass this CU's L1), then reduced
    e = 0.0;
    idx = kNoMove;
    unsigned fl = 0;
    if (ok && tid < CS) {
        const unsigned* sl = cs + 16 + ((epoch & 1u) * (unsigned)CS + (unsigned)tid) * kSlotDw;
        const unsigned lo = __hip_atomic_load(sl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned hi = __hip_atomic_load(sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx = __hip_atomic_load(sl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fl = __hip_atomic_load(sl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        e = __hiloint2double((int)hi, (int)lo);
    }
    wave_best(e, idx);
    const int anyfl = __ballot(fl != 0u) != 0ull;
    const int nw = nt >> 6, wv = tid >> 6;
    if ((tid & 63) == 0) { ctl->pE[wv] = e; ctl->pIdx[wv] = idx; ctl->pAmb[wv] = anyfl; }
    __syncthreads();
    e = ctl->pE[0];
    idx = ctl->pIdx[0];
    int f = ctl->pAmb[0];
    const int nwu = (CS + 63) >> 6 < nw ? (CS + 63) >> 6 : nw;      // waves that held slots
    for (int w = 1; w < nwu; ++w) {
        const double oe = ctl->pE[w];
        const unsigned oi = ctl->pIdx[w];
        f |= ctl->pAmb[w];
        if (better(oe, oi, e, idx)) { e = oe; idx = oi; }
    }
    flags_out = f ? 1u : 0u;
    __syncthreads();
    return ok;
}

constexpr unsigned kFlagStop = 1u;      // published by workgroup 0: wall-clock limit reached

// ------------------------------------------------------------------------------------------
// the persistent kernel
// ------------------------------------------------------------------------------------------
__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// CL = several workgroups per tour ("full-window cluster", N up to the fast path's LDS limit): every
// workgroup of a cluster holds the whole tour and the whole filter window and applies every reversal,
// but sweeps only the lane groups gp = wc (mod CS) -- interleaved, so all workgroups see the same mix
// of short and long diagonals and the same share of the tiles a reversal touches.  Per iteration each
// finds the exact best move of its tiles, the cluster meets at cluster_exchange and everybody applies
// the winner.  Clusters take tours statically (cluster, cluster + num_clusters, ...); the host
// launches at most one workgroup per CU so that a cluster is co-resident.  Independent termination only.
template <typename T, int Q, bool CL>
__global__ void __launch_bounds__(1024) k_two_opt(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Geo g = p.g;
    int64_t tk_next = 0;
    unsigned* cs = nullptr;
    unsigned epoch = 0;                         // barriers passed by this cluster so far
    if constexpr (CL) {
        g.wc = (int)(blockIdx.x % (unsigned)g.CS);
        tk_next = blockIdx.x / (unsigned)g.CS;
        cs = p.csync + (size_t)tk_next * (size_t)p.cstride;
    } else {                                    // compile-time constants for the one-workgroup kernel
        g.CS = 1;
        g.wc = 0;
        g.GPl = g.GP;
    }
    const int n = g.n, tid = threadIdx.x, nt = g.nt;
    View<T> v;
    v.ctl = reinterpret_cast<Ctl*>(smem);
    size_t off = align16(sizeof(Ctl));
    v.tour = reinterpret_cast<uint16_t*>(smem + off);
    off += align16(sizeof(uint16_t) * (size_t)(n + 1));
    v.Xq = reinterpret_cast<v2f*>(smem + off);
    off += sizeof(v2f) * 8 * (size_t)Q;
    v.Yq = reinterpret_cast<v2f*>(smem + off);
    off += sizeof(v2f) * 8 * (size_t)Q;
    v.Eq = reinterpret_cast<v2f*>(smem + off);
    off += sizeof(v2f) * 8 * (size_t)Q;
    v.Mc = nullptr;
    v.list = nullptr;
    if (g.inc) {
        const size_t tiles = (size_t)g.GPl * g.NS * (g.MS >> 1);
        v.Mc = reinterpret_cast<float*>(smem + off);
        off += align16(sizeof(float) * tiles);
        v.list = reinterpret_cast<uint16_t*>(smem + off);
        off += align16(sizeof(uint16_t) * (size_t)g.lcap);
    }
    T* xl = reinterpret_cast<T*>(smem + off);
    Ctl* ctl = v.ctl;
    Unit un{-1, 0, 0};
    if constexpr (Q != 0) {
        if (tid < g.GPl * g.NS) {
            un.gpl = tid % g.GPl;
            un.rq0 = (tid / g.GPl) * (g.MS >> 1);
            const int gp = CL ? g.wc + g.CS * un.gpl : un.gpl;
            if (!CL || gp < g.GP) un.gp = gp;
        }
    }
    for (;;) {
        __syncthreads();
        int64_t tk;
        if constexpr (CL) {
            if (tid == 0) ctl->nlist = 0;
            tk = tk_next;
            tk_next += p.num_clusters;
        } else {
            if (tid == 0) { ctl->tour_id = (int)atomicAdd(p.queue, 1u); ctl->nlist = 0; }
            __syncthreads();
            tk = ctl->tour_id;
        }
        if (tk >= p.num_tours) break;          // uniform exit: every wave leaves here

        const int64_t grp = tk / p.group_size;
        Prof pf{p.stamps != nullptr, 0, 0, 0};
        const unsigned long long t_tour = pf.on ? clock64() : 0;
        int it = 0, target = 0;
        if (p.mode == kModeForced) {
            it = p.iters[tk];
            target = p.target[grp];
            const int st = p.status ? p.status[tk] : 0;
            if (it >= target || (st & (TSP2OPT_ST_BAD_INDEX | TSP2OPT_ST_NOT_CLOSED |
                                       TSP2OPT_ST_NONFINITE | TSP2OPT_ST_RANGE)))
                continue;
        }
        const T* X = reinterpret_cast<const T*>(p.points) + (size_t)(tk / p.tpi) * (size_t)n * 2;
        int64_t* gt = p.tours + (size_t)tk * (size_t)(n + 1);

        // ---- stage the tour, validate it (no out-of-range index may reach a gather) --------
        int bad = 0;
        for (int k = tid; k <= n; k += nt) {
            const int64_t t = gt[k];
            if (t < 0 || t >= n) { bad |= TSP2OPT_ST_BAD_INDEX; v.tour[k] = 0; }
            else v.tour[k] = (uint16_t)t;
        }
        if (tid == 0 && gt[0] != gt[n]) bad |= TSP2OPT_ST_NOT_CLOSED;
        // ---- bounding box of the instance (for centring/scaling the filter copy) -----------
        double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
        for (int k = tid; k < n; k += nt) {
            const T x = X[2 * k], y = X[2 * k + 1];
            if (g.xlds) { xl[2 * k] = x; xl[2 * k + 1] = y; }
            const double dx = (double)x, dy = (double)y;
            if (!(fabs(dx) <= 1.0e150) || !(fabs(dy) <= 1.0e150))
                bad |= (dx != dx || dy != dy || fabs(dx) == INFINITY || fabs(dy) == INFINITY)
                           ? TSP2OPT_ST_NONFINITE : TSP2OPT_ST_RANGE;
            mnx = fmin(mnx, dx); mxx = fmax(mxx, dx);
            mny = fmin(mny, dy); mxy = fmax(mxy, dy);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mnx = fmin(mnx, __shfl_xor(mnx, o)); mxx = fmax(mxx, __shfl_xor(mxx, o));
            mny = fmin(mny, __shfl_xor(mny, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
            bad |= __shfl_xor(bad, o);
        }
        if ((tid & 63) == 0) {
            const int w = tid >> 6;
            ctl->redd[4 * w] = mnx; ctl->redd[4 * w + 1] = mxx;
            ctl->redd[4 * w + 2] = mny; ctl->redd[4 * w + 3] = mxy;
            ctl->pIdx[w] = (unsigned)bad;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < (nt >> 6); ++w) {
                mnx = fmin(mnx, ctl->redd[4 * w]); mxx = fmax(mxx, ctl->redd[4 * w + 1]);
                mny = fmin(mny, ctl->redd[4 * w + 2]); mxy = fmax(mxy, ctl->redd[4 * w + 3]);
                bad |= (int)ctl->pIdx[w];
            }
            ctl->flags = bad;
            ctl->cx = 0.5 * (mnx + mxx);
            ctl->cy = 0.5 * (mny + mxy);
            const double half = fmax(0.5 * (mxx - mnx), 0.5 * (mxy - mny));
            int ex = 0;
            double sc = 1.0;
            if (half > 0.0 && half < INFINITY) { (void)frexp(half, &ex); sc = ldexp(1.0, -ex); }
            ctl->sc = sc;                       // half * sc in [0.5, 1)
        }
        __syncthreads();
        const int flags = ctl->flags;
        v.X = g.xlds ? xl : X;
        if (flags != 0) {                       // invalid input: tour untouched (uniform branch)
            if (tid == 0 && g.wc == 0) {
                if (p.status) p.status[tk] = flags;
                if (p.mode == kModeIndep) p.iters[tk] = 0;
            }
            continue;
        }
        if constexpr (Q != 0) {
            const double cx = ctl->cx, cy = ctl->cy, sc = ctl->sc;
            for (int w = tid; w < 8 * Q; w += nt) {
                const int na = v.tour[w % n], nb = v.tour[(w + 4) % n];
                const double xa = (double)v.X[2 * na], ya = (double)v.X[2 * na + 1];
                const double xb = (double)v.X[2 * nb], yb = (double)v.X[2 * nb + 1];
                v.Xq[slot<Q>(w)] = (v2f){(float)((xa - cx) * sc), (float)((xb - cx) * sc)};
                v.Yq[slot<Q>(w)] = (v2f){(float)((ya - cy) * sc), (float)((yb - cy) * sc)};
            }
            __syncthreads();
            for (int w = tid; w < 8 * Q; w += nt) {
                const int s0 = slot<Q>(w), s1 = slot<Q>(w + 1 < 8 * Q ? w + 1 : w);
                const v2f x0 = v.Xq[s0], y0 = v.Yq[s0], x1 = v.Xq[s1], y1 = v.Yq[s1];
                v.Eq[s0] = (v2f){dist32(x0.x, y0.x, x1.x, y1.x), dist32(x0.y, y0.y, x1.y, y1.y)};
            }
        }
        __syncthreads();

        // ---- improvement loop ----------------------------------------------------------------
        if (pf.on && tid == 0) atomicAdd(&p.stamps[3], clock64() - t_tour);
        const long long t_start = p.time_limit_ticks ? wall_clock64() : 0;
        int sweeps = 0, st = 0;
        double e;
        unsigned idx;
        if constexpr (Q != 0) {
            if (p.dbg_sweeps > 0) {     // diagnostic: pure filter-sweep throughput, no selection/reversal
                float acc = 0.f;
                for (int k = 0; k < p.dbg_sweeps; ++k) {
                    float a1, a2;
                    unsigned chunk;
                    filter_sweep<Q, CL>(v.Xq, v.Yq, v.Eq, v.Mc, g, tid, a1, a2, chunk);
                    acc += a1 + a2 + (float)chunk;
                    asm volatile("" ::: "memory");
                }
                if (acc == 12345.678f) p.iters[tk] = 1;      // keeps the sweeps live
                if (tid == 0) { p.iters[tk] = p.dbg_sweeps; if (p.status) p.status[tk] = 0; }
                continue;
            }
        }
        // One loop for both modes (a single inlined copy of the sweep keeps the hot code small).
        //   independent: stop on the tour's own minimum (two_opt.py:40) or the cap (:47).
        //   forced (coupled mode, continuation): the group keeps iterating until `target`; this tour
        //   applies its own strictly negative argmin every iteration (two_opt.py:41-42).  A fixed
        //   point (no negative delta) or a proven 2-cycle (the same reversal twice in a row) fixes
        //   the rest of the trajectory without sweeping; otherwise one more sweep at `target` tells
        //   whether the tour is still active there (then the group is not finished: k_group_lockstep).
        const bool forced = p.mode == kModeForced;
        bool have_prev = false, prev_active = false, active_at_target = false;
        unsigned prev_idx = kNoMove;
        unsigned last = kFullSweep;             // reversal applied since the previous sweep
        float spd = 0.f;                        // thread 0: cached filter value of the pair (0, n-1)
        for (;;) {
            const bool check_only = forced && it >= target;
            if (check_only && target >= p.max_iter) break;
            find_best<T, Q, CL>(v, g, tid, un, forced, last, spd, e, idx, pf);
            ++sweeps;
            unsigned xfl = 0;
            if constexpr (CL) {
                unsigned myflags = 0;
                if (g.wc == 0 && p.time_limit_ticks && wall_clock64() - t_start > p.time_limit_ticks)
                    myflags |= kFlagStop;
                if (!cluster_exchange(ctl, cs, g.CS, g.wc, epoch, tid, nt, myflags, e, idx, xfl)) {
                    st = TSP2OPT_ST_INTERNAL;
                    break;
                }
                ++epoch;
            }
            const bool active = e < -1.0e-6;
            bool stop_after = false;
            if (!forced) {
                if (!active) break;                                         // two_opt.py:40
            } else {
                if (check_only) { active_at_target = active; break; }
                if (!(e < 0.0)) { it = target; break; }                     // fixed point: inactive forever
                if (have_prev && idx == prev_idx) {                         // 2-cycle A -> B -> A
                    const int rem = target - it;
                    it = target;
                    if (!(rem & 1)) { active_at_target = active; break; }
                    active_at_target = prev_active;
                    stop_after = true;                                      // one more reversal: back to A
                }
                have_prev = true; prev_idx = idx; prev_active = active;
            }
            unsigned long long ta = 0;
            if (pf.on) ta = clock64();
            apply_move<T, Q>(v, g, tid, idx, pf);
            last = idx;
            if (pf.on) pf.apply += clock64() - ta;
            if (stop_after) break;
            ++it;
            if (!forced) {
                if (it >= p.max_iter) { st |= TSP2OPT_ST_CAPPED; break; }   // two_opt.py:47
                if (CL ? (xfl & kFlagStop) != 0u
                       : (p.time_limit_ticks && wall_clock64() - t_start > p.time_limit_ticks)) {
                    st |= TSP2OPT_ST_TIME;                                  // plugin two_opt.py:63-65
                    break;
                }
            }
        }
        if (forced && active_at_target && target < p.max_iter && tid == 0)
            atomicOr(&p.group_flag[grp], 1);

        // ---- write back ------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(0);
        if (CL && st == TSP2OPT_ST_INTERNAL) {  // cluster broken (a workgroup is missing): tours untouched
            if (g.wc == 0 && tid == 0 && p.status) p.status[tk] = st;
            return;
        }
        if (CL && g.wc != 0) continue;          // every workgroup holds the same tour; the first writes it
        for (int k = tid; k <= n; k += nt) gt[k] = (int64_t)v.tour[k];
        if (tid == 0) {
            p.iters[tk] = it;
            if (p.work_sweeps) p.work_sweeps[tk] += sweeps;
            if (p.status && p.mode == kModeIndep) p.status[tk] = st;
            if (pf.on) {
                atomicAdd(&p.stamps[0], pf.sweep);
                atomicAdd(&p.stamps[1], pf.select);
                atomicAdd(&p.stamps[2], pf.apply);
                atomicAdd(&p.stamps[4], clock64() - t_tour);
                atomicAdd(&p.stamps[5], (unsigned long long)pf.amb);
                atomicAdd(&p.stamps[6], (unsigned long long)pf.cands);
#ifdef TSP2OPT_FINE
                for (int k = 0; k < 16; ++k) atomicAdd(&p.stamps[8 + k], pf.fine[k]);
#endif
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// cluster kernel: several workgroups per tour (few tours, or N beyond one workgroup's LDS)
// ------------------------------------------------------------------------------------------
// The (row segment x diagonal group) rectangle of one tour is tiled over CS = CR x CD workgroups.
// Every workgroup keeps the whole tour (uint16) in LDS and applies every reversal to it, so the
// tour needs no exchange; its filter copy covers only its own tile: the column pair arrays for
// window groups [ilo, ilo+Qw) and plain row arrays for its row band.  Per iteration each
// workgroup finds the exact best move of its tile (same filter + exact re-evaluation as above),
// publishes (delta, index) in a global slot and meets the others at a counter barrier
// (agent-scope release/acquire, write-through payload: guide G16); all of them then reduce the CS
// slots identically and apply the same move.  Entries of the filter copy whose position lies in
// the reversed segment are re-staged from the tour.  Workgroups of a cluster must be co-resident:
// the host launches at most one workgroup per CU, and every spin is bounded.
struct CGeo {
    int n, nt;
    int GP;                 // lane groups of the whole tour: group gp owns cyclic diagonals 2+8gp .. 9+8gp
                            // (stream A the first four, stream B the next four)
    int CS, CR, CD;         // workgroups per tour = CR row bands x CD diagonal-group bands
    int GPw, NSw, MS;       // per workgroup: lane groups, row segments; macro steps per segment
    int Qw;                 // entries per sub-array of the column pair arrays
    int rowsw;              // row positions of the band = 4*NSw*MS
    int xlds;
    int K, Ksh;             // macro steps per cache entry ("step group", power of two) and log2(4*K)
    int NG;                 // step groups per unit = ceil(MS / K), <= 64
    int inc;                // group-minimum cache + incremental sweeps (needs GPw*NSw <= nt)
    int lcap;               // capacity of the changed-group list
    float eps2;
};

struct CParams {
    const void* points;
    int64_t* tours;
    int32_t* iters;
    int32_t* work_sweeps;
    int32_t* status;
    unsigned* csync;        // per cluster `cstride` dwords: [0] arrivals, [1] abort, [16..] slots
    int cstride;
    int64_t num_tours;
    int tpi;
    int max_iter;
    long long time_limit_ticks;
    int num_clusters;
    unsigned long long* stamps;     // optional phase timers (tsp2opt_debug_phase_cycles)
    CGeo g;
};

template <typename T>
struct CView {
    Ctl* ctl;
    uint16_t* tour;
    v2f *Xq, *Yq, *Eq;      // column pair arrays, slot = u*Qw + local index
    float *RX, *RY, *RE;    // row band, natural order
    float* Mc;              // filter minimum of every step group, Mc[group * units + unit], or null
    uint16_t* list;         // ids (unit * NG + group) of the step groups the last reversal touched
    const T* X;
    double cx, cy, sc;
    int ilo;                // window group of local column index 0
    int rbase;              // tour position (unwrapped) of local row 0
    int rs0, gp0;           // first row segment / lane group of this workgroup
};

template <typename T>
__device__ __forceinline__ void scaled_pos(const CView<T>& v, int p, float& x, float& y)
{
    const int node = v.tour[p];
    x = (float)(((double)v.X[2 * node] - v.cx) * v.sc);
    y = (float)(((double)v.X[2 * node + 1] - v.cy) * v.sc);
}

// (re)build the filter entries whose tour position is in [lo, hi] (coordinates) / [elo, ehi]
// (edge lengths); positions are taken modulo n
template <typename T>
__device__ void stage_c(const CView<T>& v, const CGeo& g, int tid, int lo, int hi, int elo, int ehi)
{
    const int n = g.n;
    for (int k = tid; k < 4 * g.Qw; k += g.nt) {
        const int il = k >> 2, u = k & 3, sl = u * g.Qw + il;
        const int w = 4 * (v.ilo + il) + u;
        const int pa = w % n, pb = (w + 4) % n;
        if (pa >= lo && pa <= hi) { float x, y; scaled_pos(v, pa, x, y); v.Xq[sl].x = x; v.Yq[sl].x = y; }
        if (pb >= lo && pb <= hi) { float x, y; scaled_pos(v, pb, x, y); v.Xq[sl].y = x; v.Yq[sl].y = y; }
        if (pa >= elo && pa <= ehi) {
            float x0, y0, x1, y1;
            scaled_pos(v, pa, x0, y0); scaled_pos(v, pa + 1, x1, y1);
            v.Eq[sl].x = dist32(x0, y0, x1, y1);
        }
        if (pb >= elo && pb <= ehi) {
            float x0, y0, x1, y1;
            scaled_pos(v, pb, x0, y0); scaled_pos(v, pb + 1, x1, y1);
            v.Eq[sl].y = dist32(x0, y0, x1, y1);
        }
    }
    for (int k = tid; k < g.rowsw + 8; k += g.nt) {
        const int pp = (v.rbase + k) % n;
        if (pp >= lo && pp <= hi) { float x, y; scaled_pos(v, pp, x, y); v.RX[k] = x; v.RY[k] = y; }
        if (pp >= elo && pp <= ehi) {
            float x0, y0, x1, y1;
            scaled_pos(v, pp, x0, y0); scaled_pos(v, pp + 1, x1, y1);
            v.RE[k] = dist32(x0, y0, x1, y1);
        }
    }
}

__device__ __forceinline__ void load_block_c(const v2f* Xq, const v2f* Yq, const v2f* Eq, int Q, int b,
                                             v2f (&CX)[4], v2f (&CY)[4], v2f (&E)[4])
{
    CX[0] = Xq[2 * Q + b]; CX[1] = Xq[3 * Q + b]; CX[2] = Xq[b + 1]; CX[3] = Xq[Q + b + 1];
    CY[0] = Yq[2 * Q + b]; CY[1] = Yq[3 * Q + b]; CY[2] = Yq[b + 1]; CY[3] = Yq[Q + b + 1];
    E[0] = Eq[Q + b]; E[1] = Eq[2 * Q + b]; E[2] = Eq[3 * Q + b]; E[3] = Eq[b + 1];
}

// same macro step as the single-workgroup kernel, rows from the band arrays (rl = local row of r0)
__device__ __forceinline__ float macro_step_c(const v2f* Xq, const v2f* Yq, const v2f* Eq, int Q,
                                              const float* RX, const float* RY, const float* RE,
                                              int rl, int colIdx, const v2f (&CXo)[4],
                                              const v2f (&CYo)[4], v2f (&CXn)[4], v2f (&CYn)[4],
                                              const v2f (&Eo)[4], v2f (&En)[4], v2f (&Dc)[4])
{
    load_block_c(Xq, Yq, Eq, Q, colIdx, CXn, CYn, En);
    float rx[4], ry[4], er[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { rx[a] = RX[rl + 1 + a]; ry[a] = RY[rl + 1 + a]; er[a] = RE[rl + a]; }
    float mm = INFINITY;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        v2f t[4], Dn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int vp = a + q + 1;
            Dn[q] = dist32x2(rx[a], ry[a], vp < 4 ? CXo[vp] : CXn[vp - 4], vp < 4 ? CYo[vp] : CYn[vp - 4]);
            const int ve = a + q;
            t[q] = (Dc[q] + Dn[q]) - (ve < 3 ? Eo[ve + 1] : En[ve - 3]);
        }
        float m = __builtin_fminf(__builtin_fminf(t[0].x, t[0].y), t[1].x);
        m = __builtin_fminf(__builtin_fminf(m, t[1].y), t[2].x);
        m = __builtin_fminf(__builtin_fminf(m, t[2].y), t[3].x);
        m = __builtin_fminf(m, t[3].y) - er[a];
        mm = __builtin_fminf(mm, m);
#pragma unroll
        for (int q = 0; q < 4; ++q) Dc[q] = Dn[q];
    }
    return mm;
}

// walk `count` consecutive macro steps of one unit starting at local row rl / column index col - 1
// (Dc rebuilt from scratch); f(mm) receives the filter minimum of every macro step in order
template <typename T, typename F>
__device__ __forceinline__ void walk_c(const CView<T>& v, const CGeo& g, int rl, int col, int count, F&& f)
{
    v2f CX0[4], CY0[4], CX1[4], CY1[4], E0[4], E1[4], Dc[4];
    load_block_c(v.Xq, v.Yq, v.Eq, g.Qw, col - 1, CX0, CY0, E0);
    const float ax = v.RX[rl], ay = v.RY[rl];
#pragma unroll
    for (int q = 0; q < 4; ++q) Dc[q] = dist32x2(ax, ay, CX0[q], CY0[q]);
    int m = 0;
    for (; m + 1 < count; m += 2) {
        f(macro_step_c(v.Xq, v.Yq, v.Eq, g.Qw, v.RX, v.RY, v.RE, rl, col, CX0, CY0, CX1, CY1, E0, E1, Dc));
        f(macro_step_c(v.Xq, v.Yq, v.Eq, g.Qw, v.RX, v.RY, v.RE, rl + 4, col + 1, CX1, CY1, CX0, CY0, E1, E0, Dc));
        rl += 8; col += 2;
    }
    if (m < count)
        f(macro_step_c(v.Xq, v.Yq, v.Eq, g.Qw, v.RX, v.RY, v.RE, rl, col, CX0, CY0, CX1, CY1, E0, E1, Dc));
}

// Chunk ids of the cluster kernel: id = unit * NG + group, unit = sl * GPw + gpl (local row segment,
// local lane group), group = K consecutive macro steps of the unit (K = 1 without the cache).
// Full sweep of this workgroup's tile; the group minima go to the cache when there is one.
template <typename T>
__device__ __forceinline__ void filter_sweep_c(const CView<T>& v, const CGeo& g, int tid, float& a1,
                                               float& a2, unsigned& chunk)
{
    a1 = INFINITY;
    a2 = INFINITY;
    chunk = 0;
    const int units = g.GPw * g.NSw;
    for (int u = tid; u < units; u += g.nt) {
        const int gpl = u % g.GPw, sl = u / g.GPw;
        if (v.gp0 + gpl >= g.GP) continue;              // padding of the last diagonal band
        unsigned id = (unsigned)u * (unsigned)g.NG;
        float* mc = v.Mc ? v.Mc + u : nullptr;
        float gm = INFINITY;
        int k = 0;
        const int K = g.K;
        walk_c(v, g, 4 * sl * g.MS, sl * g.MS + 2 * gpl + 1, g.MS, [&](float mm) {
            gm = __builtin_fminf(gm, mm);
            if (++k == K) {
                if (mc) { *mc = gm; mc += units; }
                track(gm, id, a1, a2, chunk);
                gm = INFINITY; k = 0; ++id;
            }
        });
        if (k) {                                        // last, shorter group
            if (mc) *mc = gm;
            track(gm, id, a1, a2, chunk);
        }
    }
}

// filter minimum of one step group, from scratch (incremental sweeps)
template <typename T>
__device__ __forceinline__ float group_min_c(const CView<T>& v, const CGeo& g, int u, int grp)
{
    const int gpl = u % g.GPw, sl = u / g.GPw;
    const int m0 = grp * g.K;
    const int cnt = g.MS - m0 < g.K ? g.MS - m0 : g.K;
    float gm = INFINITY;
    walk_c(v, g, 4 * (sl * g.MS + m0), sl * g.MS + 2 * gpl + 1 + m0, cnt,
           [&](float mm) { gm = __builtin_fminf(gm, mm); });
    return gm;
}

// Filter sweep of one iteration of the cluster kernel.  `last` = the reversal applied since the
// previous sweep or kFullSweep.  With the cache, only step groups that read a changed window
// position (coordinates last.i+1..last.j, edges last.i..last.j, any cyclic copy) are recomputed:
// each lane marks the touched groups of its own unit in closed form (rows, columns), the ids are compacted into a list and processed evenly by all lanes, then
// every lane reduces the cached minima of its unit exactly as the full sweep would.
template <typename T>
__device__ __forceinline__ void sweep_c(const CView<T>& v, const CGeo& g, int tid, unsigned last,
                                        float& a1, float& a2, unsigned& chunk, Prof& pf)
{
    Ctl* ctl = v.ctl;
    const int units = g.GPw * g.NSw;
    bool full = !g.inc || last == kFullSweep;
    FINE_START(pf);
    if (!full) {
        const int i = (int)(last >> 16), j = (int)(last & 0xffffu);
        const int gpl = tid % g.GPw, sl = tid / g.GPw;
        const bool own = tid < units && v.gp0 + gpl < g.GP;
        unsigned long long mask = 0ull;
        if (own) {
            const int rb = v.rbase + 4 * sl * g.MS;                 // window position of the unit's first row
            const int ca = rb + 2 + 8 * (v.gp0 + gpl);              // ... of its first column (stream A)
            const int wmax = ca + (g.NG << g.Ksh) + 8;
            // group q reads rows [rb + Sq, rb + Sq + S], columns [ca + Sq, ca + Sq + S + 7], S = 4K
            for (int lo = i, hi = j; lo <= wmax; lo += g.n, hi += g.n) {
                mask |= bit_range((lo - rb - 1) >> g.Ksh, (hi - rb) >> g.Ksh, g.NG);
                mask |= bit_range((lo - ca - 8) >> g.Ksh, (hi - ca) >> g.Ksh, g.NG);
            }
        }
        unsigned wtotal = 0;
        for (int q = 0; q < g.NG; ++q) wtotal += (unsigned)__popcll(__ballot((mask >> q) & 1ull));
        unsigned base = 0;
        if (wtotal) {
            if ((tid & 63) == 0) base = atomicAdd(&ctl->nlist, wtotal);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            for (int q = 0; q < g.NG; ++q) {
                const unsigned long long b = __ballot((mask >> q) & 1ull);
                if ((mask >> q) & 1ull) {
                    const unsigned pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
                    if (pos < (unsigned)g.lcap) v.list[pos] = (uint16_t)(tid * g.NG + q);
                }
                base += (unsigned)__popcll(b);
            }
        }
        FINE(pf, 0);
        __syncthreads();
        const unsigned nl = ctl->nlist;
        if (nl > (unsigned)g.lcap) {
            full = true;                                    // most of the tile moved: plain sweep
        } else {
            __builtin_amdgcn_s_setprio(0);
            for (unsigned t = (unsigned)tid; t < nl; t += (unsigned)g.nt) {
                const int id = v.list[t];
                const int u = id / g.NG, q = id - u * g.NG;
                v.Mc[q * units + u] = group_min_c(v, g, u, q);
            }
            __builtin_amdgcn_s_setprio(3);
        }
        FINE(pf, 1);
        __syncthreads();
        if (tid == 0) ctl->nlist = 0;
        if (!full) {
            a1 = INFINITY;
            a2 = INFINITY;
            chunk = 0;
            if (own) {
                unsigned id = (unsigned)tid * (unsigned)g.NG;
                const float* mc = v.Mc + tid;
                for (int q = 0; q < g.NG; ++q) {
                    track(*mc, id, a1, a2, chunk);
                    ++id;
                    mc += units;
                }
            }
            FINE(pf, 2);
            return;
        }
    }
    __builtin_amdgcn_s_setprio(0);
    filter_sweep_c(v, g, tid, a1, a2, chunk);
    __builtin_amdgcn_s_setprio(3);
    FINE(pf, 3);
}

template <typename T>
__device__ __forceinline__ double exact_delta_c(const CView<T>& v, int i, int j)
{
    View<T> w;
    w.tour = v.tour;
    w.X = v.X;
    return exact_delta(w, i, j);
}

// exact value of pair `pr` (0..31) of macro step `m` of local unit `u`; false for padding units
template <typename T>
__device__ __forceinline__ bool eval_chunk_pair_c(const CView<T>& v, const CGeo& g, int u, int m, int pr,
                                                  double& e, unsigned& idx)
{
    const int gpl = u % g.GPw, sl = u / g.GPw;
    const int gp = v.gp0 + gpl;
    if (gp >= g.GP) return false;
    const int gi = 2 * gp + (pr >> 4);
    const int r0 = 4 * ((v.rs0 + sl) * g.MS + m);
    const int c0 = r0 + 2 + 4 * gi;
    const int a = (pr >> 2) & 3, q = pr & 3;
    const int r = (r0 + a) % g.n, c = (c0 + a + q) % g.n;
    const int i = r < c ? r : c, j = r < c ? c : r;
    e = exact_delta_c(v, i, j);
    idx = ((unsigned)i << 16) | (unsigned)j;
    return true;
}

// exact evaluation of step group `id` (wave-uniform) by one wave: folds the group's pairs into (e, idx)
template <typename T>
__device__ __forceinline__ void eval_group_c(const CView<T>& v, const CGeo& g, int id, float thr, int lane,
                                             double& e, unsigned& idx)
{
    const int u = id / g.NG, m0 = (id - u * g.NG) * g.K;       // the K macro steps of the group
    // coarse groups: lane l recomputes the filter minimum of macro step m0 + l; only steps inside the
    // window can hold the argmin (same bound as for the group) and go to the exact evaluation
    unsigned long long sm = 1ull;
    if (g.K > 1) {
        float mm = INFINITY;
        if (lane < g.K && m0 + lane < g.MS) {
            const int gpl = u % g.GPw, sl = u / g.GPw, m = m0 + lane;
            walk_c(v, g, 4 * (sl * g.MS + m), sl * g.MS + 2 * gpl + 1 + m, 1, [&](float x) { mm = x; });
        }
        sm = __ballot(mm <= thr);
    }
    while (sm) {
        const int m = m0 + __builtin_amdgcn_readfirstlane(__ffsll((long long)sm) - 1);
        sm &= sm - 1;
        double d;
        unsigned di;
        if (lane < 32 && eval_chunk_pair_c(v, g, u, m, lane, d, di) && better(d, di, e, idx)) {
            e = d;
            idx = di;
        }
    }
}

// best move of this workgroup's tile: e = min(0, min exact delta), idx packed; valid in all threads
template <typename T>
__device__ void find_best_tile(const CView<T>& v, const CGeo& g, int tid, bool first_wg, unsigned last,
                               double& e, unsigned& idx, Prof& pf)
{
    Ctl* ctl = v.ctl;
    float a1, a2;
    unsigned chunk;
    unsigned long long t0 = 0;
    if (pf.on) t0 = clock64();
    sweep_c(v, g, tid, last, a1, a2, chunk, pf);
    if (pf.on) { const unsigned long long t1 = clock64(); pf.sweep += t1 - t0; t0 = t1; }
    const int nw = g.nt >> 6, lane = tid & 63, wv = tid >> 6;
    const float wm = wave_min(a1);
    if (lane == 0) ctl->wmin[wv] = wm;
    __syncthreads();
    float m = ctl->wmin[0];
    for (int w = 1; w < nw; ++w) m = fminf(m, ctl->wmin[w]);
    // only moves below -1e-6 are ever applied: pairs whose filter value proves delta >= -1e-6 need no
    // exact evaluation (a tile without improving moves must not chase ties among its own minima)
    const float thr = fminf(m + g.eps2, (float)(-1.0e-6 * v.sc) + 0.625f * g.eps2);
    unsigned long long cmask = __ballot(a1 <= thr);
    int amb = __ballot(a2 <= thr) != 0ull || __popcll(cmask) > 8;
    FINE(pf, 4);                                            // workgroup minimum of the filter (incl. barrier)
    if (pf.on) pf.cands += (unsigned)__popcll(cmask);
    const unsigned long long tc0 = pf.on ? clock64() : 0;
    e = 0.0;
    idx = kNoMove;
    if (!amb) {
        while (cmask) {
            const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)cmask) - 1);
            cmask &= cmask - 1;
            eval_group_c(v, g, __builtin_amdgcn_readlane((int)chunk, src), thr, lane, e, idx);
        }
        if (first_wg && tid == 32) {        // the pair (0, n-1), once per tour
            const double d = exact_delta_c(v, 0, g.n - 1);
            if (better(d, (unsigned)(g.n - 1), e, idx)) { e = d; idx = (unsigned)(g.n - 1); }
        }
        wave_best(e, idx);
    }
    if (pf.on) pf.apply += 0, pf.candcyc += clock64() - tc0;
    FINE(pf, 5);                                            // exact candidates of this wave
    if (lane == 0) { ctl->pE[wv] = e; ctl->pIdx[wv] = idx; ctl->pAmb[wv] = amb; }
    __syncthreads();
    FINE(pf, 6);                                            // barrier: the other waves' candidates
    e = ctl->pE[0];
    idx = ctl->pIdx[0];
    amb = ctl->pAmb[0];
    for (int w = 1; w < nw; ++w) {
        const double oe = ctl->pE[w];
        const unsigned oi = ctl->pIdx[w];
        amb |= ctl->pAmb[w];
        if (better(oe, oi, e, idx)) { e = oe; idx = oi; }
    }
    if (pf.on) { pf.amb += amb ? 1 : 0; }
    if (amb) {
        // Two candidate groups in one lane (or more than 8 per wave).  With the cache every candidate
        // is still known (all groups whose cached minimum is inside the window) and is evaluated
        // exactly, one group per wave at a time; without it (or if that list overflows) the whole tile
        // of this workgroup is evaluated exactly.
        __syncthreads();
        bool listed = false;
        e = 0.0;
        idx = kNoMove;
        if (g.inc) {
            const int units = g.GPw * g.NSw, groups = units * g.NG;
            for (int t = tid; t < groups; t += g.nt) {
                const int q = t / units, u = t - q * units;
                if (v.Mc[t] <= thr && v.gp0 + u % g.GPw < g.GP) {
                    const unsigned pos = atomicAdd(&ctl->nlist, 1u);
                    if (pos < (unsigned)g.lcap) v.list[pos] = (uint16_t)(u * g.NG + q);
                }
            }
            __syncthreads();
            const unsigned nl = ctl->nlist;
            __syncthreads();
            if (tid == 0) ctl->nlist = 0;
            if (nl <= (unsigned)g.lcap) {
                listed = true;
                for (unsigned c = (unsigned)wv; c < nl; c += (unsigned)nw)
                    eval_group_c(v, g, (int)v.list[c], thr, lane, e, idx);
            }
        }
        if (!listed) {
            const long long total = (long long)g.GPw * g.NSw * g.MS * 32;
            for (long long k = tid; k < total; k += g.nt) {
                double d;
                unsigned di;
                const int ms = (int)(k >> 5);
                if (eval_chunk_pair_c(v, g, ms / g.MS, ms % g.MS, (int)(k & 31), d, di) && better(d, di, e, idx)) {
                    e = d;
                    idx = di;
                }
            }
        }
        if (first_wg && tid == 0) {
            const double d = exact_delta_c(v, 0, g.n - 1);
            if (better(d, (unsigned)(g.n - 1), e, idx)) { e = d; idx = (unsigned)(g.n - 1); }
        }
        block_best(ctl, tid, g.nt, e, idx);
    }
    if (pf.on) pf.select += clock64() - t0;
}

template <typename T, int MAXNT>
__global__ void __launch_bounds__(MAXNT) k_two_opt_cluster(CParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const CGeo g = p.g;
    const int n = g.n, tid = threadIdx.x, nt = g.nt;
    const int cluster = blockIdx.x / g.CS, wc = blockIdx.x % g.CS;
    const int cr = wc / g.CD, cd = wc % g.CD;
    CView<T> v;
    v.ctl = reinterpret_cast<Ctl*>(smem);
    size_t off = align16(sizeof(Ctl));
    v.tour = reinterpret_cast<uint16_t*>(smem + off);
    off += align16(sizeof(uint16_t) * (size_t)(n + 1));
    v.Xq = reinterpret_cast<v2f*>(smem + off); off += sizeof(v2f) * 4 * (size_t)g.Qw;
    v.Yq = reinterpret_cast<v2f*>(smem + off); off += sizeof(v2f) * 4 * (size_t)g.Qw;
    v.Eq = reinterpret_cast<v2f*>(smem + off); off += sizeof(v2f) * 4 * (size_t)g.Qw;
    v.RX = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * (size_t)(g.rowsw + 8));
    v.RY = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * (size_t)(g.rowsw + 8));
    v.RE = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * (size_t)(g.rowsw + 8));
    v.Mc = nullptr;
    v.list = nullptr;
    if (g.inc) {
        const size_t groups = (size_t)g.GPw * g.NSw * g.NG;
        v.Mc = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * groups);
        v.list = reinterpret_cast<uint16_t*>(smem + off); off += align16(sizeof(uint16_t) * (size_t)g.lcap);
    }
    T* xl = reinterpret_cast<T*>(smem + off);
    v.rs0 = cr * g.NSw;
    v.gp0 = cd * g.GPw;
    v.ilo = v.rs0 * g.MS + 2 * v.gp0;
    v.rbase = 4 * v.rs0 * g.MS;
    Ctl* ctl = v.ctl;
    unsigned* cs = p.csync + (size_t)cluster * (size_t)p.cstride;
    unsigned epoch = 0;                         // barriers passed by this cluster so far

    for (int64_t tk = cluster; tk < p.num_tours; tk += p.num_clusters) {
        __syncthreads();
        if (tid == 0) ctl->nlist = 0;
        const T* X = reinterpret_cast<const T*>(p.points) + (size_t)(tk / p.tpi) * (size_t)n * 2;
        int64_t* gt = p.tours + (size_t)tk * (size_t)(n + 1);
        // ---- stage + validate (every workgroup of the cluster computes the same flags) -------
        int bad = 0;
        for (int k = tid; k <= n; k += nt) {
            const int64_t t = gt[k];
            if (t < 0 || t >= n) { bad |= TSP2OPT_ST_BAD_INDEX; v.tour[k] = 0; }
            else v.tour[k] = (uint16_t)t;
        }
        if (tid == 0 && gt[0] != gt[n]) bad |= TSP2OPT_ST_NOT_CLOSED;
        double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
        for (int k = tid; k < n; k += nt) {
            const T x = X[2 * k], y = X[2 * k + 1];
            if (g.xlds) { xl[2 * k] = x; xl[2 * k + 1] = y; }
            const double dx = (double)x, dy = (double)y;
            if (!(fabs(dx) <= 1.0e150) || !(fabs(dy) <= 1.0e150))
                bad |= (dx != dx || dy != dy || fabs(dx) == INFINITY || fabs(dy) == INFINITY)
                           ? TSP2OPT_ST_NONFINITE : TSP2OPT_ST_RANGE;
            mnx = fmin(mnx, dx); mxx = fmax(mxx, dx);
            mny = fmin(mny, dy); mxy = fmax(mxy, dy);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mnx = fmin(mnx, __shfl_xor(mnx, o)); mxx = fmax(mxx, __shfl_xor(mxx, o));
            mny = fmin(mny, __shfl_xor(mny, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
            bad |= __shfl_xor(bad, o);
        }
        if ((tid & 63) == 0) {
            const int w = tid >> 6;
            ctl->redd[4 * w] = mnx; ctl->redd[4 * w + 1] = mxx;
            ctl->redd[4 * w + 2] = mny; ctl->redd[4 * w + 3] = mxy;
            ctl->pIdx[w] = (unsigned)bad;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < (nt >> 6); ++w) {
                mnx = fmin(mnx, ctl->redd[4 * w]); mxx = fmax(mxx, ctl->redd[4 * w + 1]);
                mny = fmin(mny, ctl->redd[4 * w + 2]); mxy = fmax(mxy, ctl->redd[4 * w + 3]);
                bad |= (int)ctl->pIdx[w];
            }
            ctl->flags = bad;
            ctl->cx = 0.5 * (mnx + mxx);
            ctl->cy = 0.5 * (mny + mxy);
            const double half = fmax(0.5 * (mxx - mnx), 0.5 * (mxy - mny));
            int ex = 0;
            double sc = 1.0;
            if (half > 0.0 && half < INFINITY) { (void)frexp(half, &ex); sc = ldexp(1.0, -ex); }
            ctl->sc = sc;
        }
        __syncthreads();
        const int flags = ctl->flags;
        v.X = g.xlds ? xl : X;
        v.cx = ctl->cx; v.cy = ctl->cy; v.sc = ctl->sc;
        __syncthreads();
        if (flags != 0) {                       // invalid input: same verdict in every workgroup
            if (wc == 0 && tid == 0) { if (p.status) p.status[tk] = flags; p.iters[tk] = 0; }
            continue;
        }
        stage_c(v, g, tid, 0, n - 1, 0, n - 1);
        __syncthreads();

        // ---- improvement loop (independent termination) ------------------------------------------
        const long long t_start = p.time_limit_ticks ? wall_clock64() : 0;
        int it = 0, sweeps = 0, st = 0;
        bool aborted = false;
        Prof pf{p.stamps != nullptr, 0, 0, 0};
        unsigned long long t_x = 0, t_tour = pf.on ? clock64() : 0;
        unsigned last = kFullSweep;             // reversal applied since the previous sweep
        for (;;) {
            double e;
            unsigned idx, fl = 0;
            find_best_tile(v, g, tid, wc == 0, last, e, idx, pf);
            ++sweeps;
            unsigned long long tb = pf.on ? clock64() : 0;
            unsigned myflags = 0;
            if (wc == 0 && p.time_limit_ticks && wall_clock64() - t_start > p.time_limit_ticks)
                myflags |= kFlagStop;
            FINE_START(pf);
            if (!cluster_exchange(ctl, cs, g.CS, wc, epoch, tid, nt, myflags, e, idx, fl)) { aborted = true; break; }
            ++epoch;
            FINE(pf, 11);                                                    // cluster exchange
            if (pf.on) { const unsigned long long t1 = clock64(); t_x += t1 - tb; tb = t1; }
            if (!(e < -1.0e-6)) break;                                       // two_opt.py:40
            const int i = (int)(idx >> 16), j = (int)(idx & 0xffffu);
            const int h = (j - i) >> 1;
            for (int k = tid; k < h; k += nt) {                              // two_opt.py:42
                const uint16_t a = v.tour[i + 1 + k], b = v.tour[j - k];
                v.tour[i + 1 + k] = b;
                v.tour[j - k] = a;
            }
            __syncthreads();
            FINE(pf, 8);                                                     // tour reversal
            stage_c(v, g, tid, i + 1, j, i, j);
            FINE(pf, 9);                                                     // restaging of the filter copy
            __syncthreads();
            FINE(pf, 10);
            last = idx;
            if (pf.on) pf.apply += clock64() - tb;
            ++it;
            if (it >= p.max_iter) { st |= TSP2OPT_ST_CAPPED; break; }        // two_opt.py:47
            if (fl & kFlagStop) { st |= TSP2OPT_ST_TIME; break; }            // plugin two_opt.py:63-65
        }
        __builtin_amdgcn_s_setprio(0);
        if (pf.on && tid == 0) {
            atomicAdd(&p.stamps[0], pf.sweep);
            atomicAdd(&p.stamps[1], pf.select);
            atomicAdd(&p.stamps[2], pf.apply);
            atomicAdd(&p.stamps[3], t_x);
            atomicAdd(&p.stamps[5], (unsigned long long)pf.amb);
            atomicAdd(&p.stamps[6], (unsigned long long)pf.cands);
            atomicAdd(&p.stamps[7], pf.candcyc);
            atomicAdd(&p.stamps[4], clock64() - t_tour);
#ifdef TSP2OPT_FINE
            for (int k = 0; k < 16; ++k) atomicAdd(&p.stamps[8 + k], pf.fine[k]);
#endif
        }
        if (aborted) {
            if (wc == 0 && tid == 0 && p.status) p.status[tk] = TSP2OPT_ST_INTERNAL;
            return;                             // cluster broken: leave the remaining tours untouched
        }
        if (wc == 0) {
            for (int k = tid; k <= n; k += nt) gt[k] = (int64_t)v.tour[k];
            if (tid == 0) {
                p.iters[tk] = it;
                if (p.work_sweeps) p.work_sweeps[tk] += sweeps;
                if (p.status) p.status[tk] = st;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// coupled mode helpers
// ------------------------------------------------------------------------------------------
// target[g] = max over the group's own iteration counts (= the first iteration at which every
// tour of the group is inactive, unless a tour re-activates: see k_group_lockstep)
__global__ void k_group_max(const int32_t* iters, int32_t* target, int32_t* group_flag,
                            int64_t groups, int group_size, unsigned* queue)
{
    const int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gidx == 0) *queue = 0;
    if (gidx >= groups) return;
    int m = 0;
    for (int b = 0; b < group_size; ++b) m = max(m, iters[gidx * group_size + b]);
    target[gidx] = m;
    group_flag[gidx] = 0;
}

// General fallback of the coupled mode: one workgroup per flagged group replays the reference's
// lock-step loop (search/two_opt.py:21-48) from the group's current iteration, with tours and
// coordinates in global memory.  Only reached if a tour is active again at the group's target,
// which needs a delta within rounding noise of -1e-6; kept for exactness, not speed.
template <typename T>
__global__ void __launch_bounds__(1024) k_group_lockstep(Params p, int32_t* target, int force_all)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = p.g.n, tid = threadIdx.x, nt = blockDim.x;
    const int64_t grp = blockIdx.x;
    if (!force_all && p.group_flag[grp] == 0) return;
    for (int b = 0; b < p.group_size; ++b) {
        const int st = p.status ? p.status[grp * p.group_size + b] : 0;
        if (st & (TSP2OPT_ST_BAD_INDEX | TSP2OPT_ST_NOT_CLOSED | TSP2OPT_ST_NONFINITE | TSP2OPT_ST_RANGE))
            return;                                   // invalid group: left as is
    }
    View<T> v;
    v.ctl = reinterpret_cast<Ctl*>(smem);
    v.tour = reinterpret_cast<uint16_t*>(smem + align16(sizeof(Ctl)));
    unsigned* moves = reinterpret_cast<unsigned*>(smem + align16(sizeof(Ctl)) +
                                                  align16(sizeof(uint16_t) * (size_t)(n + 1)));
    v.Xq = nullptr; v.Yq = nullptr; v.Eq = nullptr;
    v.X = reinterpret_cast<const T*>(p.points) +
          (size_t)((grp * p.group_size) / p.tpi) * (size_t)n * 2;
    Geo g = p.g;
    g.nt = nt;
    int it = target[grp];
    while (it < p.max_iter) {
        bool any = false;
        for (int b = 0; b < p.group_size; ++b) {
            int64_t* gt = p.tours + (size_t)(grp * p.group_size + b) * (size_t)(n + 1);
            __syncthreads();
            for (int k = tid; k <= n; k += nt) v.tour[k] = (uint16_t)gt[k];
            __syncthreads();
            double e;
            unsigned idx;
            exact_sweep(v, n, tid, nt, e, idx);
            block_best(v.ctl, tid, nt, e, idx);
            if (tid == 0) moves[b] = idx;
            any = any || (e < -1.0e-6);
        }
        if (!any) break;
        __syncthreads();
        for (int b = 0; b < p.group_size; ++b) {
            int64_t* gt = p.tours + (size_t)(grp * p.group_size + b) * (size_t)(n + 1);
            const unsigned idx = moves[b];
            const int i = (int)(idx >> 16), j = (int)(idx & 0xffffu), h = (j - i) >> 1;
            for (int k = tid; k < h; k += nt) {
                const int64_t a = gt[i + 1 + k], c = gt[j - k];
                gt[i + 1 + k] = c;
                gt[j - k] = a;
            }
        }
        __threadfence_block();
        ++it;
    }
    if (tid == 0) target[grp] = it;
}

__global__ void k_group_finish(int32_t* iters, int32_t* sweeps, int32_t* status,
                               const int32_t* target, int64_t num_tours, int group_size, int max_iter)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_tours) return;
    const int t = target[k / group_size];
    if (status && (status[k] & (TSP2OPT_ST_BAD_INDEX | TSP2OPT_ST_NOT_CLOSED |
                                TSP2OPT_ST_NONFINITE | TSP2OPT_ST_RANGE))) {
        if (iters) iters[k] = 0;
        if (sweeps) sweeps[k] = 0;
        return;
    }
    if (iters) iters[k] = t;
    if (sweeps) sweeps[k] = t >= max_iter ? t : t + 1;
    if (status) status[k] = t >= max_iter ? TSP2OPT_ST_CAPPED : 0;
}

__global__ void k_indep_finish(const int32_t* iters, int32_t* sweeps, const int32_t* status,
                               int64_t num_tours, int max_iter)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_tours || !sweeps) return;
    const int st = status ? status[k] : 0;
    if (st & (TSP2OPT_ST_BAD_INDEX | TSP2OPT_ST_NOT_CLOSED | TSP2OPT_ST_NONFINITE | TSP2OPT_ST_RANGE))
        sweeps[k] = 0;
    else
        sweeps[k] = (st & (TSP2OPT_ST_CAPPED | TSP2OPT_ST_TIME)) ? iters[k] : iters[k] + 1;
}

// closed tour length, one wave per tour, summed in double
template <typename T>
__global__ void k_lengths(const T* points, const int64_t* tours, double* out, int64_t num_tours,
                          int n, int tpi)
{
    const int64_t tk = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tk >= num_tours) return;
    const int lane = threadIdx.x & 63;
    const T* X = points + (size_t)(tk / tpi) * (size_t)n * 2;
    const int64_t* t = tours + (size_t)tk * (size_t)(n + 1);
    double s = 0.0;
    for (int k = lane; k < n; k += 64) {
        const int64_t a = t[k], b = t[k + 1];
        if (a < 0 || a >= n || b < 0 || b >= n) { s = NAN; continue; }
        const double dx = (double)X[2 * a] - (double)X[2 * b];
        const double dy = (double)X[2 * a + 1] - (double)X[2 * b + 1];
        s += __builtin_sqrt(dx * dx + dy * dy);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[tk] = s;
}

// Validation + length of every tour, one wave per tour (replaces the per-tour Python checks of
// nar/local_search/base.py:67-72 and the per-edge loop of utils/evaluator.py:5-13).
//   valid bit 0: first n entries are a permutation of 0..n-1   (base.py:31-32 is_valid_tour)
//   valid bit 1: tour[n] == 0                                   (base.py:71)
//   valid bit 2: tour[n] == tour[0]
template <typename T>
__global__ void k_evaluate(const T* points, const int64_t* tours, double* lengths, int32_t* valid,
                           int64_t num_tours, int n, int tpi, int words)
{
    extern __shared__ unsigned seen_all[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    if (tk >= num_tours) return;                     // whole wave leaves together
    unsigned* seen = seen_all + (size_t)wv * words;
    const T* X = points + (size_t)(tk / tpi) * (size_t)n * 2;
    const int64_t* t = tours + (size_t)tk * (size_t)(n + 1);
    for (int k = lane; k < words; k += 64) seen[k] = 0u;
    __builtin_amdgcn_wave_barrier();
    int bad = 0;
    double s = 0.0;
    for (int k = lane; k < n; k += 64) {
        const int64_t a = t[k], b = t[k + 1];
        const bool ain = a >= 0 && a < n, bin = b >= 0 && b < n;
        if (!ain) bad = 1;
        else {
            const unsigned bit = 1u << (a & 31);
            if (atomicOr(&seen[a >> 5], bit) & bit) bad = 1;      // duplicate
        }
        if (ain && bin) {
            const double dx = (double)X[2 * a] - (double)X[2 * b];
            const double dy = (double)X[2 * a + 1] - (double)X[2 * b + 1];
            s += __builtin_sqrt(dx * dx + dy * dy);
        } else s = NAN;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const bool anybad = __ballot(bad) != 0ull;
    if (lane == 0) {
        int v = anybad ? 0 : 1;
        if (t[n] == 0) v |= 2;
        if (t[n] == t[0]) v |= 4;
        if (lengths) lengths[tk] = s;
        if (valid) valid[tk] = v;
    }
}

// best tour of every instance: first index of the minimal length among its tours
// (tsp_gnn.py:220-221 np.min over the parallel samples; narsolver keeps the per-instance best)
__global__ void k_best_of(const double* lengths, int64_t* best_index, double* best_length,
                          int64_t num_instances, int tpi)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_instances) return;
    int64_t bk = i * tpi;
    double bl = lengths[bk];
    for (int k = 1; k < tpi; ++k) {
        const double l = lengths[i * tpi + k];
        if (l < bl || (bl != bl && l == l)) { bl = l; bk = i * tpi + k; }
    }
    if (best_index) best_index[i] = bk;
    if (best_length) best_length[i] = bl;
}

template <typename T>
__global__ void k_sqrt_check(const T* in, T* out, int64_t count)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) out[k] = sqrt_rn(in[k]);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// optional timing of the persistent kernel with HIP events on the caller's stream
// (tsp2opt_profile_enable / tsp2opt_profile_last_ms; used by bench.py for the roofline figure)
constexpr int kMaxDev = 16;
struct ProfSlot {
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // main begin/end, forced begin/end
    bool have_main = false, have_forced = false;
};
int g_profile = 0;
int g_dbg_sweeps = 0;                // diagnostic: filter sweeps only (tsp2opt_debug_sweep_only)
int g_stamps = 0;                    // phase timers into workspace bytes [64, 128)
ProfSlot g_prof[kMaxDev];

int prof_record(int device, int which, hipStream_t stream)
{
    if (!g_profile || device < 0 || device >= kMaxDev) return 0;
    ProfSlot& s = g_prof[device];
    if (!s.ev[which]) HIP_TRY(hipEventCreate(&s.ev[which]));
    HIP_TRY(hipEventRecord(s.ev[which], stream));
    return 0;
}

struct Plan {
    Geo g;
    size_t lds;
    int S, W;
};

constexpr size_t kLdsMax = 160 * 1024;
constexpr size_t kCsyncBytes = 64 * 1024;   // cluster barrier words + slots (<= 256 workgroups)
const int kQClasses[] = {16, 32, 64, 128, 256, 512};

int g_force_nt = 0;         // diagnostic: threads per workgroup of the single-workgroup plan
int g_lcap16 = 13;          // diagnostic: changed-group list capacity of the cluster kernel, in 1/16 of the groups
int g_lcap16s = 13;         // changed-tile list capacity of the one-workgroup / full-window kernel, in 1/16 of the tiles
int g_inc_min_tiles = 2;    // tile cache only when a lane owns at least this many tiles (measured: TSP-50 / TSP-100)

// cs > 1: geometry of one workgroup of a full-window cluster of cs workgroups (lane groups interleaved)
Plan make_plan(int n, int dtype, int cs = 1)
{
    Plan pl{};
    Geo& g = pl.g;
    g.CS = cs;
    g.wc = 0;
    const size_t tsz = dtype == TSP2OPT_DTYPE_F64 ? 8 : 4;
    g.n = n;
    g.eps2 = dtype == TSP2OPT_DTYPE_F64 ? 2.0f * 0x1p-16f : 2.0f * 0x1p-15f;
    const size_t base = align16(sizeof(Ctl)) + align16(sizeof(uint16_t) * (size_t)(n + 1));
    const size_t xbytes = (size_t)n * 2 * tsz;
    // threads per workgroup: about 500 pairs per lane per sweep
    int nt = n <= 160 ? 64 : n <= 320 ? 128 : n <= 640 ? 256 : n <= 1280 ? 512 : 1024;
    if (g_force_nt) nt = g_force_nt;
    g.nt = nt;
    g.qclass = 0;
    size_t cache = 0;
    {
        const int dn = n / 2 - 1;                       // cyclic diagonals 2..n/2
        const int GP = (dn + 7) / 8;                    // lane groups of 8 adjacent diagonals
        const int G = 2 * GP;
        const int GPl = (GP + cs - 1) / cs;             // lane groups per workgroup
        g.GPl = GPl;
        // every walked diagonal d = 2..8GP+1 must be a valid pair distance: d <= n-2
        if (GP >= 1 && 8 * GP + 1 <= n - 2) {
            const int R = (n + 3) / 4;                  // macro-step rows needed
            int NS = nt / GPl;
            if (NS < 1) NS = 1;
            if (NS > (R + 1) / 2) NS = (R + 1) / 2;
            int MS = (R + NS - 1) / NS;
            MS += MS & 1;                               // tiles are two macro steps
            NS = (R + MS - 1) / MS;                     // drop segments that would be pure padding
            const int rows = 4 * NS * MS;
            const int W = rows + 8 * GP + 8;            // multiple of 8
            const int q = W / 8 + 1;
            for (int qc : kQClasses) {
                if (W >= 3 * n) break;                  // wrap3(): window positions must stay below 3n
                if (qc >= q && base + (size_t)qc * 192 <= kLdsMax) {
                    g.qclass = qc; g.GP = GP; g.G = G; g.NS = NS; g.MS = MS;
                    g.W = W;
                    pl.S = 4 * MS; pl.W = W;
                    break;
                }
            }
            if (g.qclass) {
                // minimum cache (4 B per tile) + changed-tile list (2 B per entry, 13/16 of the tiles: measured
                // on TSP-100/500/1000, beyond that a plain sweep is cheaper than recomputing tile by tile)
                const size_t tiles = (size_t)GPl * NS * (MS / 2);
                const size_t lcap = tiles * (size_t)g_lcap16s / 16 + 1;
                const size_t need = align16(4 * tiles) + align16(2 * lcap);
                if (MS / 2 >= g_inc_min_tiles && (size_t)NS * (MS / 2) <= 255 && GPl <= 255 && MS / 2 <= 64 && GPl * NS <= nt &&
                    base + (size_t)g.qclass * 192 + need <= kLdsMax) {
                    g.inc = 1;
                    g.lcap = (int)lcap;
                    cache = need;
                }
            }
        }
    }
    pl.lds = base + (size_t)g.qclass * 192 + cache;
    // cache the exact coordinates in LDS when that still leaves >= 4 workgroups per CU (or fits at all
    // for single-workgroup-per-CU sizes)
    g.xlds = 0;
    if (pl.lds + xbytes <= kLdsMax && (pl.lds + xbytes <= 40 * 1024 || nt >= 512 || cs > 1)) {
        g.xlds = 1;
        pl.lds += align16(xbytes);
    }
    return pl;
}

template <typename T, int Q>
int launch_main(const Params& p, const Plan& pl, int device, hipStream_t stream, int* blocks_per_cu)
{
    auto kern = k_two_opt<T, Q, false>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, pl.g.nt, pl.lds));
    if (per_cu < 1) per_cu = 1;
    if (blocks_per_cu) { *blocks_per_cu = per_cu; return 0; }
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    int64_t grid = (int64_t)cus * per_cu;
    if (grid > p.num_tours) grid = p.num_tours;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)pl.g.nt), pl.lds, stream, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

// full-window cluster: `clusters` x CS workgroups, at most one per CU (co-residency)
template <typename T, int Q>
int launch_fw(Params p, const Plan& pl, int cus, hipStream_t stream)
{
    auto kern = k_two_opt<T, Q, true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
    const int CS = pl.g.CS;
    int clusters = cus / CS;
    if (clusters > p.num_tours) clusters = (int)p.num_tours;
    if (clusters < 1) return fail(TSP2OPT_E_UNSUPPORTED, "cluster does not fit the device");
    p.num_clusters = clusters;
    p.cstride = ((16 + 2 * CS * kSlotDw) + 63) & ~63;
    if ((size_t)clusters * (size_t)p.cstride * 4 > kCsyncBytes)
        return fail(TSP2OPT_E_UNSUPPORTED, "internal: cluster sync area too small");
    hipLaunchKernelGGL(kern, dim3((unsigned)(clusters * CS)), dim3((unsigned)pl.g.nt), pl.lds, stream, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T>
int dispatch_fw(const Params& p, const Plan& pl, int cus, hipStream_t stream)
{
    switch (pl.g.qclass) {
    case 128: return launch_fw<T, 128>(p, pl, cus, stream);
    case 256: return launch_fw<T, 256>(p, pl, cus, stream);
    case 512: return launch_fw<T, 512>(p, pl, cus, stream);
    }
    return fail(TSP2OPT_E_ARG, "internal: no full-window cluster kernel for this filter class");
}

template <typename T>
int dispatch_main(const Params& p, const Plan& pl, int device, hipStream_t stream, int* bpc = nullptr)
{
    switch (pl.g.qclass) {
    case 0: return launch_main<T, 0>(p, pl, device, stream, bpc);
    case 16: return launch_main<T, 16>(p, pl, device, stream, bpc);
    case 32: return launch_main<T, 32>(p, pl, device, stream, bpc);
    case 64: return launch_main<T, 64>(p, pl, device, stream, bpc);
    case 128: return launch_main<T, 128>(p, pl, device, stream, bpc);
    case 256: return launch_main<T, 256>(p, pl, device, stream, bpc);
    case 512: return launch_main<T, 512>(p, pl, device, stream, bpc);
    }
    return fail(TSP2OPT_E_ARG, "internal: bad filter class");
}

// workspace layout: [0,256) control words (queue head at 0); then target[groups], flag[groups],
// iters[num_tours], work[num_tours], status[num_tours] (each 256-byte aligned)
struct WsLayout {
    size_t off_target, off_flag, off_iters, off_work, off_status, off_csync, total;
};

WsLayout ws_layout(int64_t num_tours, int group_size)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t groups = (size_t)(num_tours / (group_size > 0 ? group_size : 1)) + 1;
    WsLayout w{};
    size_t o = 256;
    w.off_target = o; o += up(groups * 4);
    w.off_flag = o; o += up(groups * 4);
    w.off_iters = o; o += up((size_t)num_tours * 4 + 4);
    w.off_work = o; o += up((size_t)num_tours * 4 + 4);
    w.off_status = o; o += up((size_t)num_tours * 4 + 4);
    w.off_csync = o; o += kCsyncBytes;
    w.total = o;
    return w;
}

struct CPlan {
    CGeo g;
    size_t lds;
    bool ok;
};

// geometry of the cluster kernel for at most `cs_cap` workgroups per tour
int g_cluster_inc = 1;      // 0: cluster kernel without the group-minimum cache (diagnostic)

CPlan make_cluster_plan(int n, int dtype, int cs_cap, bool smallest)
{
    CPlan best{};
    best.ok = false;
    if (n < 18 || cs_cap < 2) return best;
    const size_t tsz = dtype == TSP2OPT_DTYPE_F64 ? 8 : 4;
    const int dn = n / 2 - 1;
    const int GP = ((dn + 3) / 4 + 1) / 2;
    if (GP < 1 || 8 * GP + 1 > n - 2) return best;
    const int R = (n + 3) / 4;
    const size_t base = align16(sizeof(Ctl)) + align16(sizeof(uint16_t) * (size_t)(n + 1));
    const size_t xbytes = (size_t)n * 2 * tsz;
    // Exhaustive search over (row bands, diagonal bands, threads): minimise the modelled time of
    // one iteration = sweep (macro steps x passes x waves per SIMD, ~0.4 us per wave macro step on a
    // saturated SIMD) + fixed per-iteration cost (selection, reversal, cluster barrier).
    // `smallest`: many tours, tiling only for LDS capacity -> maximise work per workgroup instead.
    double best_cost = 1e300;
    for (int cr = 1; cr <= cs_cap; ++cr) {
        for (int cd = 1; cr * cd <= cs_cap; ++cd) {
            const int cs = cr * cd;
            if (cs < 2 || cd > GP || cr > R) continue;
            const int GPw = (GP + cd - 1) / cd;
            const int rmw = (R + cr - 1) / cr;          // macro rows per row band
            for (int nt : {256, 512, 768}) {            // <= 768 threads: 3 waves/SIMD, no spills
                if (n < 700 && nt != 256) continue;
                int NSw = nt / GPw;
                if (NSw < 1) NSw = 1;
                if (NSw > rmw) NSw = rmw;
                const int MS = (rmw + NSw - 1) / NSw;
                if (MS < 4 && cs > 2) continue;         // too little work per cluster barrier
                const int units = GPw * NSw;
                const int passes = (units + nt - 1) / nt;
                const int wps = (nt / 64 + 3) / 4;      // waves per SIMD
                const int Qw = NSw * MS + 2 * GPw + 3;
                const int rowsw = 4 * NSw * MS;
                size_t lds = base + (size_t)Qw * 96 + 3 * align16(sizeof(float) * (size_t)(rowsw + 8));
                if (lds > kLdsMax) continue;
                // group-minimum cache: the finest granularity K (macro steps per entry) that fits
                int K = 0, Ksh = 0, NG = MS, lcap = 0;
                size_t cache = 0;
                if (units <= nt && g_cluster_inc) {
                    for (int k = 1, sh = 2; k <= 64; k *= 2, ++sh) {
                        const int ng = (MS + k - 1) / k;
                        const size_t groups = (size_t)units * ng;
                        const size_t cap = groups * (size_t)g_lcap16 / 16 + 1;
                        const size_t need = align16(4 * groups) + align16(2 * cap);
                        if (ng <= 64 && (size_t)nt * ng <= 65535 && lds + need <= kLdsMax) {
                            K = k; Ksh = sh; NG = ng; lcap = (int)cap; cache = need;
                            break;
                        }
                    }
                }
                // measured: incremental sweeps recompute a little under half of the pairs
                const double t_iter = (K ? 0.55 : 1.0) * 0.4 * MS * passes * wps + 12.0 + 0.15 * cs;
                const double cost = smallest ? t_iter * cs : t_iter;    // total WG-time vs latency
                if (cost >= best_cost) continue;
                lds += cache;
                CGeo g{};
                g.n = n; g.nt = nt; g.GP = GP; g.CS = cs; g.CR = cr; g.CD = cd;
                g.GPw = GPw; g.NSw = NSw; g.MS = MS; g.Qw = Qw; g.rowsw = rowsw;
                g.K = K ? K : 1; g.Ksh = K ? Ksh : 2; g.NG = NG; g.inc = K ? 1 : 0; g.lcap = lcap;
                g.eps2 = dtype == TSP2OPT_DTYPE_F64 ? 2.0f * 0x1p-16f : 2.0f * 0x1p-15f;
                g.xlds = 0;
                if (lds + align16(xbytes) <= kLdsMax) { g.xlds = 1; lds += align16(xbytes); }
                best.g = g; best.lds = lds; best.ok = true;
                best_cost = cost;
            }
        }
    }
    return best;
}

template <typename T, int MAXNT>
int launch_cluster(CParams p, const CPlan& cp, int cus, hipStream_t stream)
{
    auto kern = k_two_opt_cluster<T, MAXNT>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cp.lds));
    int clusters = cus / cp.g.CS;                       // one workgroup per CU: all co-resident
    if (clusters > p.num_tours) clusters = (int)p.num_tours;
    if (clusters < 1) return fail(TSP2OPT_E_UNSUPPORTED, "cluster does not fit the device");
    p.num_clusters = clusters;
    p.cstride = ((16 + 2 * cp.g.CS * kSlotDw) + 63) & ~63;
    if ((size_t)clusters * (size_t)p.cstride * 4 > kCsyncBytes)
        return fail(TSP2OPT_E_UNSUPPORTED, "internal: cluster sync area too small");
    hipLaunchKernelGGL(kern, dim3((unsigned)(clusters * cp.g.CS)), dim3((unsigned)cp.g.nt), cp.lds,
                       stream, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

int g_cluster_mode = -1;    // -1 auto, 0 never, >= 2 force that many workgroups per tour (tests)

int g_cluster_variant = 0;  // 0 auto, 1 band kernel only (k_two_opt_cluster), 2 full-window kernel only

// Several workgroups per tour?  kind 0: no (one workgroup per tour), 1: band cluster
// (k_two_opt_cluster, any N), 2: full-window cluster (k_two_opt<.., true>, N within the fast path).
struct ClusterChoice {
    int kind;
    CPlan band;
    Plan fw;
};

ClusterChoice choose_cluster(int n, int dtype, int64_t num_tours, int cus, const Plan& pl)
{
    ClusterChoice c{};
    c.kind = 0;
    int cap = (int)std::min<int64_t>(64, cus / std::max<int64_t>(1, num_tours));
    const bool forced = g_cluster_mode >= 2;
    if (forced) cap = std::min(g_cluster_mode, cus);
    // full-window cluster: needs the fast path (and one of the instantiated filter classes) with the
    // whole window in LDS; at least 4 lane groups per workgroup
    if (g_cluster_variant != 1 && cap >= 2 && pl.g.qclass != 0 && (forced || n >= 700)) {
        const int cs = std::min(cap, std::max(1, pl.g.GP / 4));
        if (cs >= 2) {
            const Plan fw = make_plan(n, dtype, cs);
            if (fw.g.qclass >= 128) { c.kind = 2; c.fw = fw; return c; }
        }
    }
    if (g_cluster_variant == 2) return c;
    bool smallest = false;
    if (!forced) {
        if (pl.g.qclass == 0 && cap < 2) { cap = 64; smallest = true; }        // tiling for LDS only
        else if (pl.g.qclass != 0 && (n < 700 || cap < 3)) cap = 0;
        // (small tours are not worth a barrier; with two band workgroups per tour the one-workgroup
        //  kernel with its finer incremental sweeps is faster even on half the CUs)
    }
    c.band = make_cluster_plan(n, dtype, cap, smallest);
    if (c.band.ok) c.kind = 1;
    return c;
}

template <typename T>
int solve_t(const void* points, int64_t* tours, int32_t* iters_out, int32_t* sweeps_out,
            double* lengths_out, int32_t* status_out, int64_t num_tours, int n, int tpi,
            int group_size, int max_iter, double time_limit_s, int dtype, int device, void* ws,
            hipStream_t stream)
{
    const Plan pl = make_plan(n, dtype);
    const WsLayout wl = ws_layout(num_tours, group_size);
    unsigned char* w8 = static_cast<unsigned char*>(ws);
    Params p{};
    p.points = points;
    p.tours = tours;
    p.iters = iters_out ? iters_out : reinterpret_cast<int32_t*>(w8 + wl.off_iters);
    p.work_sweeps = reinterpret_cast<int32_t*>(w8 + wl.off_work);
    p.status = status_out ? status_out : reinterpret_cast<int32_t*>(w8 + wl.off_status);
    p.target = reinterpret_cast<int32_t*>(w8 + wl.off_target);
    p.group_flag = reinterpret_cast<int32_t*>(w8 + wl.off_flag);
    p.queue = reinterpret_cast<unsigned*>(w8);
    p.num_tours = num_tours;
    p.tpi = tpi;
    p.group_size = group_size;
    p.max_iter = max_iter;
    p.mode = kModeIndep;
    p.time_limit_ticks = 0;
    p.stamps = g_stamps ? reinterpret_cast<unsigned long long*>(w8 + 64) : nullptr;
    p.dbg_sweeps = g_dbg_sweeps;
    p.g = pl.g;
    if (time_limit_s > 0.0 && group_size == 1) {
        int khz = 0;
        HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device));
        if (khz <= 0) khz = 100000;
        p.time_limit_ticks = (long long)(time_limit_s * 1000.0 * (double)khz);
        if (p.time_limit_ticks < 1) p.time_limit_ticks = 1;
    }
    // zero the control words, the work counters and (coupled mode) nothing else: target/flag are
    // written by k_group_max before they are read
    HIP_TRY(hipMemsetAsync(w8, 0, 256, stream));
    HIP_TRY(hipMemsetAsync(w8 + wl.off_work, 0, (size_t)num_tours * 4 + 4, stream));
    if (num_tours == 0) return 0;
    int rc = prof_record(device, 0, stream);
    if (rc) return rc;
    // few tours (or N beyond one workgroup's LDS): several workgroups per tour
    ClusterChoice cc{};
    cc.kind = 0;
    if (group_size == 1 && g_cluster_mode != 0 && g_dbg_sweeps == 0) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        cc = choose_cluster(n, dtype, num_tours, cus, pl);
        if (cc.kind != 0) HIP_TRY(hipMemsetAsync(w8 + wl.off_csync, 0, kCsyncBytes, stream));
        if (cc.kind == 1) {
            const CPlan& cp = cc.band;
            CParams c{};
            c.points = points; c.tours = tours; c.iters = p.iters; c.work_sweeps = p.work_sweeps;
            c.status = p.status; c.csync = reinterpret_cast<unsigned*>(w8 + wl.off_csync);
            c.num_tours = num_tours; c.tpi = tpi; c.max_iter = max_iter;
            c.time_limit_ticks = p.time_limit_ticks; c.g = cp.g;
            c.stamps = p.stamps;
            rc = cp.g.nt <= 512 ? launch_cluster<T, 512>(c, cp, cus, stream)
                                : launch_cluster<T, 768>(c, cp, cus, stream);
            if (rc) return rc;
        } else if (cc.kind == 2) {
            Params pc = p;
            pc.g = cc.fw.g;
            pc.csync = reinterpret_cast<unsigned*>(w8 + wl.off_csync);
            rc = dispatch_fw<T>(pc, cc.fw, cus, stream);
            if (rc) return rc;
        }
    }
    if (cc.kind == 0) {
        rc = dispatch_main<T>(p, pl, device, stream);
        if (rc) return rc;
    }
    rc = prof_record(device, 1, stream);
    if (rc) return rc;
    if (g_profile && device < kMaxDev) { g_prof[device].have_main = true; g_prof[device].have_forced = false; }
    const unsigned tb = 256;
    const unsigned tgrid = (unsigned)((num_tours + tb - 1) / tb);
    if (group_size > 1) {
        const int64_t groups = num_tours / group_size;
        int32_t* target = reinterpret_cast<int32_t*>(w8 + wl.off_target);
        hipLaunchKernelGGL(k_group_max, dim3((unsigned)((groups + tb - 1) / tb)), dim3(tb), 0, stream,
                           p.iters, target, p.group_flag, groups, group_size, p.queue);
        HIP_TRY(hipGetLastError());
        Params pf = p;
        pf.mode = kModeForced;
        rc = prof_record(device, 2, stream);
        if (rc) return rc;
        rc = dispatch_main<T>(pf, pl, device, stream);
        if (rc) return rc;
        rc = prof_record(device, 3, stream);
        if (rc) return rc;
        if (g_profile && device < kMaxDev) g_prof[device].have_forced = true;
        const int lnt = n <= 256 ? 256 : 1024;
        const size_t llds = align16(sizeof(Ctl)) + align16(sizeof(uint16_t) * (size_t)(n + 1)) +
                            align16(sizeof(unsigned) * (size_t)group_size);
        if (llds > kLdsMax) return fail(TSP2OPT_E_UNSUPPORTED, "group_size too large");
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_group_lockstep<T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
        hipLaunchKernelGGL(k_group_lockstep<T>, dim3((unsigned)groups), dim3(lnt), llds, stream, pf,
                           target, 0);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_group_finish, dim3(tgrid), dim3(tb), 0, stream, iters_out, sweeps_out,
                           p.status, target, num_tours, group_size, max_iter);
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_indep_finish, dim3(tgrid), dim3(tb), 0, stream, p.iters, sweeps_out,
                           p.status, num_tours, max_iter);
        HIP_TRY(hipGetLastError());
    }
    if (lengths_out) {
        const unsigned lgrid = (unsigned)((num_tours + 3) / 4);
        hipLaunchKernelGGL(k_lengths<T>, dim3(lgrid), dim3(256), 0, stream,
                           reinterpret_cast<const T*>(points), tours, lengths_out, num_tours, n, tpi);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

}  // namespace t2o

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace t2o;

extern "C" int tsp2opt_version(void) { return TSP2OPT_VERSION; }

extern "C" const char* tsp2opt_last_error(void) { return g_err.c_str(); }

extern "C" int64_t tsp2opt_workspace_bytes(int64_t num_tours, int32_t n, int32_t group_size)
{
    if (num_tours < 0 || n < 1 || group_size < 1) return fail(TSP2OPT_E_ARG, "bad size");
    return (int64_t)ws_layout(num_tours, group_size).total;
}

extern "C" int tsp2opt_solve(const void* points, int64_t* tours, int32_t* iters_out,
                             int32_t* sweeps_out, double* lengths_out, int32_t* status_out,
                             int64_t num_tours, int32_t n, int32_t tours_per_instance,
                             int32_t group_size, int32_t max_iterations, double time_limit_s,
                             int32_t dtype, int32_t device, void* workspace, int64_t workspace_bytes,
                             void* hip_stream)
{
    if (num_tours < 0 || n < 1) return fail(TSP2OPT_E_ARG, "num_tours < 0 or n < 1");
    if (num_tours > 0 && (!points || !tours)) return fail(TSP2OPT_E_ARG, "null points/tours");
    if (n > TSP2OPT_MAX_N) return fail(TSP2OPT_E_UNSUPPORTED, "n > 65535 is not supported");
    if (tours_per_instance < 1 || group_size < 1) return fail(TSP2OPT_E_ARG, "tours_per_instance/group_size < 1");
    if (tours_per_instance % group_size != 0 || num_tours % group_size != 0)
        return fail(TSP2OPT_E_ARG, "group_size must divide tours_per_instance and num_tours");
    if (dtype != TSP2OPT_DTYPE_F32 && dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    if (max_iterations < 0) return fail(TSP2OPT_E_ARG, "max_iterations < 0");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const size_t need = ws_layout(num_tours, group_size).total;
    void* ws = workspace;
    bool own = false;
    if (!ws) {
        HIP_TRY(hipMalloc(&ws, need));
        own = true;
    } else if ((size_t)workspace_bytes < need) {
        return fail(TSP2OPT_E_WORKSPACE, "workspace too small");
    }
    int rc;
    if (max_iterations == 0) {
        // two_opt.py:40-48: the first sweep still runs and, if it finds an improvement, is applied
        // before the cap is tested; a cap of 0 therefore behaves like a cap of 1.
        max_iterations = 1;
    }
    if (dtype == TSP2OPT_DTYPE_F64)
        rc = solve_t<double>(points, tours, iters_out, sweeps_out, lengths_out, status_out, num_tours,
                             n, tours_per_instance, group_size, max_iterations, time_limit_s, dtype,
                             device, ws, stream);
    else
        rc = solve_t<float>(points, tours, iters_out, sweeps_out, lengths_out, status_out, num_tours,
                            n, tours_per_instance, group_size, max_iterations, time_limit_s, dtype,
                            device, ws, stream);
    if (own) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(ws);
    }
    return rc;
}

extern "C" int tsp2opt_tour_lengths(const void* points, const int64_t* tours, double* lengths_out,
                                    int64_t num_tours, int32_t n, int32_t tours_per_instance,
                                    int32_t dtype, int32_t device, void* hip_stream)
{
    if (num_tours < 0 || n < 1 || tours_per_instance < 1) return fail(TSP2OPT_E_ARG, "bad size");
    if (num_tours == 0) return 0;
    if (!points || !tours || !lengths_out) return fail(TSP2OPT_E_ARG, "null pointer");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const unsigned grid = (unsigned)((num_tours + 3) / 4);
    if (dtype == TSP2OPT_DTYPE_F64)
        hipLaunchKernelGGL(k_lengths<double>, dim3(grid), dim3(256), 0, stream,
                           static_cast<const double*>(points), tours, lengths_out, num_tours, n,
                           tours_per_instance);
    else if (dtype == TSP2OPT_DTYPE_F32)
        hipLaunchKernelGGL(k_lengths<float>, dim3(grid), dim3(256), 0, stream,
                           static_cast<const float*>(points), tours, lengths_out, num_tours, n,
                           tours_per_instance);
    else
        return fail(TSP2OPT_E_ARG, "bad dtype");
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int tsp2opt_evaluate(const void* points, const int64_t* tours, double* lengths_out,
                                int32_t* valid_out, int64_t* best_index_out, double* best_length_out,
                                int64_t num_tours, int32_t n, int32_t tours_per_instance,
                                int32_t dtype, int32_t device, void* hip_stream)
{
    if (num_tours < 0 || n < 1 || n > TSP2OPT_MAX_N || tours_per_instance < 1)
        return fail(TSP2OPT_E_ARG, "bad size");
    if (num_tours % tours_per_instance != 0) return fail(TSP2OPT_E_ARG, "num_tours % tours_per_instance != 0");
    if (num_tours == 0) return 0;
    if (!points || !tours || !lengths_out) return fail(TSP2OPT_E_ARG, "null pointer");
    if (dtype != TSP2OPT_DTYPE_F32 && dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const int words = (n + 31) / 32;
    const int wpb = words * 4 * 4 <= 48 * 1024 ? 4 : 1;             // waves (tours) per block
    const unsigned grid = (unsigned)((num_tours + wpb - 1) / wpb);
    const size_t lds = (size_t)wpb * words * 4;
    if (dtype == TSP2OPT_DTYPE_F64)
        hipLaunchKernelGGL(k_evaluate<double>, dim3(grid), dim3(64 * wpb), lds, stream,
                           static_cast<const double*>(points), tours, lengths_out, valid_out, num_tours,
                           n, tours_per_instance, words);
    else
        hipLaunchKernelGGL(k_evaluate<float>, dim3(grid), dim3(64 * wpb), lds, stream,
                           static_cast<const float*>(points), tours, lengths_out, valid_out, num_tours,
                           n, tours_per_instance, words);
    HIP_TRY(hipGetLastError());
    if (best_index_out || best_length_out) {
        const int64_t inst = num_tours / tours_per_instance;
        hipLaunchKernelGGL(k_best_of, dim3((unsigned)((inst + 255) / 256)), dim3(256), 0, stream,
                           lengths_out, best_index_out, best_length_out, inst, tours_per_instance);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int tsp2opt_sqrt_check(const void* in, void* out, int64_t count, int32_t dtype,
                                  int32_t device, void* hip_stream)
{
    if (count < 0 || (count > 0 && (!in || !out))) return fail(TSP2OPT_E_ARG, "bad argument");
    if (count == 0) return 0;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const unsigned grid = (unsigned)((count + 255) / 256);
    if (dtype == TSP2OPT_DTYPE_F64)
        hipLaunchKernelGGL(k_sqrt_check<double>, dim3(grid), dim3(256), 0, stream,
                           static_cast<const double*>(in), static_cast<double*>(out), count);
    else if (dtype == TSP2OPT_DTYPE_F32)
        hipLaunchKernelGGL(k_sqrt_check<float>, dim3(grid), dim3(256), 0, stream,
                           static_cast<const float*>(in), static_cast<float*>(out), count);
    else
        return fail(TSP2OPT_E_ARG, "bad dtype");
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int tsp2opt_profile_enable(int32_t on)
{
    g_profile = on ? 1 : 0;
    return 0;
}

extern "C" int tsp2opt_profile_last_ms(int32_t device, double* main_ms, double* forced_ms)
{
    if (device < 0 || device >= kMaxDev || !main_ms) return fail(TSP2OPT_E_ARG, "bad argument");
    ProfSlot& s = g_prof[device];
    if (!s.have_main) return fail(TSP2OPT_E_ARG, "no profiled call on this device");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    float ms = 0.f;
    HIP_TRY(hipEventSynchronize(s.ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms, s.ev[0], s.ev[1]));
    *main_ms = ms;
    if (forced_ms) {
        *forced_ms = 0.0;
        if (s.have_forced) {
            HIP_TRY(hipEventSynchronize(s.ev[3]));
            HIP_TRY(hipEventElapsedTime(&ms, s.ev[2], s.ev[3]));
            *forced_ms = ms;
        }
    }
    return 0;
}

// Diagnostic: with on != 0, wave 0 lane 0 of every workgroup accumulates shader-clock cycles per
// phase (filter sweep, argmin selection incl. exact re-evaluation, reversal, staging, whole tour)
// into workspace bytes [64,104); read them back with tsp2opt_debug_read_phase_cycles after a sync.
extern "C" int tsp2opt_debug_phase_cycles(int32_t on)
{
    g_stamps = on ? 1 : 0;
    return 0;
}

// Diagnostic: sweeps > 0 makes tsp2opt_solve run only `sweeps` filter sweeps per tour (no argmin
// selection, no reversal; tours come back unchanged, iters_out = sweeps).  Measures the hot loop alone.
extern "C" int tsp2opt_debug_sweep_only(int32_t sweeps)
{
    g_dbg_sweeps = sweeps > 0 ? sweeps : 0;
    return 0;
}

// Test/diagnostic knob: -1 automatic choice between one workgroup per tour and a cluster of
// workgroups per tour, 0 never cluster, k >= 2 use up to k workgroups per tour whenever possible.
extern "C" int tsp2opt_debug_cluster_mode(int32_t mode)
{
    g_cluster_mode = mode;
    return 0;
}

extern "C" int tsp2opt_debug_cluster_inc(int32_t on)
{
    g_cluster_inc = on ? 1 : 0;
    return 0;
}

extern "C" int tsp2opt_debug_tune(int32_t what, int32_t value)
{
    if (what == 0) g_force_nt = value;
    else if (what == 1) g_lcap16 = value < 1 ? 1 : value > 16 ? 16 : value;
    else if (what == 2) g_inc_min_tiles = value;
    else if (what == 3) g_cluster_variant = value;
    else if (what == 4) g_lcap16s = value < 1 ? 1 : value > 16 ? 16 : value;
    else return fail(TSP2OPT_E_ARG, "unknown tuning knob");
    return 0;
}

extern "C" int tsp2opt_cluster_plan(int32_t n, int32_t dtype, int64_t num_tours, int32_t cus, int32_t* out12)
{
    if (n < 1 || n > TSP2OPT_MAX_N || num_tours < 1 || cus < 1 || !out12) return fail(TSP2OPT_E_ARG, "bad argument");
    if (dtype != TSP2OPT_DTYPE_F32 && dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    const Plan pl = make_plan(n, dtype);
    for (int k = 0; k < 12; ++k) out12[k] = 0;
    if (g_cluster_mode == 0) return 0;
    const ClusterChoice cc = choose_cluster(n, dtype, num_tours, cus, pl);
    if (cc.kind == 1) {
        const CPlan& cp = cc.band;
        out12[0] = cp.g.CS; out12[1] = cp.g.CR; out12[2] = cp.g.CD; out12[3] = cp.g.nt;
        out12[4] = cp.g.GPw; out12[5] = cp.g.NSw; out12[6] = cp.g.MS; out12[7] = (int32_t)cp.lds;
        out12[8] = cp.g.inc; out12[9] = cp.g.K; out12[10] = cp.g.NG; out12[11] = cp.g.xlds;
    } else if (cc.kind == 2) {
        const Geo& g = cc.fw.g;
        out12[0] = g.CS; out12[1] = 0; out12[2] = g.CS; out12[3] = g.nt;
        out12[4] = g.GPl; out12[5] = g.NS; out12[6] = g.MS; out12[7] = (int32_t)cc.fw.lds;
        out12[8] = g.inc; out12[9] = 2; out12[10] = g.MS / 2; out12[11] = g.xlds;
    }
    return 0;
}

extern "C" int tsp2opt_plan(int32_t n, int32_t dtype, int32_t device, int32_t* out7)
{
    if (n < 1 || n > TSP2OPT_MAX_N || !out7) return fail(TSP2OPT_E_ARG, "bad argument");
    if (dtype != TSP2OPT_DTYPE_F32 && dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    const Plan pl = make_plan(n, dtype);
    out7[0] = pl.g.nt; out7[1] = pl.g.qclass; out7[2] = (int32_t)pl.lds;
    out7[3] = pl.g.NS; out7[4] = pl.g.G; out7[5] = pl.g.MS; out7[6] = 0;
    if (device >= 0) {
        DeviceGuard guard(device);
        if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
        Params p{};
        p.g = pl.g;
        int bpc = 0;
        const int rc = dtype == TSP2OPT_DTYPE_F64 ? dispatch_main<double>(p, pl, device, nullptr, &bpc)
                                                  : dispatch_main<float>(p, pl, device, nullptr, &bpc);
        if (rc) return rc;
        out7[6] = bpc;
    }
    return 0;
}
