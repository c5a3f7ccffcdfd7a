// ml4tspbench_amd/csrc/greedy.hip -- greedy edge-merge decoder on gfx950
//
// Replaces (reference, read-only): search/greedy_search.py:19-74 (tsp_greedy) and
// search/cython_merge/sources/cython_merge.pyx:19-120 (merge_cython): symmetrise the heat map,
// order ALL N^2 edges by -A_ij/|v_i - v_j| (a full argsort of N^2 doubles per heat map on one CPU
// core), insert edges in that order unless they touch an interior node or close a sub-tour, close
// the path, walk it from node 0.
//
// The sort is never materialised here.  "Process the sorted list, skipping invalid edges" is the
// same as "repeatedly take the best still-valid edge", and validity only ever decays (a node that
// became interior stays interior, two nodes of one fragment stay in one fragment).  So every node
// keeps one candidate -- its best valid partner -- and the edges whose turn in the sorted scan nothing
// can disturb any more are inserted together, many per round (see k_greedy): a few tens of rounds of
// parallel work instead of an N^2 log N sort or 3N serial steps.
//
// General path, two kernels.  k_greedy_prepare (fully parallel) turns every heat map into the matrix of
// fp32 filter keys (below), S = A + A^T formed on the fly through a tiled transpose so that A and A^T are
// both read coalesced, and extracts every node's (up to) kTopK best partners as an exact prefix of the
// row's (key, index) order.  k_greedy runs the merge rounds, one workgroup per heat map with 12 B of LDS
// state per node: each node's candidate (filter key, partner), a cursor into its short list, degree, the
// other endpoint of each fragment, and the queue of rows to rescan.  An invalidated candidate is replaced by
// the next still-valid entry of the node's list (validity only decays, so skipped entries never come back)
// and by a rescan of the node's key row once the list is used up.  Heat maps of <= 128 nodes take
// k_greedy_small instead: one kernel with the filter keys of all pairs in LDS.
//
// Arithmetic follows the reference: S is summed in the heat map's dtype (numpy adds float32 arrays
// in float32, greedy_search.py:36-38), keys are -double(S)/sqrt(dx*dx+dy*dy) in double
// (cython_merge.pyx:23,37) -- wherever two keys are close enough for the difference to matter; everything
// else is decided on fp32 approximations with a proven error bound (see "fp32 filter keys").  Where two
// different edges have exactly equal keys the reference's order is numpy's unstable argsort order; here
// (and in the oracle) the lower flat index wins.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "tsp2opt.h"
#include "host_util.hpp"
#include "wave_reduce.hpp"
#include "greedy_walk.hpp"

#pragma clang fp contract(off)

namespace t2o {

// ---- fp32 filter keys ------------------------------------------------------------------------------------
// Every pair's key is kept as an fp32 APPROXIMATION: the sum S in the heat map's dtype exactly as the
// reference forms it, times the hardware reciprocal square root of the squared distance formed from the
// double coordinate differences rounded to fp32 -- relative error below 2^-21, a fraction of the fp64 square
// root and division it replaces.  The approximations only filter:
// wherever a decision depends on two keys that are closer than kGreedyTol = 2^-18 relative (eight times the
// error bound) the exact double keys are recomputed from the heat map (exact_key: the arithmetic of the
// reference, cython_merge.pyx:23,37) and compared as (key, index).  Pairs whose fp32 evaluation leaves the
// normal range (coincident points, overflow) store the fp32 rounding of the exact key, which is monotone.
constexpr float kGreedyTol = 3.814697265625e-06f;        // 2^-18

// the largest filter key that may still precede or tie one whose filter key is m
__device__ __forceinline__ float amb_hi(float m)
{
    return fabsf(m) <= 3.0e38f ? m + fabsf(m) * kGreedyTol : m;
}

template <typename H>
__device__ __forceinline__ float filter_key(H s, double dx, double dy)
{
    const float fx = (float)dx, fy = (float)dy;
    const float d2 = fx * fx + fy * fy;
    float k = -(float)s * __builtin_amdgcn_rsqf(d2);    // v_rsq_f32: 1 ulp on normal inputs
    bool odd = !(d2 > 1.0e-30f) || !(fabsf(k) <= 3.0e38f);
    if constexpr (sizeof(H) == 8) odd = odd || !(fabsf((float)s) <= 3.0e38f);
    if (odd) {
        double ke = -(double)s / sqrt(dx * dx + dy * dy);
        ke = ke != ke ? INFINITY : ke;                  // NaN sorts last (numpy)
        k = (float)ke;
        if (k == 0.f && ke != 0.0) k = ke < 0.0 ? -1.0e-45f : 1.0e-45f;    // underflow: keep "zero" exact
    }
    if (k == 0.f && s != (H)0) k = s > (H)0 ? -1.0e-45f : 1.0e-45f;
    return k;                                           // k == 0  <=>  the exact key is 0 (S == 0)
}

template <typename H>
__device__ __forceinline__ double exact_key(const H* __restrict__ Ab, const double* __restrict__ P, int n, int i, int j)
{
    const H s = Ab[(size_t)i * n + j] + Ab[(size_t)j * n + i];
    const double dx = P[2 * i] - P[2 * j], dy = P[2 * i + 1] - P[2 * j + 1];
    const double k = -(double)s / sqrt(dx * dx + dy * dy);
    return k != k ? INFINITY : k;
}

// A row's best partners in exact (key, j) order, from the two best entries (by filter key) every lane has
// seen among its own columns (j = lane mod 64): lkey (filter key, NaN = no entry) and lj (0xffff = no entry).
// Each round takes the best head over the wave if it is SEPARATED: no other head, and not the owner's second
// entry, lies within the margin -- then it exactly precedes everything that is left in the row (a lane's
// untracked entries are not better than its second).  The extraction stops at the first head that is not,
// and where a lane has popped both its entries (its next best is unknown): the list is then shorter than
// `rounds`, possibly empty, but always an exact prefix of the row's order; the merge loop rescans the row
// once a list is used up.
constexpr int kTopK = 4;

__device__ __forceinline__ void row_topk(float k0, unsigned j0, float k1, unsigned j1, int lane, int rounds,
                                         float* __restrict__ lkey, uint16_t* __restrict__ lj)
{
    int popped = 0;
    int r = 0;
#pragma unroll 1
    for (; r < rounds; ++r) {
        if (__ballot(popped == 2)) break;               // somebody's next best is unknown
        const bool has = j0 != 0xffffu;
        const float hf = has ? k0 : INFINITY;
        const float m = wave_min_ordered(hf);           // (hf is a filter key or +inf, never NaN)
        const float hi = amb_hi(m);
        const bool in = has && hf <= hi;
        const unsigned long long bal = __ballot(in);
        if (__popcll(bal) != 1) break;                  // nothing left, or heads too close to call
        if (__ballot(in && j1 != 0xffffu && k1 <= hi)) break;   // the owner's second entry is too close
        const int src = __ffsll((long long)bal) - 1;
        if (lane == 0) {
            lkey[r] = m;
            lj[r] = (uint16_t)__builtin_amdgcn_readlane((int)j0, src);
        }
        if (lane == src) {                              // the owner pops its head
            k0 = k1; j0 = j1;
            k1 = INFINITY; j1 = 0xffffu;
            ++popped;
        }
    }
    if (lane >= r && lane < kTopK) {                    // unused entries
        lkey[lane] = __builtin_nanf("");
        lj[lane] = 0xffffu;
    }
}

// Kf[b][i][j] = filter key of edge (i, j) (+inf on the diagonal and for NaN, which numpy sorts last).
// Block (64, 8): a band of 32 rows (4 per thread), column tiles of 64; the transposed operand goes
// through LDS.
template <typename H>
__global__ void __launch_bounds__(512) k_greedy_prepare(const H* __restrict__ A, const double* __restrict__ points,
                                                        float* __restrict__ Kd, float* __restrict__ lkey,
                                                        uint16_t* __restrict__ lj, int n, int maps_per_instance,
                                                        int64_t map0, int64_t maps_here, int topk_rounds)
{
    __shared__ H tile[64][33];                          // tile[c][r] = A[c0 + c][i0 + r]
    // Workgroups are dealt to the 8 XCDs (each with its own L2) round-robin in launch order: block id % 8
    // selects the XCD, so all row bands of a heat map are given ids of one residue and follow each other there.
    // The map (1 MB at n = 500) is then read from HBM once and the transposed tiles of the later bands hit that
    // XCD's L2 (with bands spread over the XCDs the misaligned 128-byte column segments were fetched 2.8 times).
    const int bands = (n + 31) / 32;
    const unsigned lin = blockIdx.x, kq = lin >> 3;
    const int64_t ml = (int64_t)(kq / (unsigned)bands) * 8 + (lin & 7u);
    if (ml >= maps_here) return;
    const int64_t b = map0 + ml;
    const int x = threadIdx.x, y = threadIdx.y, t = y * 64 + x;
    const int i0 = (int)(kq % (unsigned)bands) * 32;
    const H* Ab = A + (size_t)b * n * n;
    float* Kb = Kd + (size_t)b * n * n;
    const double* P = points + (size_t)(b / maps_per_instance) * n * 2;
    double rx[4], ry[4];
    float b1k[4], b2k[4];                               // b1/b2: this lane's two best (filter key, j) of the row
    unsigned b1j[4], b2j[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = i0 + y + 8 * rr;
        rx[rr] = i < n ? P[2 * i] : 0.0;
        ry[rr] = i < n ? P[2 * i + 1] : 0.0;
        b1k[rr] = b2k[rr] = INFINITY;
        b1j[rr] = b2j[rr] = 0xffffu;
    }
    // software pipeline: the global loads of tile c0 + 64 (both operands) are in flight while tile c0
    // is computed
    // (indices are clamped instead of predicated: what a lane loads for a row or column past n is never used)
    H a[4], tr[4];
    double pcx, pcy;                                    // the tile's column coordinates travel with it
    const H* arow[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) arow[rr] = Ab + (size_t)min(i0 + y + 8 * rr, n - 1) * n;
    const int icl = min(i0 + (t & 31), n - 1);
    auto load_tile = [&](int c0) {
        const int j = min(c0 + x, n - 1);
        pcx = P[2 * j];
        pcy = P[2 * j + 1];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a[rr] = arow[rr][j];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jr = min(c0 + (t >> 5) + 16 * k, n - 1);
            tr[k] = Ab[(size_t)jr * n + icl];
        }
    };
    load_tile(0);
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int j = c0 + x;
        const double cx = pcx, cy = pcy;
        H ac[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) ac[rr] = a[rr];
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[(t >> 5) + 16 * k][t & 31] = tr[k];
        __syncthreads();
        if (c0 + 64 < n) load_tile(c0 + 64);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = y + 8 * rr, i = i0 + r;
            if (i < n && j < n) {
                float k = INFINITY;
                if (j != i) {
                    const H s = ac[rr] + tile[x][r];
                    k = filter_key(s, rx[rr] - cx, ry[rr] - cy);
                    // a lane meets its columns in increasing order, so (k, j) precedes a tracked entry exactly
                    // when k is smaller, or the slot is empty (key +inf, index 0xffff: every index precedes
                    // it); selects instead of branches: a lane sees n / 64 entries per row and most of the
                    // first ones enter
                    const bool lt1 = k < b1k[rr] || b1j[rr] == 0xffffu;
                    const bool lt2 = k < b2k[rr] || b2j[rr] == 0xffffu;
                    b2k[rr] = lt1 ? b1k[rr] : lt2 ? k : b2k[rr];
                    b2j[rr] = lt1 ? b1j[rr] : lt2 ? (unsigned)j : b2j[rr];
                    b1k[rr] = lt1 ? k : b1k[rr];
                    b1j[rr] = lt1 ? (unsigned)j : b1j[rr];
                }
                Kb[(size_t)i * n + j] = k;
            }
        }
        __syncthreads();
    }
    // candidate lists: the 64 lanes of wave y hold row y + 8 rr
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = i0 + y + 8 * rr;
        if (i < n) {
            const size_t row = (size_t)b * n + i;
            row_topk(b1k[rr], b1j[rr], b2k[rr], b2j[rr], x, topk_rounds, lkey + row * kTopK, lj + row * kTopK);
        }
    }
}

// optional counters (tsp2opt_debug_greedy_cycles), summed over heat maps by thread 0 of every workgroup:
// cycles of [0] list advances, [1] row rescans, [2] choose + apply; [3] merges, [4] rescans, [5] total cycles
// of the merge loop, [6] rounds, [7] rounds that inserted one exactly compared edge
__device__ unsigned long long g_greedy_prof[8];

struct RLds {               // carve of the dynamic LDS for one heat map, 12 B per node
    float* key;             // fp32 rounding of the candidate's key; NaN = none (interior node, nothing valid)
    uint16_t* cj;           // the candidate's partner
    uint16_t* oend;         // for a fragment endpoint: the fragment's other endpoint (itself if alone)
    uint16_t* queue;        // rows to rescan this round
    uint8_t* deg;
    uint8_t* lp;            // cursor into the node's list (kTopK: used up, candidates come from rescans);
                            // bit 7: the node's candidate edge was chosen this round
};

// can node i still take partner j?  (j is a fragment endpoint of another fragment)
__device__ __forceinline__ bool partner_ok(const RLds& L, int i, int oe, int j)
{
    return j != 0xffff && L.deg[j] < 2 && !(j == oe && oe != i);
}

// ---- row rescans: best valid partner of node i from its row of filter keys, one wave per row ----------------
// Every lane reports its best valid entry (k, j), whether it saw a second valid entry within the margin of the
// row minimum is settled afterwards: if exactly one entry of the whole row lies within the margin it is the
// exact best; otherwise the entries within the margin are compared exactly (a second pass over the row for
// rows longer than the registers hold).
template <typename H>
struct RowCtx {
    const float* Kb;        // this heat map's filter keys
    const H* Ab;            // this heat map
    const double* P;        // its instance's points
    int n;
};

__device__ __forceinline__ void rescan_store(const RLds& L, int i, int lane, float fk, unsigned bj)
{
    if (lane == 0) {
        L.key[i] = bj == 0xffffu ? __builtin_nanf("") : fk;
        L.cj[i] = (uint16_t)bj;
    }
}

// exact decision among the valid entries of row i whose filter key is <= hi (all lanes)
template <typename H>
__device__ __forceinline__ void rescan_exact(const RLds& L, const RowCtx<H>& C, int i, int oe, float hi, int lane)
{
    const float* Krow = C.Kb + (size_t)i * C.n;
    double bk = INFINITY;
    unsigned bj = 0xffffu;
    for (int j = lane; j < C.n; j += 64) {
        if (j == i || !partner_ok(L, i, oe, j) || !(Krow[j] <= hi)) continue;
        const double k = exact_key(C.Ab, C.P, C.n, i, j);
        if (k < bk || (k == bk && (unsigned)j < bj)) { bk = k; bj = (unsigned)j; }
    }
    wave_best(bk, bj);
    rescan_store(L, i, lane, bj == 0xffffu ? 0.f : Krow[bj], bj);
}

// lane-local best (b1k, b1j) and the filter key of the lane's second best valid entry (b2k) -> the row's result
template <typename H>
__device__ __forceinline__ void rescan_finish(const RLds& L, const RowCtx<H>& C, int i, int oe, float b1k, unsigned b1j,
                                              float b2k, int lane)
{
    const float m = wave_min_ordered(b1j != 0xffffu ? b1k : INFINITY);
    const float hi = amb_hi(m);
    const unsigned long long e1 = __ballot(b1j != 0xffffu && b1k <= hi);
    const unsigned long long e2 = __ballot(b1j != 0xffffu && b2k <= hi);
    if (!e1) { rescan_store(L, i, lane, 0.f, 0xffffu); return; }       // no valid partner at all
    if (__popcll(e1) == 1 && !e2) {
        const int src = __ffsll((long long)e1) - 1;
        rescan_store(L, i, lane, m, (unsigned)__builtin_amdgcn_readlane((int)b1j, src));
        return;
    }
    if (m == 0.f) {
        // a zero filter key is an exactly zero key (filter_key): the tied entries are all equal, the lowest
        // index wins -- every lane's best is its lowest column with the minimum (thresholded / densified
        // sparse heat maps: most of a row)
        unsigned jm = (b1j != 0xffffu && b1k == 0.f) ? b1j : 0xffffffffu;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) jm = min(jm, (unsigned)__shfl_xor((int)jm, sft));
        rescan_store(L, i, lane, m, jm);
        return;
    }
    rescan_exact(L, C, i, oe, hi, lane);
}

__device__ __forceinline__ void lane_track(float k, unsigned j, float& b1k, unsigned& b1j, float& b2k)
{
    if (b1j == 0xffffu || k < b1k) { b2k = b1j == 0xffffu ? INFINITY : b1k; b1k = k; b1j = j; }
    else if (k < b2k) b2k = k;                          // includes k == b1k: a second entry as good as the best
}

template <typename H>
__device__ __forceinline__ void rescan_row(const RLds& L, const RowCtx<H>& C, int i, int lane)
{
    constexpr int W = 16;
    const int n = C.n;
    const float* Krow = C.Kb + (size_t)i * n;
    const int oe = L.oend[i];
    float b1k = INFINITY, b2k = INFINITY;
    unsigned b1j = 0xffffu;
    for (int j0 = lane; j0 < n; j0 += 64 * W) {
        float k[W];
#pragma unroll
        for (int u = 0; u < W; ++u) k[u] = j0 + 64 * u < n ? Krow[j0 + 64 * u] : INFINITY;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int j = j0 + 64 * u;
            if (j >= n || j == i || !partner_ok(L, i, oe, j)) continue;
            lane_track(k[u], (unsigned)j, b1k, b1j, b2k);
        }
    }
    rescan_finish(L, C, i, oe, b1k, b1j, b2k, lane);
}

// short rows (n <= 64 NL): the keys of 16 / NL queued rows are fetched in one global round trip, then every
// row is reduced -- a rescan is a latency chain, and a workgroup has many of them per round
template <typename H, int NL>
__device__ __forceinline__ void rescan_rows(const RLds& L, const RowCtx<H>& C, int q0, int qn, int lane)
{
    constexpr int R = 16 / NL;
    const int n = C.n;
    int row[R];
    float k[R][NL];
#pragma unroll
    for (int u = 0; u < R; ++u) row[u] = q0 + u < qn ? (int)L.queue[q0 + u] : -1;
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int v = 0; v < NL; ++v)
            k[u][v] = (row[u] >= 0 && lane + 64 * v < n) ? C.Kb[(size_t)row[u] * n + lane + 64 * v] : INFINITY;
#pragma unroll
    for (int u = 0; u < R; ++u) {
        if (row[u] < 0) continue;                       // wave-uniform
        const int i = row[u], oe = L.oend[i];
        float b1k = INFINITY, b2k = INFINITY;
        unsigned b1j = 0xffffu;
#pragma unroll
        for (int v = 0; v < NL; ++v) {
            const int j = lane + 64 * v;
            if (j >= n || j == i || !partner_ok(L, i, oe, j)) continue;
            lane_track(k[u][v], (unsigned)j, b1k, b1j, b2k);
        }
        rescan_finish(L, C, i, oe, b1k, b1j, b2k, lane);
    }
}

__device__ __forceinline__ void link_nodes(const RLds& L, uint16_t* __restrict__ nb, int i, int j)
{
    const int oi = L.oend[i], oj = L.oend[j];
    const int di = L.deg[i], dj = L.deg[j];
    nb[2 * i + di] = (uint16_t)j;               // slot = degree before the merge
    nb[2 * j + dj] = (uint16_t)i;
    L.deg[i] = (uint8_t)(di + 1);
    L.deg[j] = (uint8_t)(dj + 1);
    if (di >= 1) L.key[i] = __builtin_nanf("");
    if (dj >= 1) L.key[j] = __builtin_nanf("");
    L.oend[oi] = (uint16_t)oj;                  // the merged fragment runs from oi to oj
    L.oend[oj] = (uint16_t)oi;
}

// The merge loop, one workgroup per heat map (64 threads for n <= 128, 256 up to 2048 nodes, 1024 above),
// in ROUNDS that insert many edges at once (the same scheme as the sparse decoder, greedy_sparse.hip).
// Every round first makes every open node's candidate its best VALID partner: the next valid entry of
// its short list (a thread per node, lists read where the prepare kernel left them), else a rescan of its
// key row (queued; a wave per queued row).  An edge f = (i, j) is then inserted when
//   * it is the candidate of both i and j -- no valid edge at i or j precedes it -- and
//   * the far ends a, b of the two fragments it joins hold candidates whose filter keys are larger by more
//     than the margin (so their exact keys are strictly larger):
//     nothing can happen at i, j, a or b before f's turn in the sorted scan, so f is valid at its turn, and
//     the only edge whose validity f changes, (a, b), comes after it.
// Edges chosen in one round join disjoint pairs of fragments: a thread each applies them.  The valid edge
// with the globally smallest key always qualifies unless another candidate's filter key lies within the
// margin; a round that finds nothing inserts exactly that edge after comparing those candidates exactly
// (double key, then flat index): the serial rule.  A few tens of rounds replace about 3N serial steps.
// NL: key loads per lane that cover a row (2, 4, 8 for n <= 128, 256, 512), 0 = longer rows
template <typename H, int NL>
__global__ void __launch_bounds__(1024) k_greedy(const float* __restrict__ Kd, const H* __restrict__ A,
                                                 const double* __restrict__ points, const float* __restrict__ lkey,
                                                 const uint16_t* __restrict__ lj0, uint16_t* __restrict__ nbg,
                                                 int64_t* __restrict__ tours, int64_t num_maps, int n, int mpi, int K,
                                                 int prof)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red_f[16];
    __shared__ double red_e[16];
    __shared__ unsigned red_i[16];
    __shared__ int ctl[3];                              // edges chosen this round; any candidate left; queue length
    const int64_t b = blockIdx.x;
    if (b >= num_maps) return;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    const int G = (n + 63) >> 6;
    RLds L;
    size_t off = 0;
    L.key = reinterpret_cast<float*>(smem + off); off += sizeof(float) * (size_t)n;
    L.cj = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.oend = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.queue = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.deg = reinterpret_cast<uint8_t*>(smem + off); off += (size_t)n;
    L.lp = reinterpret_cast<uint8_t*>(smem + off);
    const RowCtx<H> C{Kd + (size_t)b * (size_t)n * (size_t)n, A + (size_t)b * (size_t)n * (size_t)n,
                      points + (size_t)(b / mpi) * (size_t)n * 2, n};
    const float* glk = lkey + (size_t)b * (size_t)n * kTopK;
    const uint16_t* glj = lj0 + (size_t)b * (size_t)n * kTopK;
    uint16_t* nb = nbg + (size_t)b * 2 * (size_t)n;
    for (int i = tid; i < n; i += T) {
        L.oend[i] = (uint16_t)i;
        L.deg[i] = 0;
        nb[2 * i] = nb[2 * i + 1] = 0xffffu;
        const float hk = glk[(size_t)i * kTopK];        // first candidates: the heads of the lists; an empty
        const unsigned hj = glj[(size_t)i * kTopK];     // list (the row's best was too close to call) makes the
        L.lp[i] = hj == 0xffffu ? (uint8_t)kTopK : (uint8_t)0;     // first round rescan the row
        L.key[i] = hj == 0xffffu ? 0.f : hk;
        L.cj[i] = (uint16_t)hj;
    }
    if (tid == 0) ctl[0] = ctl[1] = ctl[2] = 0;
    __syncthreads();

    int merges = 0;
    unsigned long long t_adv = 0, t_resc = 0, t_apply = 0, n_resc = 0, rounds = 0, serial_rounds = 0;
    const unsigned long long t_begin = prof ? clock64() : 0;
    for (int guard = 0; merges < n - 1 && guard < n + 8; ++guard) {
        const unsigned long long t0 = prof ? clock64() : 0;
        // (1a) invalid candidates: the next valid list entry, else queue the row
        if (tid == 0) ctl[0] = ctl[1] = 0;              // last read before the barrier that ended the previous round
        for (int i = tid; i < n; i += T) {
            const float k = L.key[i];
            if (k != k) continue;                       // none (also: interior)
            const int oe = L.oend[i];
            if (partner_ok(L, i, oe, L.cj[i])) continue;
            int p = L.lp[i];
            bool found = false;
            if (p + 1 < K) {
                float lk[kTopK];
                unsigned lj[kTopK];
#pragma unroll
                for (int u = 0; u < kTopK; ++u) { lk[u] = glk[(size_t)i * kTopK + u]; lj[u] = glj[(size_t)i * kTopK + u]; }
#pragma unroll
                for (int u = 1; u < kTopK; ++u) {
                    if (!found && u > p && u < K && partner_ok(L, i, oe, (int)lj[u])) {
                        found = true;
                        L.key[i] = lk[u];
                        L.cj[i] = (uint16_t)lj[u];
                        L.lp[i] = (uint8_t)u;
                    }
                }
            }
            if (!found) {
                L.lp[i] = (uint8_t)kTopK;
                L.queue[atomicAdd(&ctl[2], 1)] = (uint16_t)i;
            }
        }
        __syncthreads();
        const unsigned long long t1 = prof ? clock64() : 0;
        // (1b) queued rows: a wave each
        const int qn = ctl[2];
        if constexpr (NL > 0) {
            for (int q0 = wv * (16 / NL); q0 < qn; q0 += nw * (16 / NL)) rescan_rows<H, NL>(L, C, q0, qn, lane);
        } else {
            for (int q = wv; q < qn; q += nw) rescan_row(L, C, (int)L.queue[q], lane);
        }
        n_resc += (unsigned long long)qn;
        __syncthreads();
        const unsigned long long t2 = prof ? clock64() : 0;
        // (2) choose: the lower endpoint of an edge decides and marks itself
        if (tid == 0) ctl[2] = 0;
        int mine = 0;
        bool any = false;
        for (int i = tid; i < n; i += T) {
            const float k = L.key[i];
            if (k != k) continue;
            any = true;
            const int j = L.cj[i];
            if (j < i || L.cj[j] != i) continue;
            const float kj = L.key[j];
            if (kj != kj) continue;
            const int a = L.oend[i], c = L.oend[j];
            const float hi = amb_hi(k);
            if (a != i && L.key[a] <= hi) continue;     // NaN (no candidate) compares false: passes
            if (c != j && L.key[c] <= hi) continue;
            L.lp[i] |= 0x80u;
            ++mine;
        }
        if (any) ctl[1] = 1;
        if (mine) atomicAdd(&ctl[0], mine);
        __syncthreads();
        const int chosen = ctl[0];
        if (!ctl[1]) break;                             // no candidate anywhere (cannot happen for n >= 2)
        ++rounds;
        if (chosen) {
            // (3) apply: disjoint fragments, one thread per edge
            for (int i = tid; i < n; i += T) {
                const unsigned r = L.lp[i];
                if (r & 0x80u) {
                    L.lp[i] = (uint8_t)(r & 0x7fu);
                    link_nodes(L, nb, i, L.cj[i]);
                }
            }
            merges += chosen;
        } else {
            // different edges within the margin of the smallest filter key: exact keys, then the flat index
            ++serial_rounds;
            float lm = INFINITY;
            for (int i = tid; i < n; i += T) {
                const float k = L.key[i];
                if (k == k) lm = fminf(lm, k);
            }
            lm = wave_min_ordered(lm);
            if (lane == 0) red_f[wv] = lm;
            __syncthreads();
            float m = INFINITY;
            for (int w = 0; w < nw; ++w) m = fminf(m, red_f[w]);
            m = amb_hi(m);
            double bk = INFINITY;
            unsigned bf = 0xffffffffu;
            for (int i = tid; i < n; i += T) {
                if (!(L.key[i] <= m)) continue;
                const unsigned cj = L.cj[i];
                const unsigned f = (unsigned)i * (unsigned)n + cj;
                const double key = exact_key(C.Ab, C.P, n, i, (int)cj);
                if (better(key, f, bk, bf)) { bk = key; bf = f; }
            }
            wave_best(bk, bf);
            if (lane == 0) { red_e[wv] = bk; red_i[wv] = bf; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 0; w < nw; ++w)
                    if (better(red_e[w], red_i[w], bk, bf)) { bk = red_e[w]; bf = red_i[w]; }
                if (bf != 0xffffffffu) {
                    const int i = (int)(bf / (unsigned)n);
                    link_nodes(L, nb, i, (int)(bf - (unsigned)i * (unsigned)n));
                }
            }
            ++merges;
        }
        __syncthreads();
        if (prof) {
            const unsigned long long t3 = clock64();
            t_adv += t1 - t0; t_resc += t2 - t1; t_apply += t3 - t2;
        }
    }
    __syncthreads();
    if (prof && tid == 0) {
        atomicAdd(&g_greedy_prof[0], t_adv);
        atomicAdd(&g_greedy_prof[1], t_resc);
        atomicAdd(&g_greedy_prof[2], t_apply);
        atomicAdd(&g_greedy_prof[3], (unsigned long long)merges);
        atomicAdd(&g_greedy_prof[4], n_resc);
        atomicAdd(&g_greedy_prof[5], clock64() - t_begin);
        atomicAdd(&g_greedy_prof[6], rounds);
        atomicAdd(&g_greedy_prof[7], serial_rounds);
    }
    // closing edge between the two remaining endpoints (cython_merge.pyx:99-102), then the walk of
    // greedy_search.py:65-73 (from node 0, never back to the previous node, larger index first) over an LDS
    // copy of the neighbour table: a parallel list ranking (greedy_walk.hpp), or -- should the merge ever end
    // without a Hamiltonian path -- the serial walk, step by step like the reference
    if (tid < 64) {
        int a = -1, c = -1;
        for (int g = 0; g < G; ++g) {
            const int k = (g << 6) + lane;
            unsigned long long bal = __ballot(k < n && L.deg[k] < 2);
            while (bal) {
                const int k2 = (g << 6) + __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                if (a < 0) a = k2; else c = k2;
            }
        }
        if (lane == 0) { ctl[0] = a; ctl[1] = c; }
    }
    __syncthreads();                                        // (also: the merge loop's last writes of nb)
    const int a = ctl[0], c = ctl[1];
    unsigned* nbl = reinterpret_cast<unsigned*>(L.key);    // the candidates are dead by now
    bool whole = true;
    for (int k = tid; k < n; k += T) {
        unsigned lo = nb[2 * k], hi = nb[2 * k + 1];
        if (a >= 0 && c >= 0) {
            if (k == a) { if (lo == 0xffffu) lo = (unsigned)c; else hi = (unsigned)c; }
            if (k == c) { if (lo == 0xffffu) lo = (unsigned)a; else hi = (unsigned)a; }
        }
        nbl[k] = lo | (hi << 16);
        whole = whole && lo != 0xffffu && hi != 0xffffu && lo != hi;
    }
    int64_t* t = tours + (size_t)b * (size_t)(n + 1);
    if (__syncthreads_and(whole ? 1 : 0) && merges == n - 1 && walk_parallel_fits(n, T)) {      // (uniform)
        walk_parallel(nbl, reinterpret_cast<unsigned*>(L.cj), n, tid, T, t);    // cj .. lp: 8 B per node, all dead
        return;
    }
    if (lane < 2 && tid < 64) {
        // lane 0 walks from node 0 towards its larger neighbour and fills positions 1 .. n/2, lane 1 walks the
        // other way round and fills n-1 .. n/2+1
        const int h = n >> 1, steps = lane == 0 ? h : n - 1 - h;
        int prev = -1, cur = 0;
        if (lane == 0) { t[0] = 0; t[n] = n == 2 ? -1 : 0; }   // n = 2: the reference's walk has no step left (it raises); -1 like the oracle
        for (int s2 = 1; s2 <= steps; ++s2) {
            const unsigned w = nbl[cur];
            const int n0 = (int)(w & 0xffffu), n1 = (int)(w >> 16);
            int best = -1;
            if (n0 != 0xffff && n0 != prev) best = n0;
            if (n1 != 0xffff && n1 != prev && (s2 == 1 && lane == 1 ? (best < 0 || n1 < best) : n1 > best)) best = n1;
            t[lane == 0 ? s2 : n - s2] = best;
            prev = cur;
            cur = best < 0 ? 0 : best;
        }
    }
}

// ---- small heat maps (n <= kGreedySmallN): one kernel, everything on chip -------------------------------
// The filter keys of a heat map's n (n - 1) / 2 pairs fit the LDS (20 KB at n = 100), so no key matrix goes
// to memory: the workgroup reads its heat map once, keeps the filter key of every pair, and runs the same
// rounds with row rescans out of LDS (margins and exact comparisons as described at kGreedyTol).
constexpr int kGreedySmallN = 128;

struct SLds {
    float* Kf;              // the n (n - 1) / 2 keys of the pairs i < j, row-major triangle (see kf_at)
    double* px;
    double* py;
    float* key;
    unsigned* nbl;          // two 16-bit neighbour slots per node
    uint16_t* cj;
    uint16_t* oend;
    uint16_t* queue;
    uint8_t* deg;
    uint8_t* flag;
    int n;
};

// key of the pair {i, j}, i != j
__device__ __forceinline__ float& kf_at(const SLds& L, int i, int j)
{
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    return L.Kf[lo * (2 * L.n - lo - 1) / 2 + (hi - lo - 1)];
}

template <typename H>
__device__ __forceinline__ double small_exact_key(const H* __restrict__ Ab, const SLds& L, int n, int i, int j)
{
    const H s = Ab[(size_t)i * n + j] + Ab[(size_t)j * n + i];
    const double dx = L.px[i] - L.px[j], dy = L.py[i] - L.py[j];
    const double k = -(double)s / sqrt(dx * dx + dy * dy);
    return k != k ? INFINITY : k;
}

__device__ __forceinline__ bool small_ok(const SLds& L, int i, int oe, int j)
{
    return L.deg[j] < 2 && !(j == oe && oe != i);
}

// best valid partner of nodes ia and ib (ib < 0: none), one wave, the two latency chains interleaved: fp32
// minimum over the node's keys, exact comparison among the entries within the margin
template <typename H>
__device__ __forceinline__ void small_rescan2(const SLds& L, const H* __restrict__ Ab, int n, int ia, int ib, int lane)
{
    const int row[2] = {ia, ib};
    float k[2][2];
    bool v[2][2];
    float hi[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = row[u] < 0 ? 0 : row[u];
        const int oe = L.oend[i];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int j = lane + 64 * w;
            v[u][w] = row[u] >= 0 && j < n && j != i && small_ok(L, i, oe, j);
            k[u][w] = v[u][w] ? kf_at(L, i, j) : INFINITY;
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) hi[u] = amb_hi(wave_min(fminf(k[u][0], k[u][1])));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (row[u] < 0) continue;                       // wave-uniform
        const int i = row[u];
        const bool a0 = v[u][0] && k[u][0] <= hi[u], a1 = v[u][1] && k[u][1] <= hi[u];
        const unsigned long long e0 = __ballot(a0), e1 = __ballot(a1);
        const int hits = __popcll(e0) + __popcll(e1);
        unsigned bj = 0xffffu;
        if (hits == 1) {
            bj = e0 ? (unsigned)(__ffsll((long long)e0) - 1) : (unsigned)(64 + __ffsll((long long)e1) - 1);
        } else if (hits > 1) {
            double bk = INFINITY;
            if (a0) { bk = small_exact_key(Ab, L, n, i, lane); bj = (unsigned)lane; }
            if (a1) {
                const double ke = small_exact_key(Ab, L, n, i, lane + 64);
                if (better(ke, (unsigned)(lane + 64), bk, bj)) { bk = ke; bj = (unsigned)(lane + 64); }
            }
            wave_best(bk, bj);
        }
        if (lane == 0) {
            L.key[i] = hits ? kf_at(L, i, (int)bj) : __builtin_nanf("");
            L.cj[i] = (uint16_t)bj;
        }
    }
}

__device__ __forceinline__ void small_link(const SLds& L, int i, int j)
{
    const int oi = L.oend[i], oj = L.oend[j];
    const int di = L.deg[i], dj = L.deg[j];
    uint16_t* nb = reinterpret_cast<uint16_t*>(L.nbl);
    nb[2 * i + di] = (uint16_t)j;
    nb[2 * j + dj] = (uint16_t)i;
    L.deg[i] = (uint8_t)(di + 1);
    L.deg[j] = (uint8_t)(dj + 1);
    if (di >= 1) L.key[i] = __builtin_nanf("");
    if (dj >= 1) L.key[j] = __builtin_nanf("");
    L.oend[oi] = (uint16_t)oj;
    L.oend[oj] = (uint16_t)oi;
}

inline size_t greedy_small_lds(int n) { return (size_t)n * (size_t)(n - 1) * 2 + (size_t)n * 32 + 64; }

template <typename H>
__global__ void __launch_bounds__(256) k_greedy_small(const H* __restrict__ A, const double* __restrict__ points,
                                                      int64_t* __restrict__ tours, int64_t num_maps, int n, int mpi,
                                                      int prof)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red_f[4];
    __shared__ double red_e[4];
    __shared__ unsigned red_i[4];
    __shared__ int ctl[3];
    const int64_t b = blockIdx.x;
    if (b >= num_maps) return;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    SLds L;
    L.n = n;
    size_t off = 0;
    L.px = reinterpret_cast<double*>(smem + off); off += sizeof(double) * (size_t)n;
    L.py = reinterpret_cast<double*>(smem + off); off += sizeof(double) * (size_t)n;
    L.Kf = reinterpret_cast<float*>(smem + off); off += sizeof(float) * ((size_t)n * (size_t)(n - 1) / 2);
    L.key = reinterpret_cast<float*>(smem + off); off += sizeof(float) * (size_t)n;
    L.nbl = reinterpret_cast<unsigned*>(smem + off); off += sizeof(unsigned) * (size_t)n;
    L.cj = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.oend = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.queue = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.deg = reinterpret_cast<uint8_t*>(smem + off); off += (size_t)n;
    L.flag = reinterpret_cast<uint8_t*>(smem + off);
    const H* Ab = A + (size_t)b * (size_t)n * (size_t)n;
    const double* P = points + (size_t)(b / mpi) * (size_t)n * 2;
    const unsigned long long t_begin = prof ? clock64() : 0;
    for (int i = tid; i < n; i += T) {
        L.px[i] = P[2 * i];
        L.py[i] = P[2 * i + 1];
        L.oend[i] = (uint16_t)i;
        L.deg[i] = 0;
        L.flag[i] = 0;
        L.nbl[i] = 0xffffffffu;
        L.key[i] = 0.f;
        L.queue[i] = (uint16_t)i;                        // round 0: every row is scanned
    }
    if (tid == 0) { ctl[0] = ctl[1] = 0; ctl[2] = n; }
    __syncthreads();
    // keys of all pairs i < j, stored both ways.  Rows r and n-2-r of the triangle hold n-1-r and r+1 pairs:
    // together one line of n cells, so cell f of an (n/2) x n rectangle decodes with one division
    const int lines = n >> 1, cells = lines * n;
    const int mid = (n & 1) ? -1 : (n - 2) >> 1;        // n even: the middle row pairs with itself, once
    for (int f0 = tid; f0 < cells; f0 += 2 * T) {
        int pi[2], pj[2];
        H sa[2], sb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = f0 + u * T;
            const int r = f / n, c = f - r * n;
            int i = r, j = r + 1 + c;
            if (c >= n - 1 - r) { i = n - 2 - r; j = i + 1 + (c - (n - 1 - r)); }
            const bool in = f < cells && !(r == mid && c >= n - 1 - r);
            pi[u] = in ? i : -1;
            pj[u] = j;
            sa[u] = in ? Ab[(size_t)i * n + j] : (H)0;
            sb[u] = in ? Ab[(size_t)j * n + i] : (H)0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (pi[u] < 0) continue;
            const int i = pi[u], j = pj[u];
            const H s2 = sa[u] + sb[u];                 // coo(i,j) + coo(j,i) in the heat map's dtype
            const double dx = L.px[i] - L.px[j], dy = L.py[i] - L.py[j];
            const float k = filter_key(s2, dx, dy);
            kf_at(L, i, j) = k;
        }
    }
    __syncthreads();

    int merges = 0;
    unsigned long long n_resc = 0, rounds = 0, serial_rounds = 0, t_adv = 0, t_resc = 0, t_apply = 0;
    const unsigned long long t_fill = prof ? clock64() - t_begin : 0;
    for (int guard = 0; merges < n - 1 && guard < n + 8; ++guard) {
        // (1a) nodes whose candidate is no longer valid queue their row (round 0: all rows are queued)
        const unsigned long long t0 = prof ? clock64() : 0;
        if (tid == 0) ctl[0] = ctl[1] = 0;
        if (guard > 0) {
            for (int i = tid; i < n; i += T) {
                const float k = L.key[i];
                if (k != k) continue;
                if (!small_ok(L, i, L.oend[i], L.cj[i])) L.queue[atomicAdd(&ctl[2], 1)] = (uint16_t)i;
            }
        }
        __syncthreads();
        // (1b) a wave per queued row
        const unsigned long long t1 = prof ? clock64() : 0;
        const int qn = ctl[2];
        for (int q = wv; q < qn; q += 2 * nw)
            small_rescan2(L, Ab, n, (int)L.queue[q], q + nw < qn ? (int)L.queue[q + nw] : -1, lane);
        n_resc += (unsigned long long)qn;
        __syncthreads();
        // (2) choose
        const unsigned long long t2 = prof ? clock64() : 0;
        if (tid == 0) ctl[2] = 0;
        int mine = 0;
        bool any = false;
        for (int i = tid; i < n; i += T) {
            const float k = L.key[i];
            if (k != k) continue;
            any = true;
            const int j = L.cj[i];
            if (j < i || L.cj[j] != i) continue;
            const float kj = L.key[j];
            if (kj != kj) continue;
            const int a = L.oend[i], c = L.oend[j];
            const float hi = amb_hi(k);
            if (a != i && L.key[a] <= hi) continue;     // NaN (no candidate) compares false: passes
            if (c != j && L.key[c] <= hi) continue;
            L.flag[i] = 1;
            ++mine;
        }
        if (any) ctl[1] = 1;
        if (mine) atomicAdd(&ctl[0], mine);
        __syncthreads();
        const int chosen = ctl[0];
        if (!ctl[1]) break;
        ++rounds;
        if (chosen) {
            for (int i = tid; i < n; i += T) {
                if (L.flag[i]) {
                    L.flag[i] = 0;
                    small_link(L, i, L.cj[i]);
                }
            }
            merges += chosen;
        } else {
            ++serial_rounds;
            float lm = INFINITY;
            for (int i = tid; i < n; i += T) {
                const float k = L.key[i];
                if (k == k) lm = fminf(lm, k);
            }
            lm = wave_min_ordered(lm);
            if (lane == 0) red_f[wv] = lm;
            __syncthreads();
            float m = INFINITY;
            for (int w = 0; w < nw; ++w) m = fminf(m, red_f[w]);
            m = amb_hi(m);
            double bk = INFINITY;
            unsigned bf = 0xffffffffu;
            for (int i = tid; i < n; i += T) {
                if (!(L.key[i] <= m)) continue;
                const unsigned cj = L.cj[i];
                const unsigned f = (unsigned)i * (unsigned)n + cj;
                const double key = small_exact_key(Ab, L, n, i, (int)cj);
                if (better(key, f, bk, bf)) { bk = key; bf = f; }
            }
            wave_best(bk, bf);
            if (lane == 0) { red_e[wv] = bk; red_i[wv] = bf; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 0; w < nw; ++w)
                    if (better(red_e[w], red_i[w], bk, bf)) { bk = red_e[w]; bf = red_i[w]; }
                if (bf != 0xffffffffu) {
                    const int i = (int)(bf / (unsigned)n);
                    small_link(L, i, (int)(bf - (unsigned)i * (unsigned)n));
                }
            }
            ++merges;
        }
        __syncthreads();
        if (prof) {
            const unsigned long long t3 = clock64();
            t_adv += t1 - t0; t_resc += t2 - t1; t_apply += t3 - t2;
        }
    }
    __syncthreads();
    if (prof && tid == 0) {
        atomicAdd(&g_greedy_prof[0], t_adv + t_fill);       // key fill + validity checks
        atomicAdd(&g_greedy_prof[1], t_resc);
        atomicAdd(&g_greedy_prof[2], t_apply);
        atomicAdd(&g_greedy_prof[3], (unsigned long long)merges);
        atomicAdd(&g_greedy_prof[4], n_resc);
        atomicAdd(&g_greedy_prof[5], clock64() - t_begin);
        atomicAdd(&g_greedy_prof[6], rounds);
        atomicAdd(&g_greedy_prof[7], serial_rounds);
    }
    if (tid >= 64) return;
    // closing edge, then the walk (as in k_greedy)
    int a = -1, c = -1;
    for (int g = 0; g < ((n + 63) >> 6); ++g) {
        const int k = (g << 6) + lane;
        unsigned long long bal = __ballot(k < n && L.deg[k] < 2);
        while (bal) {
            const int k2 = (g << 6) + __ffsll((long long)bal) - 1;
            bal &= bal - 1;
            if (a < 0) a = k2; else c = k2;
        }
    }
    if (lane == 0 && a >= 0 && c >= 0) {
        uint16_t* nb = reinterpret_cast<uint16_t*>(L.nbl);
        nb[2 * a + (nb[2 * a] != 0xffffu)] = (uint16_t)c;
        nb[2 * c + (nb[2 * c] != 0xffffu)] = (uint16_t)a;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 2) {
        int64_t* t = tours + (size_t)b * (size_t)(n + 1);
        const int h = n >> 1, steps = lane == 0 ? h : n - 1 - h;
        int prev = -1, cur = 0;
        if (lane == 0) { t[0] = 0; t[n] = n == 2 ? -1 : 0; }   // n = 2: the reference's walk has no step left (it raises); -1 like the oracle
        for (int s2 = 1; s2 <= steps; ++s2) {
            const unsigned w = L.nbl[cur];
            const int n0 = (int)(w & 0xffffu), n1 = (int)(w >> 16);
            int best = -1;
            if (n0 != 0xffff && n0 != prev) best = n0;
            if (n1 != 0xffff && n1 != prev && (s2 == 1 && lane == 1 ? (best < 0 || n1 < best) : n1 > best)) best = n1;
            t[lane == 0 ? s2 : n - s2] = best;
            prev = cur;
            cur = best < 0 ? 0 : best;
        }
    }
}

// diagnostic switches belong to the calling thread, like the 2-opt knobs (tsp2opt.hip: thread_local Opts)
thread_local int t_greedy_onchip = 1; // tsp2opt_debug_greedy_cycles enable = 2 / 3: small maps on chip or not
constexpr int kGreedyMaxN = 10200;      // partner indices are 16-bit, 12 B of LDS state per node

// entries the prepare kernel extracts per node and the merge loop walks before it rescans a row: the row's
// best only for small n, where tracking and extracting more costs the prepare kernel more than the loop gains
inline int greedy_rounds(int n) { return kTopK; }

inline size_t greedy_lds(int n) { return (size_t)n * 12 + 64; }

// threads of a merge workgroup (one per heat map): 256 keep many maps resident; few maps of more than a few
// hundred nodes get 1024 (more waves for the queued row rescans and the per-node phases; the chip is not full anyway)
inline unsigned merge_threads(int n, int64_t num_maps)
{
    if (n <= 128) return 64;
    if (n > 2048 || (n > 512 && num_maps <= 512)) return 1024;
    return 256;
}

struct GreedyWs {
    size_t off_lkey, off_lj, off_nb, total;     // key matrices at 0
};

inline GreedyWs greedy_ws(int64_t num_maps, int n)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    GreedyWs w{};
    size_t o = up((size_t)num_maps * (size_t)n * (size_t)n * 4);
    w.off_lkey = o; o += up((size_t)num_maps * (size_t)n * kTopK * 4);
    w.off_lj = o; o += up((size_t)num_maps * (size_t)n * kTopK * 2);
    w.off_nb = o; o += up((size_t)num_maps * (size_t)n * 4);
    w.total = o + 256;
    return w;
}

thread_local int t_greedy_prof_on = 0;

template <typename H>
int greedy_t(const void* heat, const double* points, int64_t* tours, int64_t num_maps, int n, int mpi,
             void* ws, hipStream_t stream)
{
    const H* A = static_cast<const H*>(heat);
    if (n <= kGreedySmallN && t_greedy_onchip) {       // everything on chip: no key matrix, no workspace
        const size_t lds = greedy_small_lds(n);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_greedy_small<H>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_greedy_small<H>, dim3((unsigned)num_maps), dim3(n <= 64 ? 128 : 256), lds, stream, A, points,
                           tours, num_maps, n, mpi, t_greedy_prof_on);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const GreedyWs wl = greedy_ws(num_maps, n);
    unsigned char* w8 = static_cast<unsigned char*>(ws);
    float* Kd = reinterpret_cast<float*>(w8);
    float* lkey = reinterpret_cast<float*>(w8 + wl.off_lkey);
    uint16_t* lj = reinterpret_cast<uint16_t*>(w8 + wl.off_lj);
    // prepare in slabs of heat maps (block ids stay below 2^31)
    for (int64_t b0 = 0; b0 < num_maps; b0 += 32768) {
        const int64_t here = std::min<int64_t>(32768, num_maps - b0);
        const dim3 grid((unsigned)(((here + 7) / 8) * 8 * ((n + 31) / 32)), 1, 1);
        hipLaunchKernelGGL((k_greedy_prepare<H>), grid, dim3(64, 8), 0, stream, A, points, Kd, lkey, lj, n, mpi, b0, here,
                           greedy_rounds(n));
        HIP_TRY(hipGetLastError());
    }
    uint16_t* nb = reinterpret_cast<uint16_t*>(w8 + wl.off_nb);
    const size_t lds = greedy_lds(n);
    auto kern = n <= 128 ? k_greedy<H, 2> : n <= 256 ? k_greedy<H, 4> : n <= 512 ? k_greedy<H, 8> : k_greedy<H, 0>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)num_maps), dim3(merge_threads(n, num_maps)), lds, stream, Kd,
                       A, points, lkey, lj, nb, tours, num_maps, n, mpi, greedy_rounds(n), t_greedy_prof_on);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace t2o

using namespace t2o;

extern "C" int64_t tsp2opt_greedy_workspace_bytes(int64_t num_maps, int32_t n, int32_t heat_dtype)
{
    if (num_maps < 0 || n < 2) return fail(TSP2OPT_E_ARG, "bad size");
    if (heat_dtype != TSP2OPT_DTYPE_F32 && heat_dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    return (int64_t)greedy_ws(num_maps, n).total;
}

extern "C" int tsp2opt_greedy_decode(const void* heat, const double* points, int64_t* tours_out,
                                     int64_t num_maps, int32_t n, int32_t maps_per_instance,
                                     int32_t heat_dtype, int32_t device, void* workspace,
                                     int64_t workspace_bytes, void* hip_stream)
{
    if (num_maps < 0 || n < 2 || maps_per_instance < 1) return fail(TSP2OPT_E_ARG, "bad size");
    if (n > kGreedyMaxN) return fail(TSP2OPT_E_UNSUPPORTED, "greedy decode supports n <= 10200");
    if (heat_dtype != TSP2OPT_DTYPE_F32 && heat_dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    if (num_maps == 0) return 0;
    if (!heat || !points || !tours_out || !workspace) return fail(TSP2OPT_E_ARG, "null pointer");
    if (workspace_bytes < tsp2opt_greedy_workspace_bytes(num_maps, n, heat_dtype))
        return fail(TSP2OPT_E_WORKSPACE, "workspace too small");
    if (num_maps > 0x7fffffffll) return fail(TSP2OPT_E_UNSUPPORTED, "too many heat maps");
    if ((reinterpret_cast<uintptr_t>(heat) % (heat_dtype == TSP2OPT_DTYPE_F64 ? 8 : 4)) ||
        (reinterpret_cast<uintptr_t>(points) % 8) || (reinterpret_cast<uintptr_t>(tours_out) % 8))
        return fail(TSP2OPT_E_ARG, "misaligned pointer (arrays must be aligned to their element size)");
    if (reinterpret_cast<uintptr_t>(workspace) % 256)
        return fail(TSP2OPT_E_ARG, "workspace must be 256-byte aligned");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (heat_dtype == TSP2OPT_DTYPE_F64)
        return greedy_t<double>(heat, points, tours_out, num_maps, n, maps_per_instance, workspace, stream);
    return greedy_t<float>(heat, points, tours_out, num_maps, n, maps_per_instance, workspace, stream);
}

// Diagnostic: enable = 1 zeroes and enables the merge-loop phase timers of k_greedy on `device`,
// enable = 0 disables them and copies the 8 counters to out8 (see g_greedy_prof).  Synchronises.
extern "C" int tsp2opt_debug_greedy_cycles(int32_t enable, int32_t device, uint64_t* out8)
{
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    HIP_TRY(hipDeviceSynchronize());
    if (enable == 2 || enable == 3) { t_greedy_onchip = enable == 2; return 0; }
    if (enable) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_greedy_prof), z, sizeof(z)));
        t_greedy_prof_on = 1;
        return 0;
    }
    t_greedy_prof_on = 0;
    if (out8) HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_greedy_prof), 8 * sizeof(unsigned long long)));
    return 0;
}
