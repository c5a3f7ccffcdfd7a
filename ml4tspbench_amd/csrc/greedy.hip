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
// keeps one candidate -- its best valid partner when last looked at -- and each step takes the
// minimum (key, flat index) over the endpoints' candidates: if that edge is still valid it is the
// edge the sorted scan would insert next; if not, that node's row is rescanned.  About 3N steps of
// O(N/64) work per lane instead of an N^2 log N sort.
//
// Two kernels.  k_greedy_prepare (fully parallel) turns every heat map into the matrix of edge
// keys, S = A + A^T formed on the fly through a tiled transpose so that A and A^T are both read
// coalesced, and extracts every node's (up to) kTopK best partners in exact (key, index) order from the
// two best entries each lane saw in its own columns.  k_greedy runs the merge loop, one WAVE per heat map (no workgroup barriers;
// every step is a short latency chain and many heat maps run side by side on a CU), with per-map
// state in LDS: the fp32 rounding of each node's candidate key and its partner, the node's short
// candidate list and a cursor into it, degree, the other endpoint of each fragment, two neighbour
// slots, and the minimum candidate key of every group of 64 nodes.  An invalidated candidate is
// replaced by the next still-valid entry of the node's list (validity only decays, so skipped
// entries never come back) -- LDS only, 3 of 4 replacements -- and by a rescan of the node's key row
// (one global round trip) once the list is used up.  The best candidate over all nodes is the
// minimum of the group minima; only the groups of the one or two nodes that changed are refreshed.
//
// Arithmetic follows the reference: S is summed in the heat map's dtype (numpy adds float32 arrays
// in float32, greedy_search.py:36-38), keys are -double(S)/sqrt(dx*dx+dy*dy) in double
// (cython_merge.pyx:23,37).  The per-step choice among the nodes' candidates first compares the
// fp32 ROUNDINGS of the keys, which is monotone: the exact minimum is among the entries equal to
// the fp32 minimum, and only those are compared in double (fetched from the key matrix; the common
// tie, two nodes naming each other, is the same edge and needs no fetch).  Where two different edges have
// exactly equal keys the reference's order is numpy's unstable argsort order; here (and in the
// oracle) the lower flat index wins.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "tsp2opt.h"
#include "host_util.hpp"
#include "wave_reduce.hpp"

#pragma clang fp contract(off)

namespace t2o {

// A row's best partners in exact (key, j) order, from the two best entries every lane has seen among
// its own columns (j = lane mod 64): lkey (fp32 rounding, NaN = no entry) and lj (0xffff = no entry).
// Each round takes the best head over the wave (fp32 first, exact on fp32 ties) and the owner pops.
// A lane that has popped both its entries no longer knows its next best, so the extraction stops
// there: the list is then shorter than `rounds` but still an exact prefix of the row's order (the
// merge loop falls back to a row rescan once a list is used up).
constexpr int kTopK = 4;

__device__ __forceinline__ void row_topk(double k0, unsigned j0, double k1, unsigned j1, int lane, int rounds,
                                         float* __restrict__ lkey, uint16_t* __restrict__ lj)
{
    int popped = 0;
    int r = 0;
#pragma unroll 1
    for (; r < rounds; ++r) {
        if (__ballot(popped == 2)) break;               // somebody's next best is unknown
        // an empty head (j = 0xffff) must lose against any real entry, also one whose key is +inf:
        // compare the fp32 key AND the presence bit
        const bool has = j0 != 0xffffu;
        const float hf = has ? (float)k0 : INFINITY;
        const float m = wave_min(hf);
        const unsigned long long bal = __ballot(has && hf == m);
        if (!bal) break;                                // nothing left in this row
        unsigned bj;
        if (__popcll(bal) == 1) {
            bj = (unsigned)__builtin_amdgcn_readlane((int)j0, __ffsll((long long)bal) - 1);
        } else {                                        // fp32 tie: exact (key, j)
            double bk = (has && hf == m) ? k0 : INFINITY;
            bj = (has && hf == m) ? j0 : 0xffffffffu;
            wave_best(bk, bj);
        }
        if (lane == 0) {
            lkey[r] = m;
            lj[r] = (uint16_t)bj;
        }
        if (has && j0 == bj) {                          // the owner pops its head
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

// Kd[b][i][j] = key of edge (i, j) (+inf on the diagonal and for NaN, which numpy sorts last).
// Block (64, 8): a band of 32 rows (4 per thread), column tiles of 64; the transposed operand goes
// through LDS.
// TK = entries tracked per lane and row (2, or 1 for small n where only the row's best is extracted)
template <typename H, int TK>
__global__ void __launch_bounds__(512) k_greedy_prepare(const H* __restrict__ A, const double* __restrict__ points,
                                                        double* __restrict__ Kd, float* __restrict__ lkey,
                                                        uint16_t* __restrict__ lj, int n, int maps_per_instance,
                                                        int64_t map0, int topk_rounds)
{
    __shared__ H tile[64][33];                          // tile[c][r] = A[c0 + c][i0 + r]
    const int64_t b = map0 + blockIdx.y;
    const int x = threadIdx.x, y = threadIdx.y, t = y * 64 + x;
    const int i0 = blockIdx.x * 32;
    const H* Ab = A + (size_t)b * n * n;
    double* Kb = Kd + (size_t)b * n * n;
    const double* P = points + (size_t)(b / maps_per_instance) * n * 2;
    double rx[4], ry[4], b1k[4], b2k[4];                // b1/b2: this lane's two best (key, j) of the row
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
    H a[4], tr[4];
    auto load_tile = [&](int c0) {
        const int j = c0 + x;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = i0 + y + 8 * rr;
            a[rr] = (i < n && j < n) ? Ab[(size_t)i * n + j] : (H)0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (t >> 5) + 16 * k, r = t & 31;
            const int jr = c0 + c, ic = i0 + r;
            tr[k] = (jr < n && ic < n) ? Ab[(size_t)jr * n + ic] : (H)0;
        }
    };
    load_tile(0);
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int j = c0 + x;
        const double cx = j < n ? P[2 * j] : 0.0, cy = j < n ? P[2 * j + 1] : 0.0;
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
                double k = INFINITY;
                if (j != i) {
                    const H s = ac[rr] + tile[x][r];
                    const double dx = rx[rr] - cx, dy = ry[rr] - cy;
                    k = -(double)s / sqrt(dx * dx + dy * dy);
                    k = k != k ? INFINITY : k;
                    if constexpr (TK == 1) {
                        if (better(k, (unsigned)j, b1k[rr], b1j[rr])) { b1k[rr] = k; b1j[rr] = (unsigned)j; }
                    } else if (better(k, (unsigned)j, b2k[rr], b2j[rr])) {
                        if (better(k, (unsigned)j, b1k[rr], b1j[rr])) {
                            b2k[rr] = b1k[rr]; b2j[rr] = b1j[rr];
                            b1k[rr] = k; b1j[rr] = (unsigned)j;
                        } else {
                            b2k[rr] = k; b2j[rr] = (unsigned)j;
                        }
                    }
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

// optional phase timers (tsp2opt_debug_greedy_cycles): [0] scan, [1] merge, [2] rescan cycles,
// [3] merges, [4] rescans, [5] total cycles of the merge loop, summed over heat maps (lane 0)
__device__ unsigned long long g_greedy_prof[8];

struct Cand {               // node i's candidate edge i -> cj, one 8-byte LDS read
    float key;              // fp32 rounding of its key; NaN = the node has no candidate (interior / none left)
    unsigned cj;            // partner
};

struct GreedyLds {          // carve of the dynamic LDS for one heat map (one wave)
    Cand* cand;
    float* lk;              // K list entries per node: fp32 key (NaN = none) ...
    uint16_t* lj;           // ... and partner (0xffff = none), exact (key, j) order
    uint16_t* oend;         // for a fragment endpoint: the fragment's other endpoint (itself if alone)
    uint16_t* nb;           // 2 neighbour slots per node, 0xffff = empty
    uint8_t* deg;
    uint8_t* lp;            // cursor into the node's list (K: list used up, candidates come from rescans)
    int* cmd;               // NW > 1: row to rescan (or -1: done), published by wave 0 to the helper waves
    struct Part* part;      // NW > 1: every wave's best valid partner of that row
};

struct Part {
    double k;
    unsigned j, pad;
};

// The minimum candidate key of every group of 64 nodes lives in registers: lane l holds groups l
// (gm0), l + 64 (gm1) and, for n > 8192 (BIG kernels only), l + 128 (gm2); +inf = no candidate in the
// group.  Refresh the minimum of node x's group after its candidate changed (all 64 lanes).
template <bool BIG>
__device__ __forceinline__ void refresh_group(const GreedyLds& L, int n, int x, int lane, float& gm0, float& gm1,
                                              float& gm2)
{
    if (n <= 128) return;                               // two groups at most: scanned directly
    const int g = x >> 6, k = (g << 6) + lane;
    const float v = k < n ? L.cand[k].key : __builtin_nanf("");
    const float m = wave_min(v != v ? INFINITY : v);
    if (lane == (g & 63)) {
        if (g < 64) gm0 = m;
        else if (!BIG || g < 128) gm1 = m;
        else gm2 = m;
    }
}

// can node i still take partner j?  (j is a fragment endpoint of another fragment)
__device__ __forceinline__ bool partner_ok(const GreedyLds& L, int i, int oe, int j)
{
    return j != 0xffff && L.deg[j] < 2 && !(j == oe && oe != i);
}

// best valid partner of node i from its key row: minimum (key, j) over valid j.  All 64 lanes; the
// row's keys are loaded unconditionally, W per lane in flight (8: n <= 512 in one global round trip
// within 64 VGPRs, i.e. 8 heat maps per SIMD; 16 for larger n, where there are fewer maps than
// SIMDs anyway).  Result stored by lane 0.
// NW waves share a row: wave w takes the columns [64 W (w + NW t), 64 W (w + NW t + 1)), t = 0, 1, ...
template <int W, int NW>
__device__ __forceinline__ void rescan_part(const GreedyLds& L, const double* __restrict__ Kb, int n, int i, int lane,
                                            int wave, double& bk, unsigned& bj)
{
    const double* Krow = Kb + (size_t)i * n;
    const int oe = L.oend[i];
    bk = INFINITY;
    bj = 0xffffu;
    for (int j0 = wave * 64 * W + lane; j0 < n; j0 += 64 * W * NW) {
        double k[W];
#pragma unroll
        for (int u = 0; u < W; ++u) k[u] = j0 + 64 * u < n ? Krow[j0 + 64 * u] : INFINITY;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int j = j0 + 64 * u;
            if (j >= n || j == i || !partner_ok(L, i, oe, j)) continue;
            if (k[u] < bk || (k[u] == bk && (unsigned)j < bj)) { bk = k[u]; bj = (unsigned)j; }
        }
    }
    wave_best(bk, bj);
}

// wave 0's side of a rescan.  NW > 1 (large maps, one workgroup per heat map): the helper waves sleep
// at a barrier between rescans (helper_loop); wave 0 publishes the row, everybody scans a share of it,
// wave 0 reduces the NW partial results.
template <int W, int NW>
__device__ __forceinline__ void rescan(const GreedyLds& L, const double* __restrict__ Kb, int n, int i, int lane)
{
    double bk;
    unsigned bj;
    if constexpr (NW > 1) {
        if (lane == 0) L.cmd[0] = i;
        __syncthreads();
        rescan_part<W, NW>(L, Kb, n, i, lane, 0, bk, bj);
        __syncthreads();
        if (lane >= 1 && lane < NW) { bk = L.part[lane].k; bj = L.part[lane].j; }
        else if (lane != 0) { bk = INFINITY; bj = 0xffffu; }
        wave_best(bk, bj);
    } else {
        rescan_part<W, 1>(L, Kb, n, i, lane, 0, bk, bj);
    }
    if (lane == 0) L.cand[i] = Cand{bj == 0xffffu ? __builtin_nanf("") : (float)bk, bj};
}

template <int W, int NW>
__device__ __forceinline__ void helper_loop(const GreedyLds& L, const double* __restrict__ Kb, int n, int lane, int wave)
{
    for (;;) {
        __syncthreads();                                // wave 0 has published a row, or the end
        const int i = L.cmd[0];
        if (i < 0) return;
        double bk;
        unsigned bj;
        rescan_part<W, NW>(L, Kb, n, i, lane, wave, bk, bj);
        if (lane == 0) L.part[wave] = Part{bk, bj, 0u};
        __syncthreads();                                // partial results are in LDS
    }
}

// node i's candidate is no longer valid: the next valid entry of its list, else a rescan
__device__ __forceinline__ bool advance(const GreedyLds& L, int K, int i, int lane, const float* __restrict__ glk,
                                        const uint16_t* __restrict__ glj)
{
    if (K == 0) {
        // large n: no room for the lists in LDS, they are read where the prepare kernel left them (kTopK
        // entries per node, one global round trip: still less than half of a row rescan)
        const int p0 = L.lp[i], oe = L.oend[i];
        if (p0 >= kTopK - 1) {
            if (lane == 0) L.lp[i] = (uint8_t)kTopK;
            return false;
        }
        float k = 0.f;
        unsigned j = 0xffffu;
        if (lane < kTopK) { k = glk[(size_t)i * kTopK + lane]; j = glj[(size_t)i * kTopK + lane]; }
        const bool ok = lane > p0 && lane < kTopK && partner_ok(L, i, oe, (int)j);
        const unsigned long long bal = __ballot(ok);
        if (!bal) {
            if (lane == 0) L.lp[i] = (uint8_t)kTopK;
            return false;
        }
        const int p = __ffsll((long long)bal) - 1;
        const float pk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k), p));
        const unsigned pj = (unsigned)__builtin_amdgcn_readlane((int)j, p);
        if (lane == 0) {
            L.lp[i] = (uint8_t)p;
            L.cand[i] = Cand{pk, pj};
        }
        return true;
    }
    const int p0 = L.lp[i], oe = L.oend[i];
    bool ok = false;
    if (lane > p0 && lane < K) ok = partner_ok(L, i, oe, L.lj[i * K + lane]);
    const unsigned long long bal = __ballot(ok);
    if (!bal) {
        if (lane == 0) L.lp[i] = (uint8_t)K;
        return false;
    }
    const int p = __ffsll((long long)bal) - 1;
    if (lane == 0) {
        L.lp[i] = (uint8_t)p;
        L.cand[i] = Cand{L.lk[i * K + p], L.lj[i * K + p]};
    }
    return true;
}

// K = list entries kept per node in LDS (kTopK, or fewer when n is large, none above kGreedyListMaxN);
// W: see rescan (the W = 16 kernel is also the one that carries the third group-minimum register)
// NW = waves per heat map: 1 (many maps side by side on a CU) or 16 for maps above kGreedyListMaxN
// nodes, whose state fills a CU's LDS anyway and whose every replacement is a rescan of a long row
template <int W, int NW>
__global__ void __launch_bounds__(64 * NW) k_greedy(const double* __restrict__ Kd, const float* __restrict__ lkey,
                                              const uint16_t* __restrict__ lj0, int64_t* __restrict__ tours,
                                              int64_t num_maps, int n, int K, int prof)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t b = blockIdx.x;
    if (b >= num_maps) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = (n + 63) >> 6;                        // groups of 64 nodes, <= 192
    GreedyLds L;
    size_t off = 0;
    L.cand = reinterpret_cast<Cand*>(smem + off); off += sizeof(Cand) * (size_t)n;
    L.lk = reinterpret_cast<float*>(smem + off); off += sizeof(float) * (size_t)n * K;
    L.lj = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n * K;
    L.oend = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.nb = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * 2 * (size_t)n;
    L.deg = reinterpret_cast<uint8_t*>(smem + off); off += (size_t)n;
    L.lp = reinterpret_cast<uint8_t*>(smem + off);
    off = (off + (size_t)n + 15) & ~(size_t)15;
    L.cmd = reinterpret_cast<int*>(smem + off);         // NW > 1 only (greedy_lds)
    L.part = reinterpret_cast<Part*>(smem + off + 16);
    constexpr bool BIG = W == 16;
    const double* Kb = Kd + (size_t)b * (size_t)n * (size_t)n;
    for (int i = threadIdx.x; i < n; i += 64 * NW) {
        L.oend[i] = (uint16_t)i;
        L.nb[2 * i] = L.nb[2 * i + 1] = 0xffffu;
        L.deg[i] = 0;
        L.lp[i] = 0;
        const size_t src = ((size_t)b * n + i) * kTopK;
        for (int p = 0; p < K; ++p) {
            L.lk[i * K + p] = lkey[src + p];
            L.lj[i * K + p] = lj0[src + p];
        }
        L.cand[i] = Cand{lkey[src], lj0[src]};          // first candidates: the heads of the lists
    }
    if constexpr (NW > 1) {
        __syncthreads();
        if (wave != 0) { helper_loop<W, NW>(L, Kb, n, lane, wave); return; }
    }
    __builtin_amdgcn_wave_barrier();
    float gm0 = INFINITY, gm1 = INFINITY, gm2 = INFINITY;
    for (int g = 0; g < G; ++g) refresh_group<BIG>(L, n, g << 6, lane, gm0, gm1, gm2);
    __builtin_amdgcn_wave_barrier();

    int merges = 0;
    unsigned long long t_scan = 0, t_merge = 0, t_rescan = 0, n_rescan = 0, t_begin = prof ? clock64() : 0;
    // every pass either merges or replaces one node's candidate; a candidate is replaced at most once
    // per change of its validity, so the loop ends (bounded anyway: never spin forever on bad input)
    for (long long guard = 0; merges < n - 1 && guard < (long long)n * n + 1024; ++guard) {
        // best candidate edge over all nodes: minimum of the group minima, then the nodes that attain
        // it (fp32 roundings; NaN = no candidate, never equal); exact (key, flat index) comparison
        // only when more than one node does and they do not simply name each other
        const unsigned long long t0 = prof ? clock64() : 0;
        float m;
        unsigned long long gb0 = 0ull, gb1 = 0ull, gb2 = 0ull;
        int i = -1, second = -1, hits = 0;
        unsigned cji = 0, cjs = 0;                      // partners of i and second
        auto take = [&](unsigned long long bal, int base, unsigned cjl) {
            while (bal && second < 0) {
                const int src = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)cjl, src);
                if (i < 0) { i = base + src; cji = c; } else { second = base + src; cjs = c; }
            }
        };
        if (n <= 128) {                                 // small maps: the candidates themselves
            const Cand c0 = lane < n ? L.cand[lane] : Cand{__builtin_nanf(""), 0u};
            const Cand c1 = lane + 64 < n ? L.cand[lane + 64] : Cand{__builtin_nanf(""), 0u};
            m = wave_min(fminf(c0.key, c1.key));
            const unsigned long long bal = __ballot(c0.key == m), bal1 = __ballot(c1.key == m);
            hits = __popcll(bal) + __popcll(bal1);
            take(bal, 0, c0.cj);
            take(bal1, 64, c1.cj);
        } else {
            m = wave_min(BIG ? fminf(fminf(gm0, gm1), gm2) : fminf(gm0, gm1));
            gb0 = __ballot(gm0 == m);
            gb1 = __ballot(gm1 == m);
            if (BIG) gb2 = __ballot(gm2 == m);
        }
        while (gb0 | gb1 | gb2) {                       // groups whose minimum is m (usually one)
            int g;
            if (gb0) { g = __ffsll((long long)gb0) - 1; gb0 &= gb0 - 1; }
            else if (gb1) { g = 64 + __ffsll((long long)gb1) - 1; gb1 &= gb1 - 1; }
            else { g = 128 + __ffsll((long long)gb2) - 1; gb2 &= gb2 - 1; }
            const int k = (g << 6) + lane;
            const Cand c = k < n ? L.cand[k] : Cand{__builtin_nanf(""), 0u};
            const unsigned long long bal = __ballot(c.key == m);
            hits += __popcll(bal);
            take(bal, g << 6, c.cj);
        }
        if (hits == 0) break;                           // no valid edge left (cannot happen for n >= 2)
        if (hits > 1 && !(hits == 2 && cji == (unsigned)second && cjs == (unsigned)i)) {
            // different edges with the same fp32 key: exact keys from the key matrix.  (Two nodes that
            // name each other are one edge with one key; the lower flat index is the lower node = i.)
            double bk = INFINITY;
            unsigned bf = 0xffffffffu;
            for (int k = lane; k < n; k += 64) {
                const Cand c = L.cand[k];
                if (!(c.key == m)) continue;
                const unsigned f = (unsigned)k * (unsigned)n + c.cj;
                const double key = Kb[f];
                if (better(key, f, bk, bf)) { bk = key; bf = f; }
            }
            wave_best(bk, bf);
            i = (int)(bf / (unsigned)n);
            cji = bf - (unsigned)i * (unsigned)n;
        }
        const int j = (int)cji;
        const int oi = L.oend[i], oj = L.oend[j];
        const bool valid = L.deg[j] < 2 && !(oi == j && oi != i);
        const unsigned long long t1 = prof ? clock64() : 0;
        if (valid) {
            if (lane == 0) {
                L.nb[2 * i + (L.nb[2 * i] != 0xffffu)] = (uint16_t)j;
                L.nb[2 * j + (L.nb[2 * j] != 0xffffu)] = (uint16_t)i;
                const int di = L.deg[i] + 1, dj = L.deg[j] + 1;
                L.deg[i] = (uint8_t)di;
                L.deg[j] = (uint8_t)dj;
                if (di >= 2) L.cand[i].key = __builtin_nanf("");
                if (dj >= 2) L.cand[j].key = __builtin_nanf("");
                L.oend[oi] = (uint16_t)oj;      // the merged fragment runs from oi to oj
                L.oend[oj] = (uint16_t)oi;
            }
            ++merges;
            __builtin_amdgcn_wave_barrier();
            refresh_group<BIG>(L, n, i, lane, gm0, gm1, gm2);
            if ((j >> 6) != (i >> 6)) refresh_group<BIG>(L, n, j, lane, gm0, gm1, gm2);
        } else {
            if (!advance(L, K, i, lane, lkey + (size_t)b * n * kTopK, lj0 + (size_t)b * n * kTopK))
                rescan<W, NW>(L, Kb, n, i, lane);
            __builtin_amdgcn_wave_barrier();
            refresh_group<BIG>(L, n, i, lane, gm0, gm1, gm2);
        }
        __builtin_amdgcn_wave_barrier();
        if (prof) {
            const unsigned long long t2 = clock64();
            t_scan += t1 - t0;
            if (valid) t_merge += t2 - t1; else { t_rescan += t2 - t1; ++n_rescan; }
        }
    }
    if constexpr (NW > 1) {                             // release the helper waves
        if (lane == 0) L.cmd[0] = -1;
        __syncthreads();
    }
    if (prof && lane == 0) {
        atomicAdd(&g_greedy_prof[0], t_scan);
        atomicAdd(&g_greedy_prof[1], t_merge);
        atomicAdd(&g_greedy_prof[2], t_rescan);
        atomicAdd(&g_greedy_prof[3], (unsigned long long)merges);
        atomicAdd(&g_greedy_prof[4], n_rescan);
        atomicAdd(&g_greedy_prof[5], clock64() - t_begin);
    }
    // closing edge between the two remaining endpoints (cython_merge.pyx:99-102), then the walk of
    // greedy_search.py:65-73: from node 0, never back to the previous node, larger index first
    if (lane == 0) {
        int a = -1, c = -1;
        for (int i = 0; i < n; ++i)
            if (L.deg[i] < 2) { if (a < 0) a = i; else c = i; }
        if (a >= 0 && c >= 0) {
            L.nb[2 * a + (L.nb[2 * a] != 0xffffu)] = (uint16_t)c;
            L.nb[2 * c + (L.nb[2 * c] != 0xffffu)] = (uint16_t)a;
        }
        int64_t* t = tours + (size_t)b * (size_t)(n + 1);
        int prev = -1, cur = 0;
        t[0] = 0;
        for (int k = 1; k <= n; ++k) {
            const int n0 = L.nb[2 * cur], n1 = L.nb[2 * cur + 1];
            int best = -1;
            if (n0 != 0xffff && n0 != prev) best = n0;
            if (n1 != 0xffff && n1 != prev && n1 > best) best = n1;
            t[k] = best;
            prev = cur;
            cur = best < 0 ? 0 : best;
        }
    }
}

constexpr int kGreedyListMaxN = 4500;   // lists of 2 fit the LDS of one wave's heat map up to here
constexpr int kGreedyMaxN = 10200;      // 16 B of state per node in 160 KB; 160 groups of 64 nodes (<= 192)
constexpr int kGreedyBigWaves = 16;     // waves per heat map above kGreedyListMaxN nodes

// list entries per node (extracted by the prepare kernel and kept in LDS by the merge loop): kTopK
// while a heat map's state stays near 64 KB; 2 for larger n; none above kGreedyListMaxN (the state of
// a TSP-10000 map alone is 160 KB: the lists stay in global memory); 1 (the row's best only) for small
// n, where tracking and extracting more costs the prepare kernel more than the merge loop gains
inline int greedy_k(int n) { return n <= 128 ? 1 : n <= 1700 ? kTopK : n <= kGreedyListMaxN ? 2 : 0; }
// entries the prepare kernel extracts per node
inline int greedy_rounds(int n) { return greedy_k(n) ? greedy_k(n) : kTopK; }

inline size_t greedy_lds(int n)
{
    const int K = greedy_k(n);
    return (size_t)n * (8 + 6 * (size_t)K + 2 + 4 + 1 + 1) + 64 +
           (K ? 0 : 32 + sizeof(Part) * kGreedyBigWaves);
}

struct GreedyWs {
    size_t off_lkey, off_lj, total;     // key matrices at 0
};

inline GreedyWs greedy_ws(int64_t num_maps, int n)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    GreedyWs w{};
    size_t o = up((size_t)num_maps * (size_t)n * (size_t)n * 8);
    w.off_lkey = o; o += up((size_t)num_maps * (size_t)n * kTopK * 4);
    w.off_lj = o; o += up((size_t)num_maps * (size_t)n * kTopK * 2);
    w.total = o + 256;
    return w;
}

int g_greedy_prof_on = 0;

template <typename H>
int greedy_t(const void* heat, const double* points, int64_t* tours, int64_t num_maps, int n, int mpi,
             void* ws, hipStream_t stream)
{
    const H* A = static_cast<const H*>(heat);
    const GreedyWs wl = greedy_ws(num_maps, n);
    unsigned char* w8 = static_cast<unsigned char*>(ws);
    double* Kd = reinterpret_cast<double*>(w8);
    float* lkey = reinterpret_cast<float*>(w8 + wl.off_lkey);
    uint16_t* lj = reinterpret_cast<uint16_t*>(w8 + wl.off_lj);
    // grid.y is limited to 65535: prepare in slabs of heat maps
    for (int64_t b0 = 0; b0 < num_maps; b0 += 32768) {
        const dim3 grid((unsigned)((n + 31) / 32), (unsigned)std::min<int64_t>(32768, num_maps - b0), 1);
        if (greedy_rounds(n) == 1)
            hipLaunchKernelGGL((k_greedy_prepare<H, 1>), grid, dim3(64, 8), 0, stream, A, points, Kd, lkey, lj, n, mpi, b0, 1);
        else
            hipLaunchKernelGGL((k_greedy_prepare<H, 2>), grid, dim3(64, 8), 0, stream, A, points, Kd, lkey, lj, n, mpi, b0, greedy_rounds(n));
        HIP_TRY(hipGetLastError());
    }
    const size_t lds = greedy_lds(n);
    const bool big = greedy_k(n) == 0;
    auto kern = n <= 512 ? k_greedy<8, 1> : big ? k_greedy<16, kGreedyBigWaves> : k_greedy<16, 1>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)num_maps), dim3(big ? 64 * kGreedyBigWaves : 64), lds, stream, Kd,
                       lkey, lj, tours, num_maps, n, greedy_k(n), g_greedy_prof_on);
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
    if (enable) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_greedy_prof), z, sizeof(z)));
        g_greedy_prof_on = 1;
        return 0;
    }
    g_greedy_prof_on = 0;
    if (out8) HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_greedy_prof), 8 * sizeof(unsigned long long)));
    return 0;
}
