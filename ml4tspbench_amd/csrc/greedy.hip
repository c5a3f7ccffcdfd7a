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
// coalesced, and finds each node's first candidate.  k_greedy runs the merge loop, one WAVE per
// heat map (no workgroup barriers; every step is a short latency chain and many heat maps run side
// by side on a CU), with per-map state in LDS: candidate key (double, plus its fp32 rounding) and
// partner, degree, the other endpoint of each fragment, two neighbour slots.  A rescan is one
// coalesced read of the node's key row: a single global round trip.
//
// Arithmetic follows the reference: S is summed in the heat map's dtype (numpy adds float32 arrays
// in float32, greedy_search.py:36-38), keys are -double(S)/sqrt(dx*dx+dy*dy) in double
// (cython_merge.pyx:23,37).  The per-step choice among the nodes' candidates first compares the
// fp32 ROUNDINGS of the keys, which is monotone: the exact minimum is among the entries equal to
// the fp32 minimum, and only those are compared in double.  Where two different edges have
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

// Kd[b][i][j] = key of edge (i, j) (+inf on the diagonal and for NaN, which numpy sorts last);
// ckey/cj[b][i] = the row's minimum (key, j).  Block (64, 4): a band of 32 rows (8 per thread),
// column tiles of 64; the transposed operand goes through LDS.
template <typename H>
__global__ void __launch_bounds__(256) k_greedy_prepare(const H* __restrict__ A, const double* __restrict__ points,
                                                        double* __restrict__ Kd, double* __restrict__ ckey,
                                                        unsigned* __restrict__ cj, int n, int maps_per_instance,
                                                        int64_t map0)
{
    __shared__ H tile[64][33];                          // tile[c][r] = A[c0 + c][i0 + r]
    const int64_t b = map0 + blockIdx.y;
    const int x = threadIdx.x, y = threadIdx.y, t = y * 64 + x;
    const int i0 = blockIdx.x * 32;
    const H* Ab = A + (size_t)b * n * n;
    double* Kb = Kd + (size_t)b * n * n;
    const double* P = points + (size_t)(b / maps_per_instance) * n * 2;
    double rx[8], ry[8], bk[8];
    unsigned bj[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int i = i0 + y + 4 * rr;
        rx[rr] = i < n ? P[2 * i] : 0.0;
        ry[rr] = i < n ? P[2 * i + 1] : 0.0;
        bk[rr] = INFINITY;
        bj[rr] = 0xffffu;
    }
    for (int c0 = 0; c0 < n; c0 += 64) {
        // both operands' global loads are issued before the barrier: one round trip per tile
        const int j = c0 + x;
        H a[8], tr[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int i = i0 + y + 4 * rr;
            a[rr] = (i < n && j < n) ? Ab[(size_t)i * n + j] : (H)0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = (t >> 5) + 8 * k, r = t & 31;
            const int jr = c0 + c, ic = i0 + r;
            tr[k] = (jr < n && ic < n) ? Ab[(size_t)jr * n + ic] : (H)0;
        }
        const double cx = j < n ? P[2 * j] : 0.0, cy = j < n ? P[2 * j + 1] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[(t >> 5) + 8 * k][t & 31] = tr[k];
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = y + 4 * rr, i = i0 + r;
            if (i < n && j < n) {
                double k = INFINITY;
                if (j != i) {
                    const H s = a[rr] + tile[x][r];
                    const double dx = rx[rr] - cx, dy = ry[rr] - cy;
                    k = -(double)s / sqrt(dx * dx + dy * dy);
                    k = k != k ? INFINITY : k;
                    if (k < bk[rr] || (k == bk[rr] && (unsigned)j < bj[rr])) { bk[rr] = k; bj[rr] = (unsigned)j; }
                }
                Kb[(size_t)i * n + j] = k;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {                    // the 64 lanes of wave y hold row y + 4 rr
        const int i = i0 + y + 4 * rr;
        double k = bk[rr];
        unsigned j = bj[rr];
        wave_best(k, j);
        if (x == 0 && i < n) { ckey[(size_t)b * n + i] = k; cj[(size_t)b * n + i] = j; }
    }
}

// optional phase timers (tsp2opt_debug_greedy_cycles): [0] scan, [1] merge, [2] rescan cycles,
// [3] merges, [4] rescans, [5] total cycles of the merge loop, summed over heat maps (lane 0)
__device__ unsigned long long g_greedy_prof[8];

struct GreedyLds {          // carve of the dynamic LDS for one heat map (one wave)
    double* ckey;           // exact key of node i's candidate edge i -> cj[i]
    float* capx;            // its fp32 rounding; NaN = node has no candidate (interior / none left)
    uint16_t* cj;           // candidate partner
    uint16_t* oend;         // for a fragment endpoint: the fragment's other endpoint (itself if alone)
    uint16_t* nb;           // 2 neighbour slots per node, 0xffff = empty
    uint8_t* deg;
};

// best valid partner of node i: minimum (key, j) over j that are fragment endpoints of another
// fragment.  All 64 lanes; result stored by lane 0.  The row's keys are loaded unconditionally,
// eight per lane in flight, so a rescan costs one global round trip for n <= 512.
__device__ __forceinline__ void rescan(const GreedyLds& L, const double* __restrict__ Kb, int n, int i, int lane)
{
    const double* Krow = Kb + (size_t)i * n;
    const int oe = L.oend[i];
    double bk = INFINITY;
    unsigned bj = 0xffffu;
    for (int j0 = lane; j0 < n; j0 += 512) {
        double k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = j0 + 64 * u < n ? Krow[j0 + 64 * u] : INFINITY;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + 64 * u;
            if (j >= n || j == i || L.deg[j] >= 2 || (j == oe && oe != i)) continue;
            if (k[u] < bk || (k[u] == bk && (unsigned)j < bj)) { bk = k[u]; bj = (unsigned)j; }
        }
    }
    wave_best(bk, bj);
    if (lane == 0) {
        L.ckey[i] = bk;
        L.cj[i] = (uint16_t)bj;
        L.capx[i] = bj == 0xffffu ? __builtin_nanf("") : (float)bk;
    }
}

// n <= 64 U: minimum of the candidates' fp32 keys over the wave, how many nodes attain it, and one of them
template <int U>
__device__ __forceinline__ void scan_small(const GreedyLds& L, int n, int lane, float& m, int& i, int& hits)
{
    float a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = lane + 64 * u < n ? L.capx[lane + 64 * u] : __builtin_nanf("");
    float la = INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) la = fminf(la, a[u]);
    m = wave_min(la);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned long long bal = __ballot(a[u] == m);
        if (bal) { hits += __popcll(bal); i = 64 * u + __ffsll((long long)bal) - 1; }
    }
}

__global__ void __launch_bounds__(64) k_greedy(const double* __restrict__ Kd, const double* __restrict__ ckey0,
                                              const unsigned* __restrict__ cj0, int64_t* __restrict__ tours,
                                              int64_t num_maps, int n, int prof)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t b = blockIdx.x;
    if (b >= num_maps) return;
    const int lane = threadIdx.x;
    GreedyLds L;
    size_t off = 0;
    L.ckey = reinterpret_cast<double*>(smem + off); off += sizeof(double) * (size_t)n;
    L.capx = reinterpret_cast<float*>(smem + off); off += sizeof(float) * (size_t)n;
    L.cj = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.oend = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.nb = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * 2 * (size_t)n;
    L.deg = reinterpret_cast<uint8_t*>(smem + off);
    const double* Kb = Kd + (size_t)b * (size_t)n * (size_t)n;
    for (int i = lane; i < n; i += 64) {
        L.oend[i] = (uint16_t)i;
        L.nb[2 * i] = L.nb[2 * i + 1] = 0xffffu;
        L.deg[i] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) {               // first candidates: from the prepare kernel
        const double k = ckey0[(size_t)b * n + i];
        L.ckey[i] = k;
        L.cj[i] = (uint16_t)cj0[(size_t)b * n + i];
        L.capx[i] = (float)k;
    }
    __builtin_amdgcn_wave_barrier();

    int merges = 0;
    unsigned long long t_scan = 0, t_merge = 0, t_rescan = 0, n_rescan = 0, t_begin = prof ? clock64() : 0;
    // every pass either merges or rescans one node; a node is rescanned at most once per change of
    // validity of its candidate, so the loop ends (bounded anyway: never spin forever on bad input)
    for (long long guard = 0; merges < n - 1 && guard < (long long)n * n + 1024; ++guard) {
        // best candidate edge over all nodes: minimum of the fp32 roundings (NaN = no candidate,
        // ignored by min and never equal), then the nodes that attain it; exact (key, flat index)
        // comparison only when more than one does
        int i = -1, hits = 0;
        float m;
        const unsigned long long t0 = prof ? clock64() : 0;
        if (n <= 128) {
            scan_small<2>(L, n, lane, m, i, hits);
        } else if (n <= 256) {
            scan_small<4>(L, n, lane, m, i, hits);
        } else if (n <= 512) {
            scan_small<8>(L, n, lane, m, i, hits);
        } else {
            float la = INFINITY;
            for (int k = lane; k < n; k += 64) la = fminf(la, L.capx[k]);
            m = wave_min(la);
            for (int k0 = 0; k0 < n; k0 += 64) {
                const unsigned long long bal = __ballot(k0 + lane < n && L.capx[k0 + lane] == m);
                if (bal) { hits += __popcll(bal); i = k0 + __ffsll((long long)bal) - 1; }
            }
        }
        if (hits == 0) break;                           // no valid edge left (cannot happen for n >= 2)
        if (hits > 1) {
            double bk = INFINITY;
            unsigned bf = 0xffffffffu;
            for (int k = lane; k < n; k += 64) {
                if (!(L.capx[k] == m)) continue;
                const unsigned f = (unsigned)k * (unsigned)n + L.cj[k];
                if (better(L.ckey[k], f, bk, bf)) { bk = L.ckey[k]; bf = f; }
            }
            wave_best(bk, bf);
            i = (int)(bf / (unsigned)n);
        }
        const int j = L.cj[i];
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
                if (di >= 2) L.capx[i] = __builtin_nanf("");
                if (dj >= 2) L.capx[j] = __builtin_nanf("");
                L.oend[oi] = (uint16_t)oj;      // the merged fragment runs from oi to oj
                L.oend[oj] = (uint16_t)oi;
            }
            ++merges;
            __builtin_amdgcn_wave_barrier();
        } else {
            rescan(L, Kb, n, i, lane);
        }
        __builtin_amdgcn_wave_barrier();
        if (prof) {
            const unsigned long long t2 = clock64();
            t_scan += t1 - t0;
            if (valid) t_merge += t2 - t1; else { t_rescan += t2 - t1; ++n_rescan; }
        }
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

constexpr int kGreedyMaxN = 4500;       // 21 B of LDS per node

inline size_t greedy_lds(int n) { return (size_t)n * (8 + 4 + 2 + 2 + 4 + 1) + 64; }

struct GreedyWs {
    size_t off_key, off_cj, total;      // key matrices at 0
};

inline GreedyWs greedy_ws(int64_t num_maps, int n)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    GreedyWs w{};
    size_t o = up((size_t)num_maps * (size_t)n * (size_t)n * 8);
    w.off_key = o; o += up((size_t)num_maps * (size_t)n * 8);
    w.off_cj = o; o += up((size_t)num_maps * (size_t)n * 4);
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
    double* ckey = reinterpret_cast<double*>(w8 + wl.off_key);
    unsigned* cj = reinterpret_cast<unsigned*>(w8 + wl.off_cj);
    // grid.y is limited to 65535: prepare in slabs of heat maps
    for (int64_t b0 = 0; b0 < num_maps; b0 += 32768) {
        const dim3 grid((unsigned)((n + 31) / 32), (unsigned)std::min<int64_t>(32768, num_maps - b0), 1);
        hipLaunchKernelGGL(k_greedy_prepare<H>, grid, dim3(64, 4), 0, stream, A, points, Kd, ckey, cj, n, mpi, b0);
        HIP_TRY(hipGetLastError());
    }
    const size_t lds = greedy_lds(n);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_greedy),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_greedy, dim3((unsigned)num_maps), dim3(64), lds, stream, Kd, ckey, cj, tours,
                       num_maps, n, g_greedy_prof_on);
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
    if (n > kGreedyMaxN) return fail(TSP2OPT_E_UNSUPPORTED, "greedy decode supports n <= 4500");
    if (heat_dtype != TSP2OPT_DTYPE_F32 && heat_dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    if (num_maps == 0) return 0;
    if (!heat || !points || !tours_out || !workspace) return fail(TSP2OPT_E_ARG, "null pointer");
    if (workspace_bytes < tsp2opt_greedy_workspace_bytes(num_maps, n, heat_dtype))
        return fail(TSP2OPT_E_WORKSPACE, "workspace too small");
    if (num_maps > 0x7fffffffll) return fail(TSP2OPT_E_UNSUPPORTED, "too many heat maps");
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
