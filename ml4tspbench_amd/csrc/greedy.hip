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
// One WAVE per heat map (no workgroup barriers; every step is a short latency chain and many heat
// maps run side by side on a CU).  Per-map state lives in LDS: coordinates (double), candidate key
// and partner, degree, the other endpoint of each fragment, two neighbour slots.  The symmetrised
// heat map S = A + A^T is built first by a tiled transpose-add so that row rescans are coalesced.
//
// Arithmetic follows the reference: S is summed in the heat map's dtype (numpy adds float32 arrays
// in float32, greedy_search.py:36-38), keys are -double(S)/sqrt(dx*dx+dy*dy) in double
// (cython_merge.pyx:23,37).  Where two different edges have exactly equal keys the reference's
// order is numpy's unstable argsort order; here (and in the oracle) the lower flat index wins.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "tsp2opt.h"
#include "host_util.hpp"
#include "wave_reduce.hpp"

#pragma clang fp contract(off)

namespace t2o {

// S[b][i][j] = A[b][i][j] + A[b][j][i], 32x32 tiles through LDS (both reads coalesced)
template <typename H>
__global__ void k_symmetrize(const H* __restrict__ A, H* __restrict__ S, int n)
{
    __shared__ H tile[32][33];
    const size_t base = (size_t)blockIdx.z * (size_t)n * (size_t)n;
    const int bi = blockIdx.y * 32, bj = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {         // tile of A^T source: rows bj.., cols bi..
        const int i = bj + r, j = bi + threadIdx.x;
        tile[r][threadIdx.x] = (i < n && j < n) ? A[base + (size_t)i * n + j] : (H)0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = bi + r, j = bj + threadIdx.x;
        if (i < n && j < n) S[base + (size_t)i * n + j] = A[base + (size_t)i * n + j] + tile[threadIdx.x][r];
    }
}

struct GreedyLds {          // carve of the dynamic LDS for one heat map (one wave)
    double* px;
    double* py;
    double* ckey;           // candidate key of node i (key of edge i -> cj[i])
    uint16_t* cj;           // candidate partner
    uint16_t* oend;         // for a fragment endpoint: the fragment's other endpoint (itself if alone)
    uint16_t* nb;           // 2 neighbour slots per node, 0xffff = empty
    uint8_t* deg;
};

__device__ __forceinline__ double edge_key(const GreedyLds& L, double s, int i, int j)
{
    const double dx = L.px[i] - L.px[j], dy = L.py[i] - L.py[j];
    const double k = -s / sqrt(dx * dx + dy * dy);
    return k != k ? INFINITY : k;               // NaN keys sort last in numpy
}

// best valid partner of node i: minimum (key, j) over j that are fragment endpoints of another
// fragment.  All 64 lanes; result stored by lane 0.
template <typename H>
__device__ __forceinline__ void rescan(const GreedyLds& L, const H* __restrict__ Srow, int n, int i, int lane)
{
    double bk = INFINITY;
    unsigned bj = 0xffffu;
    const int oe = L.oend[i];
    for (int j = lane; j < n; j += 64) {
        if (j == i || L.deg[j] >= 2 || (j == oe && oe != i)) continue;
        const double k = edge_key(L, (double)Srow[j], i, j);
        if (k < bk || (k == bk && (unsigned)j < bj)) { bk = k; bj = (unsigned)j; }
    }
    wave_best(bk, bj);
    if (lane == 0) { L.ckey[i] = bk; L.cj[i] = (uint16_t)bj; }
}

template <typename H>
__global__ void __launch_bounds__(64) k_greedy(const H* __restrict__ S, const double* __restrict__ points,
                                              int64_t* __restrict__ tours, int64_t num_maps, int n,
                                              int maps_per_instance)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t b = blockIdx.x;
    if (b >= num_maps) return;
    const int lane = threadIdx.x;
    GreedyLds L;
    size_t off = 0;
    L.px = reinterpret_cast<double*>(smem + off); off += sizeof(double) * (size_t)n;
    L.py = reinterpret_cast<double*>(smem + off); off += sizeof(double) * (size_t)n;
    L.ckey = reinterpret_cast<double*>(smem + off); off += sizeof(double) * (size_t)n;
    L.cj = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.oend = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * (size_t)n;
    L.nb = reinterpret_cast<uint16_t*>(smem + off); off += sizeof(uint16_t) * 2 * (size_t)n;
    L.deg = reinterpret_cast<uint8_t*>(smem + off);
    const H* Sb = S + (size_t)b * (size_t)n * (size_t)n;
    const double* P = points + (size_t)(b / maps_per_instance) * (size_t)n * 2;
    for (int i = lane; i < n; i += 64) {
        L.px[i] = P[2 * i];
        L.py[i] = P[2 * i + 1];
        L.oend[i] = (uint16_t)i;
        L.nb[2 * i] = L.nb[2 * i + 1] = 0xffffu;
        L.deg[i] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < n; ++i) rescan(L, Sb + (size_t)i * n, n, i, lane);
    __builtin_amdgcn_wave_barrier();

    int merges = 0;
    // every pass either merges or rescans one node; a node is rescanned at most once per change of
    // validity of its candidate, so the loop ends (bounded anyway: never spin forever on bad input)
    for (long long guard = 0; merges < n - 1 && guard < (long long)n * n + 1024; ++guard) {
        double bk = INFINITY;
        unsigned bf = 0xffffffffu;              // flat index i*n + j of the best candidate edge
        for (int i = lane; i < n; i += 64) {
            if (L.deg[i] >= 2) continue;
            const unsigned j = L.cj[i];
            if (j == 0xffffu) continue;
            const double k = L.ckey[i];
            const unsigned f = (unsigned)i * (unsigned)n + j;
            if (k < bk || (k == bk && f < bf)) { bk = k; bf = f; }
        }
        wave_best(bk, bf);
        if (bf == 0xffffffffu) break;           // no valid edge left (cannot happen for n >= 2)
        const int i = (int)(bf / (unsigned)n), j = (int)(bf % (unsigned)n);
        const int oi = L.oend[i], oj = L.oend[j];
        const bool valid = L.deg[j] < 2 && !(oi == j && oi != i);
        if (valid) {
            if (lane == 0) {
                L.nb[2 * i + (L.nb[2 * i] != 0xffffu)] = (uint16_t)j;
                L.nb[2 * j + (L.nb[2 * j] != 0xffffu)] = (uint16_t)i;
                L.deg[i] = L.deg[i] + 1;
                L.deg[j] = L.deg[j] + 1;
                L.oend[oi] = (uint16_t)oj;      // the merged fragment runs from oi to oj
                L.oend[oj] = (uint16_t)oi;
            }
            ++merges;
            __builtin_amdgcn_wave_barrier();
        } else {
            rescan(L, Sb + (size_t)i * n, n, i, lane);
        }
        __builtin_amdgcn_wave_barrier();
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

constexpr int kGreedyMaxN = 4500;       // 33 B of LDS per node

inline size_t greedy_lds(int n) { return (size_t)n * (8 + 8 + 8 + 2 + 2 + 4 + 1) + 64; }

template <typename H>
int greedy_t(const void* heat, const double* points, int64_t* tours, int64_t num_maps, int n, int mpi,
             void* ws, hipStream_t stream)
{
    const H* A = static_cast<const H*>(heat);
    H* S = static_cast<H*>(ws);
    const dim3 grid((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32), 1);
    // grid.z is limited to 65535: symmetrise in slabs
    for (int64_t b0 = 0; b0 < num_maps; b0 += 32768) {
        dim3 g = grid;
        g.z = (unsigned)std::min<int64_t>(32768, num_maps - b0);
        hipLaunchKernelGGL(k_symmetrize<H>, g, dim3(32, 8), 0, stream, A + (size_t)b0 * n * n,
                           S + (size_t)b0 * n * n, n);
        HIP_TRY(hipGetLastError());
    }
    const size_t lds = greedy_lds(n);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_greedy<H>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_greedy<H>, dim3((unsigned)num_maps), dim3(64), lds, stream, S, points, tours,
                       num_maps, n, mpi);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace t2o

using namespace t2o;

extern "C" int64_t tsp2opt_greedy_workspace_bytes(int64_t num_maps, int32_t n, int32_t heat_dtype)
{
    if (num_maps < 0 || n < 2) return fail(TSP2OPT_E_ARG, "bad size");
    if (heat_dtype != TSP2OPT_DTYPE_F32 && heat_dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    return (int64_t)((size_t)num_maps * (size_t)n * (size_t)n * (heat_dtype == TSP2OPT_DTYPE_F64 ? 8 : 4)) + 256;
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
