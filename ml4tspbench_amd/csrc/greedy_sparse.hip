// ml4tspbench_amd/csrc/greedy_sparse.hip -- greedy edge-merge decoder for SPARSE heat maps
//
// Replaces (reference, read-only) the sparse branch of tsp_greedy (search/greedy_search.py:39-47) +
// merge_cython (search/cython_merge/sources/cython_merge.pyx:19-120): the reference densifies the
// (values, edge_index) heat map into two N x N arrays, adds them, and argsorts all N^2 keys
// -S_ij / |v_i - v_j| -- 800 MB of doubles per heat map at N = 10 000 for a k = 50 input.
//
// Here the listed edges stay a list.  Structure phase (once per instance, the edge_index is shared by the
// heat maps of an instance; a call takes one list per instance, all of one length, and works on all of
// them at once): count / scan / fill build, for every node, the entries (partner, edge id,
// direction) of every listed edge that touches it, and a rank sort per row orders them by partner so
// that the two directions (and duplicates) of one pair sit next to each other.  Per heat map: the row's
// entries are summed per partner in the heat map's dtype exactly like the reference's two coo matrices
// (S = sum of (i,j) entries + sum of (j,i) entries), keyed in double, and the partners with S > 0 are
// rank-sorted by exact (key, j): every node's complete candidate list, best first.  The merge loop
// (one workgroup per heat map, 15 B of LDS state per node) then only ever advances cursors: "the best
// still-valid edge" over the nodes' current candidates is the edge the sorted scan would insert next, an
// invalidated candidate is replaced by the next valid entry of the node's list, and edges whose turn nothing
// can disturb are inserted many per round (see k_greedy_sparse).
// When no listed edge is valid any more, every remaining pair has S = 0, i.e. the same key -0.0 (NaN, which
// sorts last, where the two points coincide), and the reference's order among them is numpy's unstable
// argsort (undefined); like the dense path and the oracle, the lower flat index i*n + j wins: the smallest
// node that is still a fragment end takes its smallest valid partner, until one path is left.  Negative sums (keys that would sort AFTER the
// zeros) are reported (TSP2OPT_E_UNSUPPORTED) so that the caller can take the dense path.
//
// Workspace: 12 B per listed edge and instance for the sorted rows + 12 B per listed edge and heat map for
// the lists (two entries per edge: partner, fp32 key) instead of 8 N^2.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "tsp2opt.h"
#include "host_util.hpp"
#include "wave_reduce.hpp"
#include "greedy_walk.hpp"

#pragma clang fp contract(off)

namespace t2o {

constexpr int kSpMaxN = 10900;          // 15 B of LDS state per node
constexpr int kSpMaxRow = 7000;         // entries of one row the row kernels accept (22 B of LDS each)

// ---- structure phase ---------------------------------------------------------------------------------
// blockIdx.y = instance in the structure kernels: its edge list ei + inst*2E, its rows at inst*(n+1), its
// entries at inst*2E.
// Edge lists are usually source-major (k consecutive edges share their source), so a wave's source-side
// atomics would all hit one or two counters: runs of equal consecutive sources are added by their first lane.
struct SrcRun {
    int head;               // first lane of this lane's run
    int len;                // run length (meaningful in the head lane)
};

__device__ __forceinline__ SrcRun src_run(int key, int lane)
{
    const int prev = __shfl_up(key, 1);
    const unsigned long long heads = __ballot(lane == 0 || prev != key);
    const unsigned long long upto = heads & ((2ull << lane) - 1ull);        // lane 63: shift wraps to 0, mask = all
    const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
    SrcRun r;
    r.head = 63 - __clzll((long long)upto);
    r.len = (above ? lane + __ffsll((long long)above) : 64) - lane;         // head lane: distance to the next head
    return r;
}

// Counters and cursors are privatised: a workgroup takes a chunk of one instance's edges, counts (or hands out
// positions) in an LDS table of n words with LDS atomics and touches global memory once per row it met -- one
// global atomic per (chunk, row) instead of two per edge (16 x 500 000 edges: 0.34 ms -> see profiles/).
constexpr int kSpChunkThreads = 1024;

// edge e of the chunk -> (a, b, valid); out-of-range indices are reported in *err (bit 0) when err != nullptr
__device__ __forceinline__ bool sp_edge(const int64_t* __restrict__ ei, int64_t E, int n, int64_t e, int64_t e1,
                                        int64_t& a, int64_t& b, int* __restrict__ err)
{
    a = -1; b = -1;
    if (e < e1) {
        a = ei[e];
        b = ei[E + e];
        if (a < 0 || a >= n || b < 0 || b >= n) { if (err) atomicOr(err, 1); a = -1; }
    }
    return a >= 0 && a != b;                            // the diagonal is never a valid edge
}

__global__ void __launch_bounds__(kSpChunkThreads) k_sp_count(const int64_t* __restrict__ ei, int64_t E, int n,
                                                              int* __restrict__ cnt, int* __restrict__ err, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* hist = reinterpret_cast<int*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, T = blockDim.x;
    ei += (size_t)blockIdx.y * 2 * (size_t)E;
    cnt += (size_t)blockIdx.y * ((size_t)n + 1);
    const int64_t e0 = (int64_t)blockIdx.x * chunk, e1 = e0 + chunk < E ? e0 + chunk : E;
    for (int i = tid; i < n; i += T) hist[i] = 0;
    __syncthreads();
    for (int64_t eb = e0; eb < e1; eb += T) {           // (uniform trip count: src_run shuffles across the wave)
        int64_t a, b;
        const bool valid = sp_edge(ei, E, n, eb + tid, e1, a, b, err);
        const SrcRun r = src_run(valid ? (int)a : -1 - lane, lane);
        if (valid) {
            if (r.head == lane) atomicAdd(&hist[a], r.len);
            atomicAdd(&hist[b], 1);
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += T) {
        const int c = hist[i];
        if (c) atomicAdd(&cnt[i], c);
    }
}

// exclusive scan of cnt[0..n) into off[0..n], one block; also zeroes the fill cursors and checks the row cap
__global__ void __launch_bounds__(1024) k_sp_scan(const int* __restrict__ cnt, int* __restrict__ off, int* __restrict__ cur,
                                                  int n, int cap, int* __restrict__ err)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    cnt += (size_t)blockIdx.x * ((size_t)n + 1);                        // one block per instance
    off += (size_t)blockIdx.x * ((size_t)n + 1);
    if (cur) cur += (size_t)blockIdx.x * ((size_t)n + 1);
    int s = 0;
    for (int k = 0; k < per; ++k) {
        const int i = tid * per + k;
        if (i < n) {
            s += cnt[i];
            if (cap > 0 && cnt[i] > cap) atomicOr(err, 2);
            if (cap > 0) atomicMax(err + 1, cnt[i]);            // longest row: sizes the key kernel's LDS
            if (cur) cur[i] = 0;
        }
    }
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int k = 0; k < per; ++k) {
        const int i = tid * per + k;
        if (i < n) { off[i] = run; run += cnt[i]; }
    }
    if (tid == 1023) off[n] = part[1023];
}

// (entries land in their row in no particular order: k_sp_rowsort orders every row afterwards)
__global__ void __launch_bounds__(kSpChunkThreads) k_sp_fill(const int64_t* __restrict__ ei, int64_t E, int n,
                                                             const int* __restrict__ off, int* __restrict__ cur,
                                                             uint16_t* __restrict__ ent_j, int* __restrict__ ent_e, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* hist = reinterpret_cast<int*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, T = blockDim.x;
    ei += (size_t)blockIdx.y * 2 * (size_t)E;
    off += (size_t)blockIdx.y * ((size_t)n + 1);
    cur += (size_t)blockIdx.y * ((size_t)n + 1);
    ent_j += (size_t)blockIdx.y * 2 * (size_t)E;
    ent_e += (size_t)blockIdx.y * 2 * (size_t)E;
    const int64_t e0 = (int64_t)blockIdx.x * chunk, e1 = e0 + chunk < E ? e0 + chunk : E;
    for (int i = tid; i < n; i += T) hist[i] = 0;
    __syncthreads();
    // this chunk's entries per row ...
    for (int64_t eb = e0; eb < e1; eb += T) {
        int64_t a, b;
        const bool valid = sp_edge(ei, E, n, eb + tid, e1, a, b, nullptr);
        const SrcRun r = src_run(valid ? (int)a : -1 - lane, lane);
        if (valid) {
            if (r.head == lane) atomicAdd(&hist[a], r.len);
            atomicAdd(&hist[b], 1);
        }
    }
    __syncthreads();
    // ... reserve that many places of the row: the table now holds the chunk's first place in every row ...
    for (int i = tid; i < n; i += T) {
        const int c = hist[i];
        hist[i] = c ? off[i] + atomicAdd(&cur[i], c) : 0;
    }
    __syncthreads();
    // ... and hand them out
    for (int64_t eb = e0; eb < e1; eb += T) {
        const int64_t e = eb + tid;
        int64_t a, b;
        const bool valid = sp_edge(ei, E, n, e, e1, a, b, nullptr);
        const SrcRun r = src_run(valid ? (int)a : -1 - lane, lane);
        int base = 0;
        if (valid && r.head == lane) base = atomicAdd(&hist[a], r.len);
        base = __shfl(base, r.head);
        if (valid) {
            int pos = base + (lane - r.head);
            ent_j[pos] = (uint16_t)b;
            ent_e[pos] = (int)(2 * e);                  // row a holds (a, b): forward, the first coo matrix
            pos = atomicAdd(&hist[b], 1);
            ent_j[pos] = (uint16_t)a;
            ent_e[pos] = (int)(2 * e + 1);              // row b holds the same edge transposed: the second
        }
    }
}

// one wave per row: entries ordered by (partner, edge id, direction); ucnt[i] = distinct partners, and the
// first entry of every partner carries bit 31 of its srt_e word (edge ids stay below 2^30).  The row is staged
// in LDS as 64-bit keys (partner << 32 | edge id and direction); an entry's rank is the number of smaller keys,
// and it is a partner's first entry iff none of them has the same partner.
constexpr int kHeadBit = (int)0x80000000u;
__global__ void __launch_bounds__(64) k_sp_rowsort(const int* __restrict__ off, const uint16_t* __restrict__ ent_j,
                                                   const int* __restrict__ ent_e, uint16_t* __restrict__ srt_j,
                                                   int* __restrict__ srt_e, int* __restrict__ ucnt, int n, int64_t E)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* ks = reinterpret_cast<unsigned long long*>(smem);
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    off += (size_t)blockIdx.y * ((size_t)n + 1);
    ucnt += (size_t)blockIdx.y * ((size_t)n + 1);
    ent_j += (size_t)blockIdx.y * 2 * (size_t)E;
    ent_e += (size_t)blockIdx.y * 2 * (size_t)E;
    srt_j += (size_t)blockIdx.y * 2 * (size_t)E;
    srt_e += (size_t)blockIdx.y * 2 * (size_t)E;
    const int o = off[i], L = off[i + 1] - o;
    for (int x = lane; x < L; x += 64) ks[x] = ((unsigned long long)ent_j[o + x] << 32) | (unsigned)ent_e[o + x];
    __syncthreads();
    int heads = 0;
    for (int x0 = 0; x0 < L; x0 += 64) {
        const int x = x0 + lane;
        bool head = false;
        if (x < L) {
            const unsigned long long kx = ks[x];
            const unsigned jx = (unsigned)(kx >> 32);
            int rank = 0, same = 0;
            for (int y = 0; y < L; ++y) {               // every lane reads the same word: an LDS broadcast
                const unsigned long long ky = ks[y];
                const int lt = ky < kx ? 1 : 0;
                rank += lt;
                same += (lt && (unsigned)(ky >> 32) == jx) ? 1 : 0;
            }
            head = same == 0;
            srt_j[o + rank] = (uint16_t)jx;
            srt_e[o + rank] = (int)(unsigned)kx | (head ? kHeadBit : 0);
        }
        heads += __popcll(__ballot(head));
    }
    if (lane == 0) ucnt[i] = heads;
}

// ---- per heat map: sums per partner, keys, candidate lists ---------------------------------------------
// one wave per (row, instance), a lane per distinct partner; what the heat maps of an instance share -- where
// each partner's entries start, the partner, its distance -- is prepared once, then every heat map of the
// instance sums its values, keys them and rank-sorts the partners with S > 0 by exact (key, j) into
// list_j / list_k [map][uoff[i] + rank].
template <typename H>
__global__ void __launch_bounds__(64) k_sp_rowkeys(const H* __restrict__ vals, const double* __restrict__ points,
                                                   const int* __restrict__ off, const uint16_t* __restrict__ srt_j,
                                                   const int* __restrict__ srt_e, const int* __restrict__ uoff,
                                                   uint16_t* __restrict__ list_j, float* __restrict__ list_k,
                                                   int* __restrict__ len, int64_t num_maps, int n, int64_t E, int mpi,
                                                   int cap, int* __restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int i = blockIdx.x, lane = threadIdx.x;
    const size_t inst = blockIdx.y, E2 = 2 * (size_t)E;
    off += inst * ((size_t)n + 1);
    uoff += inst * ((size_t)n + 1);
    srt_j += inst * E2;
    srt_e += inst * E2;
    const int o = off[i], L = off[i + 1] - o;
    const int ub = uoff[i];
    double* ck = reinterpret_cast<double*>(smem);                       // kept partners of one map, compact: key,
    double* pd = ck + cap;                                              // ... every partner's distance,
    uint16_t* cj = reinterpret_cast<uint16_t*>(pd + cap);               // kept partner's node,
    uint16_t* hd = cj + cap;                                            // first entry of the h-th partner,
    uint16_t* pj = hd + cap;                                            // its node
    const double* P = points + inst * (size_t)n * 2;
    const double xi = P[2 * i], yi = P[2 * i + 1];
    int U = 0;
    for (int x0 = 0; x0 < L; x0 += 64) {
        const int x = x0 + lane;
        const bool head = x < L && srt_e[o + x] < 0;
        const unsigned long long bal = __ballot(head);
        if (head) {
            const int h = U + __popcll(bal & ((1ull << lane) - 1ull));
            const int j = srt_j[o + x];
            const double dx = xi - P[2 * j], dy = yi - P[2 * j + 1];
            hd[h] = (uint16_t)x;
            pj[h] = (uint16_t)j;
            pd[h] = sqrt(dx * dx + dy * dy);
        }
        U += __popcll(bal);
    }
    __syncthreads();
    for (int64_t m = (int64_t)inst * mpi; m < (int64_t)(inst + 1) * mpi && m < num_maps; ++m) {
        const H* v = vals + (size_t)m * (size_t)E;
        int K = 0;
        for (int h0 = 0; h0 < U; h0 += 64) {
            const int h = h0 + lane;
            double key = INFINITY;
            if (h < U) {
                const int xs = hd[h], xe = h + 1 < U ? (int)hd[h + 1] : L;
                H fwd = (H)0, bwd = (H)0;
                for (int y = xs; y < xe; ++y) {                         // the entries of one pair, by edge id
                    const int e2 = srt_e[o + y] & ~kHeadBit;
                    const H w = v[e2 >> 1];
                    if (e2 & 1) bwd += w; else fwd += w;
                }
                const H S = fwd + bwd;                                  // coo(i,j).toarray() + coo(j,i).toarray()
                if (S < (H)0 || S != S) atomicOr(err, 4);
                if (S > (H)0) {
                    key = -(double)S / pd[h];
                    if (key != key) key = INFINITY;
                }
            }
            const bool keep = key != INFINITY;
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const int c = K + __popcll(bal & ((1ull << lane) - 1ull));
                ck[c] = key;
                cj[c] = pj[h];
            }
            K += __popcll(bal);
        }
        __syncthreads();
        uint16_t* outJ = list_j + (size_t)m * E2 + ub;
        float* outK = list_k + (size_t)m * E2 + ub;
        for (int c = lane; c < K; c += 64) {
            const double kc = ck[c];
            const unsigned jc = cj[c];
            int rank = 0;
            for (int y = 0; y < K; ++y) {
                const double ky = ck[y];
                rank += (ky < kc || (ky == kc && (unsigned)cj[y] < jc)) ? 1 : 0;
            }
            outJ[rank] = (uint16_t)jc;
            outK[rank] = (float)kc;                                     // what the merge loop compares first
        }
        if (lane == 0) len[(size_t)m * n + i] = K;
        __syncthreads();                                                // ck / cj are reused by the next heat map
    }
}

// ---- merge loop ------------------------------------------------------------------------------------------
// LDS state, 15 B per node: everything a pass needs except the list entries themselves, so that a merge
// touches no global memory on its critical path and a cursor advance waits for one (coalesced) list read
struct SpLds {
    float* key;             // fp32 rounding of the current candidate's key; NaN = none
    unsigned* pos;          // position of the candidate in this heat map's lists
    uint16_t* cj;           // the candidate's partner
    uint16_t* oend;         // other end of the fragment a fragment end belongs to
    uint16_t* rem;          // entries of the node's list behind the candidate
    uint8_t* deg;
};

template <typename H>
struct SpMap {
    const uint16_t* lj;     // this heat map's lists
    const float* lk;
    const double* P;
    const H* v;             // for the exact keys of tied candidates: the heat values and the instance's rows
    const int* off;
    const uint16_t* sj;
    const int* se;
    int n;
};

template <typename H>
__device__ __forceinline__ double sp_key(const SpMap<H>& M, int i, int j, H S)
{
    const double dx = M.P[2 * i] - M.P[2 * j], dy = M.P[2 * i + 1] - M.P[2 * j + 1];
    return -(double)S / sqrt(dx * dx + dy * dy);
}

// exact key of the listed pair (i, j): its sum rebuilt from row i's entries exactly as k_sp_rowkeys adds them
template <typename H>
__device__ double sp_exact_key(const SpMap<H>& M, int i, int j)
{
    const int o = M.off[i], L = M.off[i + 1] - o;
    int lo = 0, hi = L;
    while (lo < hi) {                                   // first entry of partner j (rows are sorted by partner)
        const int mid = (lo + hi) >> 1;
        if ((int)M.sj[o + mid] < j) lo = mid + 1; else hi = mid;
    }
    H fwd = (H)0, bwd = (H)0;
    for (int y = lo; y < L && (int)M.sj[o + y] == j; ++y) {
        const int e2 = M.se[o + y] & ~kHeadBit;
        const H w = M.v[e2 >> 1];
        if (e2 & 1) bwd += w; else fwd += w;
    }
    return sp_key(M, i, j, (H)(fwd + bwd));
}

__device__ __forceinline__ bool sp_ok(const SpLds& L, int i, int oe, int j)
{
    return L.deg[j] < 2 && !(j == oe && oe != i);
}

// node i's candidate is gone: the next valid entry of its list, or none.  One thread per node: the partners
// are fetched four at a time so that a run of dead entries costs one memory latency per four
template <typename H>
__device__ __forceinline__ void sp_advance(const SpLds& L, const SpMap<H>& M, int i, int oe)
{
    unsigned p = L.pos[i] + 1;
    int rem = L.rem[i];
    while (rem > 0) {
        int j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = u < rem ? (int)M.lj[p + u] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < rem && sp_ok(L, i, oe, j[u])) {
                L.pos[i] = p + (unsigned)u;
                L.rem[i] = (uint16_t)(rem - u - 1);
                L.cj[i] = (uint16_t)j[u];
                L.key[i] = M.lk[p + u];
                return;
            }
        }
        p += 4;
        rem -= 4;
    }
    L.rem[i] = 0;
    L.key[i] = __builtin_nanf("");
}

__device__ __forceinline__ void sp_link(const SpLds& L, uint16_t* nb, int i, int j)
{
    const int oi = L.oend[i], oj = L.oend[j];
    const int di = L.deg[i], dj = L.deg[j];
    nb[2 * i + di] = (uint16_t)j;               // slot = degree before the merge: no read of the table
    nb[2 * j + dj] = (uint16_t)i;
    L.deg[i] = (uint8_t)(di + 1);
    L.deg[j] = (uint8_t)(dj + 1);
    if (di >= 1) L.key[i] = __builtin_nanf("");
    if (dj >= 1) L.key[j] = __builtin_nanf("");
    L.oend[oi] = (uint16_t)oj;                  // the merged fragment runs from oi to oj
    L.oend[oj] = (uint16_t)oi;
}

// One workgroup per heat map.  The sequential rule "insert the valid edge with the smallest (key, flat index)"
// is carried out in ROUNDS that insert many edges at once.  After every node's candidate has been moved to its
// first valid entry, an edge f = (i, j) is inserted in this round when
//   * it is the candidate of both i and j (no valid edge at i or j precedes it: the lists are exactly ordered), and
//   * the far ends a, b of the two fragments it joins hold candidates with strictly larger fp32 keys (or none):
//     then nothing can happen at i, j, a or b before f's turn, so f is valid at its turn, and the only edge
//     whose validity f changes, (a, b), comes after it -- moving f to the front changes no other decision.
// Edges chosen in one round join disjoint pairs of fragments, so a thread each applies them.  The valid edge
// with the globally smallest key always qualifies unless another candidate has the same fp32 key; a round
// that finds nothing therefore inserts exactly that edge after comparing the tied ones exactly (double key,
// then flat index), the serial rule.  Cursor advances (the only global-memory reads of the loop) run one
// node per thread, so their latencies overlap.
template <typename H>
__global__ void __launch_bounds__(1024) k_greedy_sparse(const uint16_t* __restrict__ list_j, const float* __restrict__ list_k,
                                                        const H* __restrict__ vals, const int* __restrict__ off,
                                                        const uint16_t* __restrict__ srt_j, const int* __restrict__ srt_e,
                                                        const int* __restrict__ len, const int* __restrict__ uoff,
                                                        const double* __restrict__ points, uint16_t* __restrict__ nbg,
                                                        int64_t* __restrict__ tours, int64_t num_maps, int n, int64_t E,
                                                        int mpi, unsigned long long* __restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red_f[16];
    __shared__ double red_e[16];
    __shared__ unsigned red_i[16];
    __shared__ int ctl[2];                              // edges chosen this round; any candidate left
    const int64_t b = blockIdx.x;
    if (b >= num_maps) return;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    const int G = (n + 63) >> 6;
    const size_t inst = (size_t)(b / mpi), E2 = 2 * (size_t)E;
    uoff += inst * ((size_t)n + 1);
    const unsigned long long t0 = wall_clock64();
    unsigned long long rounds = 0, serial_rounds = 0;
    SpLds L;
    size_t o = 0;
    L.key = reinterpret_cast<float*>(smem + o); o += sizeof(float) * (size_t)n;
    L.pos = reinterpret_cast<unsigned*>(smem + o); o += sizeof(unsigned) * (size_t)n;
    L.cj = reinterpret_cast<uint16_t*>(smem + o); o += sizeof(uint16_t) * (size_t)n;
    L.oend = reinterpret_cast<uint16_t*>(smem + o); o += sizeof(uint16_t) * (size_t)n;
    L.rem = reinterpret_cast<uint16_t*>(smem + o); o += sizeof(uint16_t) * (size_t)n;
    L.deg = reinterpret_cast<uint8_t*>(smem + o);
    SpMap<H> M{list_j + (size_t)b * E2, list_k + (size_t)b * E2, points + inst * (size_t)n * 2,
               vals + (size_t)b * (size_t)E, off + inst * ((size_t)n + 1), srt_j + inst * E2, srt_e + inst * E2, n};
    const int* lenm = len + (size_t)b * n;
    uint16_t* nb = nbg + (size_t)b * 2 * (size_t)n;
    for (int i = tid; i < n; i += T) {
        L.oend[i] = (uint16_t)i;
        L.deg[i] = 0;
        nb[2 * i] = nb[2 * i + 1] = 0xffffu;
        const int li = lenm[i];
        const unsigned base = (unsigned)uoff[i];
        L.pos[i] = base;
        L.rem[i] = (uint16_t)(li > 0 ? li - 1 : 0);
        L.key[i] = li > 0 ? M.lk[base] : __builtin_nanf("");       // first candidates: the heads of the lists
        L.cj[i] = li > 0 ? M.lj[base] : (uint16_t)0;
    }
    if (tid == 0) ctl[0] = ctl[1] = 0;
    __syncthreads();

    int merges = 0;
    const unsigned long long t1 = wall_clock64();
    // ---- listed edges
    for (int guard = 0; merges < n - 1 && guard < n + 8; ++guard) {
        // (1) every open node's candidate becomes its first valid entry
        if (tid == 0) ctl[0] = ctl[1] = 0;              // last read before the barrier that ended the previous round
        for (int i = tid; i < n; i += T) {
            const float k = L.key[i];
            if (k != k) continue;                       // none (also: degree 2)
            const int oe = L.oend[i];
            if (!sp_ok(L, i, oe, L.cj[i])) sp_advance(L, M, i, oe);
        }
        __syncthreads();
        // (2) choose: the lower endpoint of an edge decides and marks itself (top bit of rem)
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
            if (a != i && L.key[a] <= k) continue;      // NaN (no candidate) compares false: passes
            if (c != j && L.key[c] <= k) continue;
            L.rem[i] |= 0x8000u;
            ++mine;
        }
        if (any) ctl[1] = 1;
        if (mine) atomicAdd(&ctl[0], mine);
        __syncthreads();
        const int chosen = ctl[0];
        if (!ctl[1]) break;                             // no listed edge is valid any more
        ++rounds;
        if (chosen) {
            // (3) apply: disjoint fragments, one thread per edge
            for (int i = tid; i < n; i += T) {
                const unsigned r = L.rem[i];
                if (r & 0x8000u) {
                    L.rem[i] = (uint16_t)(r & 0x7fffu);
                    sp_link(L, nb, i, L.cj[i]);
                }
            }
            merges += chosen;
        } else {
            // the smallest fp32 key is shared by different edges: exact keys from the listed sums, then the
            // flat index -- one edge, like the serial scan
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
            double bk = INFINITY;
            unsigned bf = 0xffffffffu;
            for (int i = tid; i < n; i += T) {
                if (!(L.key[i] == m)) continue;
                const unsigned cj = L.cj[i];
                const unsigned f = (unsigned)i * (unsigned)n + cj;
                const double key = sp_exact_key(M, i, (int)cj);
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
                    sp_link(L, nb, i, (int)(bf - (unsigned)i * (unsigned)n));
                }
            }
            ++merges;
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- the pairs without a positive sum.  Their key is -0.0 / d: -0.0, all tied, the lower flat index
    // i*n + j first -- except where d is not a positive number (coincident points, squares that underflow,
    // NaN coordinates): there it is NaN, which numpy sorts after everything else (tied again, by index).  Two
    // passes, zero keys then NaN keys; in each the smallest open node that still has a partner of the pass
    // takes its smallest one.  Both only ever move up: a node left without a partner stays without one for the
    // rest of the pass (open nodes only disappear, and a fragment's new far end was open before), so one
    // ascending sweep over the open nodes serves all merges of a pass.
    const unsigned long long t2 = wall_clock64();
    const int listed_merges = merges;
    if (tid < 64) {                                     // short and serial: wave 0
        for (int pass = 0; pass < 2 && merges < n - 1; ++pass) {
            int i = -1, from = 0;                           // current node; nodes below `from` are done for this pass
            while (merges < n - 1) {
                if (i < 0 || L.deg[i] >= 2) {               // (re)locate the smallest open node at or above `from`
                    i = -1;
                    for (int g = from >> 6; g < G; ++g) {
                        const int k = (g << 6) + lane;
                        const unsigned long long bal = __ballot(k < n && k >= from && L.deg[k] < 2);
                        if (bal) { i = (g << 6) + __ffsll((long long)bal) - 1; break; }
                    }
                    if (i < 0) break;
                    from = i;
                }
                const int oe = L.oend[i];
                const double xi = M.P[2 * i], yi = M.P[2 * i + 1];
                int j = -1;
                for (int g = i >> 6; g < G && j < 0; ++g) {
                    const int k = (g << 6) + lane;
                    unsigned long long bal = __ballot(k < n && k > i && L.deg[k] < 2 && !(k == oe && oe != i));
                    while (bal) {                           // (wave-uniform: every lane tests the same candidate)
                        const int c = (g << 6) + __ffsll((long long)bal) - 1;
                        bal &= bal - 1;
                        const double dx = xi - M.P[2 * c], dy = yi - M.P[2 * c + 1];
                        const bool nankey = !(dx * dx + dy * dy > 0.0);     // sqrt(0) = 0 and -0.0 / 0 is NaN
                        if (nankey == (pass == 1)) { j = c; break; }
                    }
                }
                if (j < 0) {                                // nothing for i in this pass: on to the next open node
                    from = i + 1;
                    i = -1;
                    continue;
                }
                if (lane == 0) sp_link(L, nb, i, j);
                ++merges;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) ctl[0] = merges;
    }
    __syncthreads();
    merges = ctl[0];
    const unsigned long long t3 = wall_clock64();
    // closing edge between the two remaining ends, then the walk of greedy_search.py:65-73 over an LDS copy
    // of the neighbour table: a parallel list ranking (greedy_walk.hpp), or -- should the merge ever end
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
        if (lane == 0) { red_i[0] = (unsigned)a; red_i[1] = (unsigned)c; }
    }
    __syncthreads();
    const int a = (int)red_i[0], c = (int)red_i[1];
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
    const bool par = __syncthreads_and(whole ? 1 : 0) && merges == n - 1 && walk_parallel_fits(n, T);   // (uniform)
    if (par) {
        walk_parallel(nbl, L.pos, n, tid, T, t);           // pos, cj, oend: 8 B per node, all dead
    } else if (lane < 2 && tid < 64) {
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
    if (b == 0 && tid == 0 && dbg) {                    // diagnostics of heat map 0 (10 ns ticks, counts)
        const unsigned long long t4 = wall_clock64();
        dbg[0] = t1 - t0; dbg[1] = t2 - t1; dbg[2] = t3 - t2; dbg[3] = t4 - t3;
        dbg[4] = rounds; dbg[5] = serial_rounds; dbg[6] = 0; dbg[7] = (unsigned long long)listed_merges;
    }
}

// ---- host ---------------------------------------------------------------------------------------------------
struct SpWs {
    size_t off_err, off_cnt, off_off, off_cur, off_ucnt, off_uoff, off_srtj, off_srte, off_len, off_nb, off_lists,
        off_lk, off_entj, off_ente, total;
};

// the unsorted entries (dead once the rows are sorted) share their memory with the candidate lists (written
// afterwards): 12 B per listed edge and instance against 12 B per listed edge and heat map
SpWs sp_ws(int64_t num_maps, int64_t inst, int n, int64_t E)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    SpWs w{};
    size_t o = 0;
    const size_t N1 = ((size_t)n + 1) * (size_t)inst, E2 = 2 * (size_t)E;
    w.off_err = o; o += 256;
    w.off_cnt = o; o += up(N1 * 4);
    w.off_off = o; o += up(N1 * 4);
    w.off_cur = o; o += up(N1 * 4);
    w.off_ucnt = o; o += up(N1 * 4);
    w.off_uoff = o; o += up(N1 * 4);
    w.off_srtj = o; o += up(E2 * inst * 2);
    w.off_srte = o; o += up(E2 * inst * 4);
    w.off_len = o; o += up((size_t)num_maps * n * 4);
    w.off_nb = o; o += up((size_t)num_maps * n * 4);
    w.off_lists = o;
    w.off_lk = o + up((size_t)num_maps * E2 * 2);                       // list_j, then list_k
    const size_t lists = up((size_t)num_maps * E2 * 2) + up((size_t)num_maps * E2 * 4);
    w.off_entj = o;
    w.off_ente = o + up(E2 * inst * 2);
    const size_t ents = up(E2 * inst * 2) + up(E2 * inst * 4);
    o += std::max(lists, ents);
    w.total = o;
    return w;
}

template <typename H>
int greedy_sparse_t(const void* values, const int64_t* ei, const double* points, int64_t* tours, int64_t num_maps,
                    int n, int64_t E, int mpi, void* ws, hipStream_t stream)
{
    const int64_t inst = (num_maps + mpi - 1) / mpi;
    const SpWs wl = sp_ws(num_maps, inst, n, E);
    unsigned char* w8 = static_cast<unsigned char*>(ws);
    int* err = reinterpret_cast<int*>(w8 + wl.off_err);
    int* cnt = reinterpret_cast<int*>(w8 + wl.off_cnt);
    int* off = reinterpret_cast<int*>(w8 + wl.off_off);
    int* cur = reinterpret_cast<int*>(w8 + wl.off_cur);
    int* ucnt = reinterpret_cast<int*>(w8 + wl.off_ucnt);
    int* uoff = reinterpret_cast<int*>(w8 + wl.off_uoff);
    uint16_t* entj = reinterpret_cast<uint16_t*>(w8 + wl.off_entj);
    int* ente = reinterpret_cast<int*>(w8 + wl.off_ente);
    uint16_t* srtj = reinterpret_cast<uint16_t*>(w8 + wl.off_srtj);
    int* srte = reinterpret_cast<int*>(w8 + wl.off_srte);
    int* len = reinterpret_cast<int*>(w8 + wl.off_len);
    uint16_t* nb = reinterpret_cast<uint16_t*>(w8 + wl.off_nb);
    uint16_t* lj = reinterpret_cast<uint16_t*>(w8 + wl.off_lists);
    float* lk = reinterpret_cast<float*>(w8 + wl.off_lk);
    HIP_TRY(hipMemsetAsync(w8, 0, wl.off_off, stream));                 // err + cnt
    // chunks of at least 4 n edges (a chunk pays n table entries on entry and on exit), at most 65536
    const int chunk = (int)std::min<int64_t>(65536, std::max<int64_t>(4 * (int64_t)n, 8192));
    const unsigned eg = (unsigned)std::max<int64_t>((E + chunk - 1) / chunk, 1);   // E = 0: one idle block
    const unsigned ig = (unsigned)inst;
    const size_t hlds = sizeof(int) * (size_t)n;
    hipLaunchKernelGGL(k_sp_count, dim3(eg, ig), dim3(kSpChunkThreads), hlds, stream, ei, E, n, cnt, err, chunk);
    hipLaunchKernelGGL(k_sp_scan, dim3(ig), dim3(1024), 0, stream, cnt, off, cur, n, kSpMaxRow, err);
    // the error word and the longest row (it sizes the LDS of the two row kernels) are known after the scan
    int eh[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(eh, err, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    int errh = eh[0];
    if (errh & 1) return fail(TSP2OPT_E_ARG, "edge_index has an entry outside [0, n)");
    if (errh & 2) return fail(TSP2OPT_E_UNSUPPORTED, "a node has more than 7000 listed edges: use the dense decoder");
    const int cap = std::max(eh[1], 1);
    const size_t slds = ((size_t)cap * 8 + 15) & ~(size_t)15;
    const size_t klds = ((size_t)cap * 22 + 15) & ~(size_t)15;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sp_rowsort),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sp_rowkeys<H>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)klds));
    hipLaunchKernelGGL(k_sp_fill, dim3(eg, ig), dim3(kSpChunkThreads), hlds, stream, ei, E, n, off, cur, entj, ente, chunk);
    hipLaunchKernelGGL(k_sp_rowsort, dim3((unsigned)n, ig), dim3(64), slds, stream, off, entj, ente, srtj, srte, ucnt, n, E);
    hipLaunchKernelGGL(k_sp_scan, dim3(ig), dim3(1024), 0, stream, ucnt, uoff, (int*)nullptr, n, 0, err);
    hipLaunchKernelGGL(k_sp_rowkeys<H>, dim3((unsigned)n, ig), dim3(64), klds, stream,
                       static_cast<const H*>(values), points, off, srtj, srte, uoff, lj, lk, len, num_maps, n, E, mpi,
                       cap, err);
    HIP_TRY(hipGetLastError());
    const size_t lds = (size_t)n * 15 + 64;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_greedy_sparse<H>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_greedy_sparse<H>, dim3((unsigned)num_maps), dim3((n > 2048 || (n > 512 && num_maps <= 512)) ? 1024 : 256), lds, stream, lj, lk,
                       static_cast<const H*>(values), off, srtj, srte, len, uoff, points,
                       nb, tours, num_maps, n, E, mpi, reinterpret_cast<unsigned long long*>(err + 16));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&errh, err, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (errh & 4)
        return fail(TSP2OPT_E_UNSUPPORTED, "a listed pair has a negative (or NaN) heat sum: its key sorts after the "
                                           "unlisted pairs; use the dense decoder");
    return 0;
}

}  // namespace t2o

using namespace t2o;

extern "C" int64_t tsp2opt_greedy_sparse_workspace_bytes(int64_t num_maps, int32_t n, int64_t num_edges,
                                                         int32_t maps_per_instance, int32_t heat_dtype)
{
    if (num_maps < 0 || n < 2 || num_edges < 0 || maps_per_instance < 1) return fail(TSP2OPT_E_ARG, "bad size");
    if (heat_dtype != TSP2OPT_DTYPE_F32 && heat_dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    const int64_t inst = (num_maps + maps_per_instance - 1) / maps_per_instance;
    return (int64_t)sp_ws(num_maps, inst, n, num_edges).total;
}

extern "C" int tsp2opt_greedy_decode_sparse(const void* values, const int64_t* edge_index, const double* points,
                                            int64_t* tours_out, int64_t num_maps, int32_t n, int64_t num_edges,
                                            int32_t maps_per_instance, int32_t heat_dtype, int32_t device,
                                            void* workspace, int64_t workspace_bytes, void* hip_stream)
{
    if (num_maps < 0 || n < 2 || maps_per_instance < 1 || num_edges < 0) return fail(TSP2OPT_E_ARG, "bad size");
    if (n > kSpMaxN) return fail(TSP2OPT_E_UNSUPPORTED, "sparse greedy decode supports n <= 10900");
    if (num_edges > 0x3fffffffll) return fail(TSP2OPT_E_UNSUPPORTED, "too many edges");
    if (num_maps > 65535) return fail(TSP2OPT_E_UNSUPPORTED, "more than 65535 heat maps per call");
    if (num_edges * ((num_maps + maps_per_instance - 1) / maps_per_instance) > 0x3fffffffll)
        return fail(TSP2OPT_E_UNSUPPORTED, "too many edges in one call");
    if (heat_dtype != TSP2OPT_DTYPE_F32 && heat_dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    if (num_maps == 0) return 0;
    if (!points || !tours_out || !workspace || (num_edges > 0 && (!values || !edge_index)))
        return fail(TSP2OPT_E_ARG, "null pointer");
    if (workspace_bytes < tsp2opt_greedy_sparse_workspace_bytes(num_maps, n, num_edges, maps_per_instance, heat_dtype))
        return fail(TSP2OPT_E_WORKSPACE, "workspace too small");
    if ((reinterpret_cast<uintptr_t>(values) % (heat_dtype == TSP2OPT_DTYPE_F64 ? 8 : 4)) ||
        (reinterpret_cast<uintptr_t>(points) % 8) || (reinterpret_cast<uintptr_t>(tours_out) % 8) ||
        (reinterpret_cast<uintptr_t>(edge_index) % 8))
        return fail(TSP2OPT_E_ARG, "misaligned pointer (arrays must be aligned to their element size)");
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(TSP2OPT_E_ARG, "workspace must be 256-byte aligned");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (heat_dtype == TSP2OPT_DTYPE_F64)
        return greedy_sparse_t<double>(values, edge_index, points, tours_out, num_maps, n, num_edges,
                                       maps_per_instance, workspace, stream);
    return greedy_sparse_t<float>(values, edge_index, points, tours_out, num_maps, n, num_edges, maps_per_instance,
                                  workspace, stream);
}
