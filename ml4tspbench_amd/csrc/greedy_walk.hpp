// ml4tspbench_amd/csrc/greedy_walk.hpp -- the closing walk of the greedy decoders as a parallel list ranking
//
// Replaces (reference, read-only): search/greedy_search.py:65-73 -- from node 0, never back to the previous node,
// the larger neighbour first: an N-step pointer chase over the neighbour table.  Here every (node, arrived-from
// slot) pair is a state with one successor; pointer doubling gives each state its distance to node 0 in
// ceil(log2 N) rounds, and a node's position in the reference's walk is the distance of its state in the
// direction that reaches node 0 THROUGH the walk's first node (the larger neighbour of node 0).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace t2o {

constexpr int kWalkPer = 20;    // states a thread keeps in registers between the two barriers of a round

// can walk_parallel take this workgroup?  (2 n states over T threads)
__device__ __forceinline__ bool walk_parallel_fits(int n, int T) { return n >= 3 && 2 * n <= kWalkPer * T; }

// All T threads of the workgroup call.  nbl[k] = lo | hi << 16: the two neighbours of node k on a Hamiltonian
// cycle of n >= 3 nodes (no empty slot); S: 2 n words of LDS scratch; t: the tour, n + 1 entries.
__device__ __forceinline__ void walk_parallel(const unsigned* nbl, unsigned* S, int n, int tid, int T,
                                              int64_t* __restrict__ t)
{
    // state s = 2 v + p: at node v, having arrived from its neighbour slot p; word = successor | distance << 16;
    // the two states of node 0 are terminal (distance 0, their own successors)
    for (int s = tid; s < 2 * n; s += T) {
        const int v = s >> 1, p = s & 1;
        const unsigned w = nbl[v];
        const unsigned nxt = p ? (w & 0xffffu) : (w >> 16);         // the neighbour in the OTHER slot
        const unsigned p2 = (nbl[nxt] & 0xffffu) == (unsigned)v ? 0u : 1u;
        S[s] = v == 0 ? (unsigned)s : ((2u * nxt + p2) | (1u << 16));
    }
    __syncthreads();
    for (int span = 1; span < n - 1; span <<= 1) {
        unsigned nw[kWalkPer];
#pragma unroll
        for (int q = 0; q < kWalkPer; ++q) {
            const int s = tid + q * T;
            if (s < 2 * n) {
                const unsigned a = S[s];
                const unsigned b = S[a & 0xffffu];
                nw[q] = (b & 0xffffu) | ((a & 0xffff0000u) + (b & 0xffff0000u));
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kWalkPer; ++q) {
            const int s = tid + q * T;
            if (s < 2 * n) S[s] = nw[q];
        }
        __syncthreads();
    }
    const unsigned w0 = nbl[0];
    const unsigned through = (w0 >> 16) > (w0 & 0xffffu) ? 1u : 0u;     // slot of node 0's larger neighbour
    for (int v = tid; v < n; v += T) {
        if (v == 0) { t[0] = 0; t[n] = 0; continue; }
        const unsigned s0 = S[2 * v], s1 = S[2 * v + 1];
        const unsigned pos = (s0 & 0xffffu) == through ? (s0 >> 16) : (s1 >> 16);
        t[pos] = v;
    }
}

}  // namespace t2o
