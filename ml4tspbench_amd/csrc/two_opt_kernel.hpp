// ml4tspbench_amd/csrc/two_opt_kernel.hpp -- filter sweep, incremental sweeps, argmin selection, reversal, cluster exchange, k_two_opt
// Part of the single translation unit tsp2opt.hip (design notes at its top and in DESIGN.md).
#pragma once

namespace t2o {

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

// bits [a, b] of a 32-bit mask, clipped to [0, M-1] (M <= 32)
__device__ __forceinline__ unsigned bit_range32(int a, int b, int M)
{
    a = a < 0 ? 0 : a;
    b = b > M - 1 ? M - 1 : b;
    if (a > b) return 0u;
    return (~0u >> (31 - b)) & (~0u << a);
}

struct Unit {               // the lane's own unit, fixed for the whole kernel
    int gp;                 // lane group (of the whole tour), or -1 when the lane owns no unit
    int gpl;                // its index among this workgroup's lane groups
    int rq0;                // first 8-row block of its row segment
};

// Recomputed where it is needed rather than carried through the iteration loop: the kernel sits at
// the VGPR limit and loop-carried registers are the ones that get spilled.
template <bool CL>
__device__ __forceinline__ Unit lane_unit(const Geo& g, int tid)
{
    Unit un{-1, 0, 0};
    if (tid < g.GPl * g.NS) {
        un.gpl = tid % g.GPl;
        un.rq0 = (tid / g.GPl) * (g.MS >> 1);
        const int gp = CL ? g.wc + g.CS * un.gpl : un.gpl;
        if (!CL || gp < g.GP) un.gp = gp;
    }
    return un;
}

// Filter sweep of one iteration.  `last` = the reversal applied since the previous sweep (packed
// (i, j)) or kFullSweep.  With the tile cache, only tiles that read a changed window position
// (coordinates i+1..j, edges i..j, any cyclic copy) are recomputed: every lane marks its own
// touched tiles in closed form, the ids are compacted into a list (ballot per row block, one LDS
// atomic per wave), the list is processed evenly by all lanes, then every lane reduces its own
// cached tile minima exactly as the full sweep would.
template <int Q, bool CL, typename T>
__device__ __forceinline__ void sweep(const View<T>& v, const Geo& g, int tid,
                                      unsigned last, float& a1, float& a2, unsigned& chunk, Prof& pf)
{
    Ctl* ctl = v.ctl;
    const Unit un = lane_unit<CL>(g, tid);
    bool full = !g.inc || last == kFullSweep;
    FINE_START(pf);
    if (!full) {
        const int MS2 = g.MS >> 1;
        const int i = (int)(last >> 16), j = (int)(last & 0xffffu);
        unsigned mask = 0u;                                 // touched tiles of the lane's unit (MS2 <= 32)
        if (un.gp >= 0) {
            for (int lo = i, hi = j; lo < g.W; lo += g.n, hi += g.n) {
                mask |= bit_range32((lo >> 3) - un.rq0, (hi >> 3) - un.rq0, MS2);                              // rows
                mask |= bit_range32(((lo - 9) >> 3) - un.gp - un.rq0, ((hi - 2) >> 3) - un.gp - un.rq0, MS2);  // columns
            }
        }
        // compaction: prefix sum of the lanes' counts -> one atomic per wave -> every lane writes its run
        const unsigned cnt = (unsigned)__popc(mask);
        const unsigned incl = wave_scan_add(cnt);
        const unsigned wtotal = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        if (wtotal) {
            unsigned base = 0;
            if ((tid & 63) == 0) base = atomicAdd(&ctl->nlist, wtotal);
            unsigned pos = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + incl - cnt;
            const unsigned id0 = ((unsigned)un.rq0 << 8) | (unsigned)un.gpl;
            while (mask) {
                const unsigned b = (unsigned)__builtin_ctz(mask);
                mask &= mask - 1u;
                if (pos < (unsigned)g.lcap) v.list[pos] = (uint16_t)(id0 + (b << 8));
                ++pos;
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
// ctl->spd (thread 0 only): cached filter value of the pair (0, n-1); its operands P[1], P[n-1], e[0],
// e[n-1] change only when the last reversal started at position 1 or ended at position n-1.  (In
// LDS, not in a register: the kernel sits at the VGPR limit and loop-carried registers get spilled.)
template <typename T, int Q, bool CL>
__device__ void find_best(const View<T>& v, const Geo& g, int tid, bool need_exact_sign,
                          unsigned last, double& e, unsigned& idx, Prof& pf)
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
        sweep<Q, CL>(v, g, tid, last, a1, a2, chunk, pf);
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
                        ctl->spd = ((d0 + d1) - v.Eq[sn1].x) - v.Eq[s0].x;
                    }
                    const float d = ctl->spd;
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
// Threads per workgroup by filter class.  Class 256 (N about 670..1340) fills a CU's LDS with one
// workgroup of 512 threads = 2 waves per SIMD: bounding the launch at 512 gives the compiler 256 VGPRs
// instead of 128 (no spills); the other classes need the 128-VGPR budget for their occupancy
// (4 workgroups of 256 threads per CU at class 128) or run 1024 threads (class 512, exact kernel).
template <int Q>
constexpr int kMaxThreads = Q == 256 ? 512 : 1024;

template <typename T, int Q, bool CL>
__global__ void __launch_bounds__(kMaxThreads<Q>) k_two_opt(Params p)
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
        if (p.mode == kModeValidate) {          // (uniform branch)
            if (tid == 0) { p.iters[tk] = 0; if (p.status) p.status[tk] = 0; }
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
        for (;;) {
            const bool check_only = forced && it >= target;
            if (check_only && target >= p.max_iter) break;
            find_best<T, Q, CL>(v, g, tid, forced, last, e, idx, pf);
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
        if (CL && st == TSP2OPT_ST_INTERNAL) {
            // cluster broken (a workgroup is missing): this tour and every later tour of the cluster stay
            // untouched and are flagged, by whichever workgroups got here (same values: idempotent)
            if (tid == 0)
                for (int64_t t = tk; t < p.num_tours; t += p.num_clusters) {
                    if (p.status) p.status[t] = st;
                    p.iters[t] = 0;
                }
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

}  // namespace t2o
