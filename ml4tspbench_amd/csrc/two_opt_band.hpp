// ml4tspbench_amd/csrc/two_opt_band.hpp -- band cluster kernel k_two_opt_cluster (several workgroups per tour, N beyond the fast path)
// Part of the single translation unit tsp2opt.hip (design notes at its top and in DESIGN.md).
#pragma once

namespace t2o {

// ------------------------------------------------------------------------------------------
// cluster kernel: several workgroups per tour (few tours, or N beyond one workgroup's LDS)
// ------------------------------------------------------------------------------------------
// The (row segment x diagonal group) rectangle of one tour is tiled over CS = CR x CD workgroups.
// Every workgroup keeps the whole tour (uint16) in LDS and applies every reversal to it, so the
// tour needs no exchange; its filter copy covers only its own tile: compact column arrays (one float
// per window position, 4-way interleaved: position w sits in sub-array w & 3 at index w >> 2, so the
// operands of a lane's two streams -- positions 4 apart -- are two CONSECUTIVE floats, one
// ds_read2_b32) for window groups [ilo, ilo+Qw), and plain row arrays for its row band.
// "Shared" geometry (CGeo::sh, CR = 1): every workgroup walks ALL rows and a band of diagonals, so the
// share of a reversal's touched groups is the same in every workgroup (contiguous row bands are not:
// a band inside the reversed segment recomputes everything, one outside only a stripe).  Rows and
// columns then read the SAME arrays, which hold the whole tour once: positions [0, n + 12) with
// column (and padded row) indices wrapping at n / 4 groups when 4 divides n, else the window
// extended to the last walked column.  Per iteration each
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
    int Qw;                 // entries per sub-array of the compact column arrays
    int sh;                 // rows read from the column arrays (CR = 1: whole tour in every workgroup)
    int il;                 // sh and CD odd: lane groups interleaved over the diagonal bands (CView::gstr)
    int wrapg;              // sh: window groups wrap at this index (n / 4), or a huge value (no wrap)
    int rowsw;              // row positions of the band = 4*NSw*MS
    int xlds;
    int K, Ksh;             // macro steps per cache entry ("step group", power of two) and log2(4*K)
    int NG;                 // step groups per unit = ceil(MS / K), <= 64
    int inc;                // group-minimum cache + incremental sweeps (needs GPw*NSw <= nt)
    int Psh;                // a listed group is recomputed in parts of 2^Psh macro steps (<= K), spread over all lanes
    int lcap;               // capacity of the changed-group list
    float eps2;
};

struct CParams {
    const void* points;
    int64_t* tours;
    int32_t* iters;
    int32_t* work_sweeps;
    int32_t* status;
    unsigned* csync;        // per cluster `cstride` dwords: [1] abort, [16..] two sets of slots (cluster_exchange_w)
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
    float *Xc, *Yc, *Ec;    // compact column arrays, position w (window group ilo + il, w & 3 = u): [u*Qw + il]
    float *RX, *RY, *RE;    // row band, natural order (null when the rows are read from the column arrays)
    unsigned* Mc;           // filter minimum of every step group as an ordered key (fkey), Mc[group * units + unit], or null
    uint16_t* list;         // ids (unit * NG + group) of the step groups the last reversal touched
    const T* X;
    double cx, cy, sc;
    int ilo;                // window group of local column index 0
    int rbase;              // tour position (unwrapped) of local row 0
    int rs0, gp0;           // first row segment / lane group of this workgroup
    int gstr;               // local lane group gpl is lane group gp0 + gstr * gpl of the tour.  Row bands (local column
                            // window): contiguous, gstr = 1.  Shared rows (whole tour in LDS) with an ODD number of
                            // diagonal bands: INTERLEAVED, gp0 = cd, gstr = CD -- a pair on diagonal d reads a moved
                            // position for (L + min(d, L)) / n of its rows, so a contiguous band of short diagonals has
                            // about half the work of a band of long ones and the cluster waits for the slowest;
                            // interleaved, every workgroup sees the same mix.  (Odd: adjacent lanes' column operands are
                            // then 2 * CD = 2 (mod 4) floats apart, the same 2-way LDS bank pattern as contiguous groups;
                            // an even CD puts 4, 8, ... lanes on one bank: CD = 16 is a 32-way conflict)
    int coff;               // column index of this workgroup's first lane group relative to its row (sh: 2*gp0)
    int fo, fs;             // frame of the LDS tour against the reference's array (View::fo / fs, resolve_move)
};

// p in [0, n]: position n is position 0 of the cycle
template <typename T>
__device__ __forceinline__ void scaled_pos(const CView<T>& v, int n, int p, float& x, float& y)
{
    const int node = v.tour[p >= n ? p - n : p];
    x = (float)(((double)v.X[2 * node] - v.cx) * v.sc);
    y = (float)(((double)v.X[2 * node + 1] - v.cy) * v.sc);
}

// is position p (in [0, n)) one of the `count` cyclic positions that start at `start` (in [0, n))?
__device__ __forceinline__ bool in_cyclic(int p, int start, int count, int n)
{
    int d = p - start;
    d += d < 0 ? n : 0;
    return d < count;
}

// (re)build the filter entries whose LDS tour position is one of the `len` cyclic positions from lo0
// (coordinates) / one of the len + 1 positions from lo0 - 1 (edge lengths: the edges in front of, inside and
// behind the reversed range).  lo0 in [0, n); len = n: everything.
template <typename T>
__device__ void stage_c(const CView<T>& v, const CGeo& g, int tid, int lo0, int len)
{
    const int n = g.n;
    const int e0 = lo0 == 0 ? n - 1 : lo0 - 1, elen = len + 1;
    for (int k = tid; k < 4 * g.Qw; k += g.nt) {
        const int il = k >> 2, u = k & 3, sl = u * g.Qw + il;
        const int w = 4 * (v.ilo + il) + u;
        const int pa = w % n;
        if (in_cyclic(pa, lo0, len, n)) { float x, y; scaled_pos(v, n, pa, x, y); v.Xc[sl] = x; v.Yc[sl] = y; }
        if (in_cyclic(pa, e0, elen, n)) {
            float x0, y0, x1, y1;
            scaled_pos(v, n, pa, x0, y0); scaled_pos(v, n, pa + 1, x1, y1);
            v.Ec[sl] = dist32(x0, y0, x1, y1);
        }
    }
    if (g.sh) return;
    for (int k = tid; k < g.rowsw + 8; k += g.nt) {
        const int pp = (v.rbase + k) % n;
        if (in_cyclic(pp, lo0, len, n)) { float x, y; scaled_pos(v, n, pp, x, y); v.RX[k] = x; v.RY[k] = y; }
        if (in_cyclic(pp, e0, elen, n)) {
            float x0, y0, x1, y1;
            scaled_pos(v, n, pp, x0, y0); scaled_pos(v, n, pp + 1, x1, y1);
            v.RE[k] = dist32(x0, y0, x1, y1);
        }
    }
}

__device__ __forceinline__ v2f ld2(const float* p) { return (v2f){p[0], p[1]}; }

// operands of the lane's two streams for column block b: stream A = window positions 4b+2 .. 4b+5 (and
// edges 4b+1 .. 4b+4), stream B the same 4 positions later = the next float of the same sub-array
__device__ __forceinline__ void load_block_c(const float* Xc, const float* Yc, const float* Ec, int Q, int b,
                                             v2f (&CX)[4], v2f (&CY)[4], v2f (&E)[4])
{
    CX[0] = ld2(Xc + 2 * Q + b); CX[1] = ld2(Xc + 3 * Q + b); CX[2] = ld2(Xc + b + 1); CX[3] = ld2(Xc + Q + b + 1);
    CY[0] = ld2(Yc + 2 * Q + b); CY[1] = ld2(Yc + 3 * Q + b); CY[2] = ld2(Yc + b + 1); CY[3] = ld2(Yc + Q + b + 1);
    E[0] = ld2(Ec + Q + b); E[1] = ld2(Ec + 2 * Q + b); E[2] = ld2(Ec + 3 * Q + b); E[3] = ld2(Ec + b + 1);
}

// same macro step as the single-workgroup kernel; rl = local row of r0 (a multiple of 4), colIdx = column
// block of the NEW operands (already wrapped).  Rows come from the band arrays, or (sh) from the column
// arrays at row group rl / 4 (wrapped by the caller's rg).
template <typename T>
__device__ __forceinline__ float macro_step_c(const CView<T>& v, const CGeo& g, int rl, int rg, int colIdx,
                                              const v2f (&CXo)[4], const v2f (&CYo)[4], v2f (&CXn)[4],
                                              v2f (&CYn)[4], const v2f (&Eo)[4], v2f (&En)[4], v2f (&Dc)[4])
{
    const int Q = g.Qw;
    load_block_c(v.Xc, v.Yc, v.Ec, Q, colIdx, CXn, CYn, En);
    float rx[4], ry[4], er[4];
    if (g.sh) {         // (uniform) positions 4rg+1 .. 4rg+4, edges 4rg .. 4rg+3
        rx[0] = v.Xc[Q + rg]; rx[1] = v.Xc[2 * Q + rg]; rx[2] = v.Xc[3 * Q + rg]; rx[3] = v.Xc[rg + 1];
        ry[0] = v.Yc[Q + rg]; ry[1] = v.Yc[2 * Q + rg]; ry[2] = v.Yc[3 * Q + rg]; ry[3] = v.Yc[rg + 1];
        er[0] = v.Ec[rg]; er[1] = v.Ec[Q + rg]; er[2] = v.Ec[2 * Q + rg]; er[3] = v.Ec[3 * Q + rg];
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) { rx[a] = v.RX[rl + 1 + a]; ry[a] = v.RY[rl + 1 + a]; er[a] = v.RE[rl + a]; }
    }
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
// (Dc rebuilt from scratch); f(mm) receives the filter minimum of every macro step in order.
// Shared geometry: row groups and column blocks wrap at g.wrapg (the tour is stored once).
template <typename T, typename F>
__device__ __forceinline__ void walk_c(const CView<T>& v, const CGeo& g, int rl, int col, int count, F&& f)
{
    v2f CX0[4], CY0[4], CX1[4], CY1[4], E0[4], E1[4], Dc[4];
    const int wg = g.wrapg;
    int rg = rl >> 2;
    rg -= rg >= wg ? wg : 0;
    int cb = col - 1;
    cb -= cb >= wg ? wg : 0;
    load_block_c(v.Xc, v.Yc, v.Ec, g.Qw, cb, CX0, CY0, E0);
    const float ax = g.sh ? v.Xc[rg] : v.RX[rl], ay = g.sh ? v.Yc[rg] : v.RY[rl];
#pragma unroll
    for (int q = 0; q < 4; ++q) Dc[q] = dist32x2(ax, ay, CX0[q], CY0[q]);
    int m = 0;
    for (; m + 1 < count; m += 2) {
        cb += 1; cb -= cb >= wg ? wg : 0;
        f(macro_step_c(v, g, rl, rg, cb, CX0, CY0, CX1, CY1, E0, E1, Dc));
        rg += 1; rg -= rg >= wg ? wg : 0;
        cb += 1; cb -= cb >= wg ? wg : 0;
        f(macro_step_c(v, g, rl + 4, rg, cb, CX1, CY1, CX0, CY0, E1, E0, Dc));
        rg += 1; rg -= rg >= wg ? wg : 0;
        rl += 8;
    }
    if (m < count) {
        cb += 1; cb -= cb >= wg ? wg : 0;
        f(macro_step_c(v, g, rl, rg, cb, CX0, CY0, CX1, CY1, E0, E1, Dc));
    }
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
        if (v.gp0 + v.gstr * gpl >= g.GP) continue;              // padding of the last diagonal band
        unsigned id = (unsigned)u * (unsigned)g.NG;
        unsigned* mc = v.Mc ? v.Mc + u : nullptr;
        float gm = INFINITY;
        int k = 0;
        const int K = g.K;
        walk_c(v, g, 4 * sl * g.MS, sl * g.MS + 2 * v.gstr * gpl + 1 + v.coff, g.MS, [&](float mm) {
            gm = __builtin_fminf(gm, mm);
            if (++k == K) {
                if (mc) { *mc = fkey(gm); mc += units; }
                track(gm, id, a1, a2, chunk);
                gm = INFINITY; k = 0; ++id;
            }
        });
        if (k) {                                        // last, shorter group
            if (mc) *mc = fkey(gm);
            track(gm, id, a1, a2, chunk);
        }
    }
}

// filter minimum of one part (2^Psh macro steps) of a step group, from scratch (incremental sweeps);
// +inf for a part beyond the end of the unit
template <typename T>
__device__ __forceinline__ float part_min_c(const CView<T>& v, const CGeo& g, int u, int grp, int part)
{
    const int gpl = u % g.GPw, sl = u / g.GPw;
    const int P = 1 << g.Psh;
    const int m0 = grp * g.K + part * P;
    int cnt = g.MS - m0 < P ? g.MS - m0 : P;
    float gm = INFINITY;
    if (cnt > 0)
        walk_c(v, g, 4 * (sl * g.MS + m0), sl * g.MS + 2 * v.gstr * gpl + 1 + m0 + v.coff, cnt,
               [&](float mm) { gm = __builtin_fminf(gm, mm); });
    return gm;
}

// Filter sweep of one iteration of the cluster kernel.  `last` = the reversal applied since the
// previous sweep or kFullSweep.  With the cache, only step groups that read a changed window
// position (coordinates last.i+1..last.j, edges last.i..last.j, any cyclic copy) are recomputed:
// each lane marks the touched groups of its own unit in closed form (rows, columns), the ids are compacted into a list and processed evenly by all lanes, then
// every lane reduces the cached minima of its unit exactly as the full sweep would.
template <typename T>
__device__ __forceinline__ void sweep_c(const CView<T>& v, const CGeo& g, int tid, unsigned last, int lo0, int len,
                                        float& a1, float& a2, unsigned& chunk, Prof& pf)
{
    Ctl* ctl = v.ctl;
    const int units = g.GPw * g.NSw;
    bool full = !g.inc || last == kFullSweep;
    FINE_START(pf);
    if (!full) {
        // window positions [i, j] (+ k n) read changed values: the reversed LDS range lo0 .. lo0 + len - 1 and the
        // edge in front of it; the earliest copy that can reach the window starts below 0 when the range wraps
        int i = lo0 - 1, j = lo0 + len - 1;
        if (j >= g.n) { i -= g.n; j -= g.n; }
        const int gpl = tid % g.GPw, sl = tid / g.GPw;
        const bool own = tid < units && v.gp0 + v.gstr * gpl < g.GP;
        unsigned long long mask = 0ull;
        if (own) {
            const int rb = v.rbase + 4 * sl * g.MS;                 // window position of the unit's first row
            const int ca = rb + 2 + 8 * (v.gp0 + v.gstr * gpl);              // ... of its first column (stream A)
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
                    v.Mc[q * units + tid] = 0xffffffffu;        // rebuilt below from its parts (atomic min)
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
            // every listed group is recomputed in parts of 2^Psh macro steps, spread evenly over all lanes:
            // coarse groups (K up to 64 macro steps) handed out whole leave most lanes waiting for the few
            // that got one group more
            __builtin_amdgcn_s_setprio(0);
            const int psh = __builtin_ctz((unsigned)g.K) - g.Psh;       // log2(parts per group)
            const unsigned items = nl << psh;
            // handed out 64 at a time from a counter: waves on a
            // slower SIMD take fewer, nobody waits at the barrier for a wave that was dealt one round more
            unsigned* next = &ctl->next_item;                          // starts at nt: the first batch is dealt
            for (unsigned t0 = (unsigned)(tid & ~63); t0 < items;) {
                const unsigned t = t0 + (unsigned)(tid & 63);
                if (t < items) {
                    const int id = v.list[t >> psh];
                    const int u = id / g.NG, q = id - u * g.NG;
                    const float pm = part_min_c(v, g, u, q, (int)(t & ((1u << psh) - 1u)));
                    atomicMin(&v.Mc[q * units + u], fkey(pm));
                }
                if (items <= (unsigned)g.nt) break;
                if ((tid & 63) == 0) t0 = atomicAdd(next, 64u);
                t0 = (unsigned)__builtin_amdgcn_readfirstlane((int)t0);
            }
            __builtin_amdgcn_s_setprio(3);
        }
        FINE(pf, 1);
        __syncthreads();
        if (tid == 0) { ctl->nlist = 0; ctl->next_item = (unsigned)g.nt; }
        if (!full) {
            a1 = INFINITY;
            a2 = INFINITY;
            chunk = 0;
            if (own) {
                unsigned id = (unsigned)tid * (unsigned)g.NG;
                const unsigned* mc = v.Mc + tid;
                for (int q = 0; q < g.NG; ++q) {
                    track(funkey(*mc), id, a1, a2, chunk);
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

// exact delta of the REFERENCE pair (i, j), read through the frame
template <typename T>
__device__ __forceinline__ double exact_delta_c(const CView<T>& v, int n, int i, int j)
{
    View<T> w;
    w.tour = v.tour;
    w.X = v.X;
    w.fo = v.fo;
    w.fs = v.fs;
    return exact_delta(w, n, i, j);
}

// reference index of the pair of LDS edges r, c (pair_index of the one-workgroup kernel)
template <typename T>
__device__ __forceinline__ unsigned pair_index_c(const CView<T>& v, int n, int r, int c)
{
    const int er = v.fs > 0 ? wrapn(r - v.fo, n) : wrapn(v.fo - 1 - r, n);
    const int ec = v.fs > 0 ? wrapn(c - v.fo, n) : wrapn(v.fo - 1 - c, n);
    const int i = er < ec ? er : ec, j = er < ec ? ec : er;
    return ((unsigned)i << 16) | (unsigned)j;
}

// exact value of pair `pr` (0..31) of macro step `m` of local unit `u`; false for padding units
template <typename T>
__device__ __forceinline__ bool eval_chunk_pair_c(const CView<T>& v, const CGeo& g, int u, int m, int pr,
                                                  double& e, unsigned& idx)
{
    const int gpl = u % g.GPw, sl = u / g.GPw;
    const int gp = v.gp0 + v.gstr * gpl;
    if (gp >= g.GP) return false;
    const int gi = 2 * gp + (pr >> 4);
    const int r0 = 4 * ((v.rs0 + sl) * g.MS + m);
    const int c0 = r0 + 2 + 4 * gi;
    const int a = (pr >> 2) & 3, q = pr & 3;
    const int r = (r0 + a) % g.n, c = (c0 + a + q) % g.n;
    idx = pair_index_c(v, g.n, r, c);
    e = exact_delta_c(v, g.n, (int)(idx >> 16), (int)(idx & 0xffffu));
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
            walk_c(v, g, 4 * (sl * g.MS + m), sl * g.MS + 2 * v.gstr * gpl + 1 + m + v.coff, 1, [&](float x) { mm = x; });
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
                               int lo0, int len, double& e, unsigned& idx, Prof& pf)
{
    Ctl* ctl = v.ctl;
    float a1, a2;
    unsigned chunk;
    unsigned long long t0 = 0;
    if (pf.on) t0 = clock64();
    sweep_c(v, g, tid, last, lo0, len, a1, a2, chunk, pf);
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
            const double d = exact_delta_c(v, g.n, 0, g.n - 1);
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
                if (v.Mc[t] <= fkey(thr) && v.gp0 + v.gstr * (u % g.GPw) < g.GP) {
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
            const double d = exact_delta_c(v, g.n, 0, g.n - 1);
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
    v.Xc = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * 4 * (size_t)g.Qw);
    v.Yc = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * 4 * (size_t)g.Qw);
    v.Ec = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * 4 * (size_t)g.Qw);
    v.RX = v.RY = v.RE = nullptr;
    if (!g.sh) {
        v.RX = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * (size_t)(g.rowsw + 8));
        v.RY = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * (size_t)(g.rowsw + 8));
        v.RE = reinterpret_cast<float*>(smem + off); off += align16(sizeof(float) * (size_t)(g.rowsw + 8));
    }
    v.Mc = nullptr;
    v.list = nullptr;
    if (g.inc) {
        const size_t groups = (size_t)g.GPw * g.NSw * g.NG;
        v.Mc = reinterpret_cast<unsigned*>(smem + off); off += align16(sizeof(unsigned) * groups);
        v.list = reinterpret_cast<uint16_t*>(smem + off); off += align16(sizeof(uint16_t) * (size_t)g.lcap);
    }
    T* xl = reinterpret_cast<T*>(smem + off);
    v.rs0 = cr * g.NSw;
    v.gp0 = g.il ? cd : cd * g.GPw;
    v.gstr = g.il ? g.CD : 1;
    v.ilo = g.sh ? 0 : v.rs0 * g.MS + 2 * v.gp0;
    v.coff = g.sh ? 2 * v.gp0 : 0;
    v.rbase = 4 * v.rs0 * g.MS;
    Ctl* ctl = v.ctl;
    unsigned* cs = p.csync + (size_t)cluster * (size_t)p.cstride;
    unsigned epoch = 0;                         // barriers passed by this cluster so far

    for (int64_t tk = cluster; tk < p.num_tours; tk += p.num_clusters) {
        __syncthreads();
        if (tid == 0) { ctl->nlist = 0; ctl->next_item = (unsigned)nt; }
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
        v.fo = 0;                               // the staged copy IS the reference's array
        v.fs = 1;
        stage_c(v, g, tid, 0, n);
        __syncthreads();

        // ---- improvement loop (independent termination) ------------------------------------------
        const long long t_start = p.time_limit_ticks ? wall_clock64() : 0;
        int it = 0, sweeps = 0, st = 0;
        bool aborted = false;
        Prof pf = make_prof(p.stamps);
        unsigned long long t_x = 0, t_tour = pf.on ? clock64() : 0;
        unsigned last = kFullSweep;             // reversal applied since the previous sweep
        int lo0 = 0, len = 0;                   // ... as the range of LDS positions it moved
        for (;;) {
            double e;
            unsigned idx, fl = 0;
            find_best_tile(v, g, tid, wc == 0, last, lo0, len, e, idx, pf);
            ++sweeps;
            unsigned long long tb = pf.on ? clock64() : 0;
            unsigned myflags = 0;
            if (wc == 0 && p.time_limit_ticks && wall_clock64() - t_start > p.time_limit_ticks)
                myflags |= kFlagStop;
            FINE_START(pf);
            if (!cluster_exchange_w(ctl, cs, g.CS, wc, epoch, tid, myflags, e, idx, fl)) { aborted = true; break; }
            ++epoch;
            FINE(pf, 11);                                                    // cluster exchange
            if (pf.on) { const unsigned long long t1 = clock64(); t_x += t1 - tb; tb = t1; }
            if (!(e < -1.0e-6)) break;                                       // two_opt.py:40
            // two_opt.py:42, on the shorter side of the cycle (resolve_move: the frame records the orientation)
            {
                View<T> w;
                w.fo = v.fo;
                w.fs = v.fs;
                resolve_move<1>(w, n, idx, lo0, len);
                v.fo = w.fo;
                v.fs = w.fs;
            }
            const int h = len >> 1;
            for (int k = tid; k < h; k += nt) {
                int pa = lo0 + k, pb = lo0 + len - 1 - k;
                pa -= pa >= n ? n : 0;
                pb -= pb >= n ? n : 0;
                const uint16_t a = v.tour[pa], b = v.tour[pb];
                v.tour[pa] = b;
                v.tour[pb] = a;
            }
            __syncthreads();
            FINE(pf, 8);                                                     // tour reversal
            stage_c(v, g, tid, lo0, len);
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
#ifdef TSP2OPT_FINE
        // the same listed-groups phase (slot 1) and its closing barrier (slot 2) seen by the middle and the last wave
        if (pf.on && tid == (nt >> 1)) { atomicAdd(&p.stamps[8 + 12], pf.fine[1]); atomicAdd(&p.stamps[8 + 14], pf.fine[2]); }
        if (pf.on && tid == nt - 64) { atomicAdd(&p.stamps[8 + 13], pf.fine[1]); atomicAdd(&p.stamps[8 + 15], pf.fine[2]); }
#endif
        if (aborted) {
            // cluster broken (a workgroup is missing): this tour and every later tour of the cluster stay
            // untouched and are flagged, by whichever workgroups got here (same values: idempotent)
            if (tid == 0)
                for (int64_t t = tk; t < p.num_tours; t += p.num_clusters) {
                    if (p.status) p.status[t] = TSP2OPT_ST_INTERNAL;
                    p.iters[t] = 0;
                }
            return;
        }
        if (wc == 0) {
            for (int k = tid; k <= n; k += nt) gt[k] = (int64_t)v.tour[wrapn(v.fo + v.fs * k, n)];   // R[k]
            if (tid == 0) {
                p.iters[tk] = it;
                if (p.work_sweeps) p.work_sweeps[tk] += sweeps;
                if (p.status) p.status[tk] = st;
            }
        }
    }
}

}  // namespace t2o
