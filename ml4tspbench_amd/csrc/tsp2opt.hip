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
// Hot loop ("filter sweep", two_opt_kernel.hpp): the pair triangle is re-indexed as CYCLIC DIAGONALS
// (r, (r+d) mod N), d = 2..N/2, which is a rectangle (no triangle waste) and along which
// D(r+1,c+1) of one pair is D(r,c) of the next, so each pair costs ONE distance.  A lane owns 8
// adjacent diagonals as two "streams" A and B of 4 and walks rows in macro steps of 4 (32 pairs,
// 32 sqrt); two macro steps make a TILE of 8 rows x 8 diagonals.  Every value of a step is an
// (A, B) register pair, so the subtractions, squares and sums issue as v_pk_add/v_pk_mul/v_pk_fma_f32
// (measured on MI355X: 25 SIMD-cycles per 64 pairs against 37 for the scalar form,
// tools/valu_chain_microbench.hip).  Tour-ordered, centred and scaled fp32 coordinates sit in LDS as
// three 8-way interleaved arrays (x, y, edge length) whose entries are already the (A, B) pairs
// (value at window position w and at w + 4), so one conflict-free ds_read_b64 per operand feeds the
// packed math with no register moves; row operands are broadcast reads of the same arrays.
//
// The fp32 value is only a FILTER with a proven error bound eps: every lane keeps its best tile a1
// and runner-up a2; after a workgroup min m only tiles with a1 <= m + 2 eps can hold the true argmin.
// Those are re-evaluated pair by pair, first in fp32 (a unique in-window pair far enough below
// -1e-6 IS the argmin: no exact arithmetic), else EXACTLY (reference dtype, reference operation
// order).  The result is bit-identical to evaluating every pair exactly, at fp32 VALU cost.
//
// INCREMENTAL SWEEPS: the filter minimum of every tile is cached in LDS; a reversal of
// tour[i+1..j] only changes tiles that read a position in [i, j], which every lane marks in closed
// form; the marked tiles are compacted into a list and recomputed evenly by all lanes (about 42 % of
// the pairs per iteration at TSP-500).  The cache also names the candidates of iterations where the
// per-lane best/runner-up is not enough (two candidates in one lane).
//
// Few tours: a tour is shared by several workgroups -- the full-window cluster (k_two_opt<.., CL>,
// lane groups interleaved over the workgroups, N within the fast path) or the band cluster
// (two_opt_band.hpp, any N) -- which exchange their exact local bests once per iteration.
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
#include <mutex>
#include <string>

#include "tsp2opt.h"
#include "host_util.hpp"
#include "wave_reduce.hpp"

#pragma clang fp contract(off)

#include "two_opt_common.hpp"
#include "two_opt_kernel.hpp"
#include "two_opt_band.hpp"
#include "two_opt_aux.hpp"

namespace t2o {

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// optional timing of the persistent kernel with HIP events on the caller's stream
// (tsp2opt_profile_enable / tsp2opt_profile_last_ms / tsp2opt_profile_history_ms; used by bench.py
// for the roofline figure).  A ring of event sets per device: the last kProfRing calls stay readable,
// so a timed loop needs no synchronisation between its steps.
constexpr int kMaxDev = 16;
constexpr int kProfRing = 64;
struct ProfEntry {
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // main begin/end, forced begin/end
    bool have_main = false, have_forced = false;
};
struct ProfSlot {
    ProfEntry ring[kProfRing];
    long long calls = 0;                                        // profiled calls so far; current = (calls - 1) % ring
    ProfEntry& cur() { return ring[(calls > 0 ? calls - 1 : 0) % kProfRing]; }
};
// Test / diagnostic knobs (tsp2opt_debug_*, tsp2opt_profile_enable).  They belong to the CALLING THREAD:
// a setter changes the calling thread's copy, and every entry point snapshots its own thread's copy once
// on entry, so a knob flipped by one thread never changes a call running on another (one thread per
// device is the supported concurrency).
struct Opts {
    int profile = 0;            // HIP events around the persistent kernel (tsp2opt_profile_enable)
    int dbg_sweeps = 0;         // filter sweeps only (tsp2opt_debug_sweep_only)
    int stamps = 0;             // phase timers into workspace bytes [64, 256)
    int force_nt = 0;           // threads per workgroup of the single-workgroup plan
    int wide_few = 1;           // twice the threads per tour when the batch has at most one tour per CU
    int cl_u0 = 64;             // full-window cluster: lanes without a unit (the oldest wave applies the reversal while
                                // the others mark: TSP-1000 x 128 13.88 -> 13.67 ms; 128: 13.96); tuning knob 13
    int extra_waves = 4;        // ... plus this many waves without a unit (they apply the reversal while the others mark)
    int lcap16 = 13;            // changed-group list capacity of the band cluster kernel, in 1/16 of the groups
    int lcap16s = 13;           // changed-tile list capacity of the one-workgroup / full-window kernel, in 1/16 of the tiles
    int lds_pad = 0;            // unused LDS bytes added to the one-workgroup kernel's request (occupancy sweeps)
    int inc_min_tiles = 3;      // tile cache only when a lane owns at least this many tiles (measured: with one or
                                // two tiles per lane, N <~ 110, plain full sweeps are faster than keeping the cache)
    int cluster_inc = 1;        // 0: band cluster kernel without the group-minimum cache
    int cluster_mode = -1;      // -1 auto, 0 never, >= 2 force that many workgroups per tour (tests)
    int cluster_variant = 0;    // 0 auto, 1 band kernel only (k_two_opt_cluster), 2 full-window kernel only
    int force_lockstep = 0;     // coupled mode: replay the whole call in k_group_lockstep (tests)
    int part_sh = 4;            // band cluster: listed groups are recomputed in parts of at most 2^part_sh macro steps
    int half_units = 1;         // macro-step units for small tours without the tile cache (Geo::half)
};
thread_local Opts t_opts;

ProfSlot g_prof[kMaxDev];
std::mutex g_prof_mu[kMaxDev];      // the rings are per device, but any thread may read them

// Cluster kernels (several workgroups per tour meeting at a spin barrier) need all their workgroups
// resident at once.  The grid never exceeds what the device holds (checked at launch), and cluster
// launches of this process on one device are chained through an event, so two of them issued on
// different streams never share the CUs (each could otherwise hold slots the other waits for).
struct ClusterChain {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    bool recorded = false;
};
ClusterChain g_chain[kMaxDev];

// Scratch of the allocation-free entry point (tsp2opt_solve_noalloc): the workspace of tsp2opt_solve
// (control words, per-group targets, per-tour counters, cluster barrier words and slots) in a
// file-scope device array.  `scratch` is only a host-side stand-in for the symbol: the device
// address comes from hipGetSymbolAddress, once per device (every GPU holds its own copy), and that
// pointer is what kernels, memsets and copies get.
constexpr size_t kScratchBytes = 2u << 20;
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kScratchBytes];
struct ScratchDev {
    std::mutex mu;
    void* ptr = nullptr;
};
ScratchDev g_scratch[kMaxDev];

// once per device, with that device current: resolve the device address of `scratch` and create the
// event cluster launches are chained through (so that the entry point itself never creates anything)
int scratch_setup(int device, void** out)
{
    if (device < 0 || device >= kMaxDev) return fail(TSP2OPT_E_ARG, "device ordinal beyond the supported 16");
    ScratchDev& sd = g_scratch[device];
    std::lock_guard<std::mutex> lk(sd.mu);
    if (!sd.ptr) {
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr || (reinterpret_cast<uintptr_t>(ptr) % 256))
            return fail(TSP2OPT_E_HIP, "hipGetSymbolAddress returned an unusable address");
        {
            std::lock_guard<std::mutex> lc(g_chain[device].mu);
            if (!g_chain[device].ev)
                HIP_TRY(hipEventCreateWithFlags(&g_chain[device].ev, hipEventDisableTiming));
        }
        sd.ptr = ptr;
    }
    if (out) *out = sd.ptr;
    return 0;
}

// A stream under graph capture is left out of the chain: an event recorded inside a capture becomes a node
// of that graph, and the next uncaptured launch that waited on it would fail.  The launches of one captured
// graph are ordered by the graph itself; replays of such a graph that run at the same time as other cluster
// launches on the same GPU are the caller's to order (include/tsp2opt.h, tsp2opt_solve_noalloc).
bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// chain_begin / chain_end: the CALLER holds g_chain[device].mu from before chain_begin until after chain_end
// (launch_fw, launch_cluster), so wait + launch + record of one call are atomic with respect to other threads.
int chain_begin(int device, hipStream_t stream)
{
    if (device < 0 || device >= kMaxDev) return 0;
    ClusterChain& c = g_chain[device];
    if (stream_is_capturing(stream)) return 0;
    if (!c.ev) HIP_TRY(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
    if (c.recorded) HIP_TRY(hipStreamWaitEvent(stream, c.ev, 0));
    return 0;
}

int chain_end(int device, hipStream_t stream)
{
    if (device < 0 || device >= kMaxDev) return 0;
    ClusterChain& c = g_chain[device];
    if (stream_is_capturing(stream)) return 0;
    HIP_TRY(hipEventRecord(c.ev, stream));
    c.recorded = true;
    return 0;
}

int prof_record(const Opts& o, int device, int which, hipStream_t stream)
{
    if (!o.profile || device < 0 || device >= kMaxDev) return 0;
    std::lock_guard<std::mutex> lk(g_prof_mu[device]);
    if (which == 0) {                                           // a new call: next ring entry
        ++g_prof[device].calls;
        g_prof[device].cur().have_main = g_prof[device].cur().have_forced = false;
    }
    ProfEntry& s = g_prof[device].cur();
    if (!s.ev[which]) HIP_TRY(hipEventCreate(&s.ev[which]));
    HIP_TRY(hipEventRecord(s.ev[which], stream));
    if (which == 1) s.have_main = true;
    if (which == 3) s.have_forced = true;
    return 0;
}

struct Plan {
    Geo g;
    size_t lds;
    int S, W;
};

constexpr size_t kLdsMax = 160 * 1024;
constexpr size_t kCsyncBytes = 256 * 1024;  // cluster barrier words + slots (<= 1024 clusters of <= 6 workgroups)
const int kQClasses[] = {16, 32, 64, 128, 256, 512};


Plan make_plan_nt(const Opts& o, int n, int dtype, int cs, int nt_cap);

// cs > 1: geometry of one workgroup of a full-window cluster of cs workgroups (lane groups interleaved)
Plan make_plan(const Opts& o, int n, int dtype, int cs = 1)
{
    Plan pl = make_plan_nt(o, n, dtype, cs, 1024);
    // filter class 256 is compiled for at most 512 threads (kMaxThreads): the few N just above the
    // 512 -> 1024 thread switch that still land in class 256 are planned with 512 threads, or moved to
    // class 512 if that geometry no longer fits class 256
    if (pl.g.qclass == 256 && pl.g.nt > 512) {
        const Plan p2 = make_plan_nt(o, n, dtype, cs, 512);
        pl = p2.g.qclass == 256 ? p2 : make_plan_nt(o, n, dtype, cs, -512);
    }
    return pl;
}

// nt_cap > 0: at most that many threads; nt_cap = -512: default threads, filter class at least 512
Plan make_plan_nt(const Opts& o, int n, int dtype, int cs, int nt_cap)
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
    if (o.force_nt) nt = o.force_nt;
    if (nt_cap > 0 && nt > nt_cap) nt = nt_cap;
    g.nt = nt;
    g.qclass = 0;
    size_t cache = 0;
    {
        const int dn = n / 2 - 1;                       // cyclic diagonals 2..n/2
        const int GP = (dn + 7) / 8;                    // lane groups of 8 adjacent diagonals
        const int G = 2 * GP;
        const int GPl = (GP + cs - 1) / cs;             // lane groups per workgroup
        g.GPl = GPl;
        g.inv_gpl = 1.0f / (float)(GPl > 0 ? GPl : 1);
        // every walked diagonal d = 2..8GP+1 must be a valid pair distance: d <= n-2
        if (GP >= 1 && 8 * GP + 1 <= n - 2) {
            const int R = (n + 3) / 4;                  // macro-step rows needed
            const int u0 = cs > 1 ? std::min(o.cl_u0, nt / 2) : 0;
            g.u0 = u0;
            int NS = (nt - u0) / GPl;
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
                if (nt_cap < 0 && qc < -nt_cap) continue;
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
                const size_t lcap = tiles * (size_t)o.lcap16s / 16 + 1;
                const size_t need = align16(4 * tiles) + align16(2 * lcap) + align16(4 * (size_t)GPl * NS);
                if (MS / 2 >= o.inc_min_tiles && (size_t)NS * (MS / 2) <= 255 && GPl <= 255 && MS / 2 <= 32 && GPl * NS <= nt - u0 &&
                    base + (size_t)g.qclass * 192 + need <= kLdsMax) {
                    g.inc = 1;
                    g.lcap = (int)lcap;
                    cache = need;
                }
            }
        }
    }
    // Small tours without the cache (one or two tiles per lane): units of whole tiles leave lanes idle and round the
    // work per lane up to the next even number of macro steps (TSP-100: 49 of 64 lanes with 4 steps each for 175
    // steps of work, TSP-50: 21 lanes with 2).  Macro-step units (Geo::half) start at any macro row: 63 lanes with
    // 3 steps, 39 lanes with 1.
    if (g.qclass && !g.inc && cs == 1 && o.half_units) {
        const int R = (n + 3) / 4;
        int NS = nt / g.GPl;
        if (NS < 1) NS = 1;
        if (NS > R) NS = R;
        const int MS = (R + NS - 1) / NS;
        NS = (R + MS - 1) / MS;
        const int W = (4 * NS * MS + 8 * g.GP + 8 + 7) & ~7;
        if (MS < g.MS && MS <= 16 && R <= 255 && W / 8 + 1 <= g.qclass && W < 3 * n) {
            g.half = 1; g.NS = NS; g.MS = MS; g.W = W;
            pl.S = 4 * MS; pl.W = W;
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
int launch_main(const Opts& o, const Params& p, const Plan& plan, int device, hipStream_t stream, int* blocks_per_cu)
{
    auto kern = k_two_opt<T, Q, false>;
    if (plan.g.nt > kMaxThreads<Q>) return fail(TSP2OPT_E_ARG, "internal: too many threads for this filter class");
    Plan pl = plan;
    if (o.lds_pad > 0 && pl.lds + (size_t)o.lds_pad <= kLdsMax) pl.lds += (size_t)o.lds_pad;    // diagnostic
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
int launch_fw(Params p, const Plan& pl, int cus, int device, hipStream_t stream)
{
    auto kern = k_two_opt<T, Q, true>;
    if (pl.g.nt > kMaxThreads<Q>) return fail(TSP2OPT_E_ARG, "internal: too many threads for this filter class");
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, pl.g.nt, pl.lds));
    if (per_cu < 1) return fail(TSP2OPT_E_UNSUPPORTED, "internal: cluster workgroup does not fit a CU");
    const int CS = pl.g.CS;
    // every workgroup of the grid must be resident at once: at most per_cu workgroups per CU
    int clusters = (int)std::min<int64_t>((int64_t)cus * per_cu / CS, 1024);
    if (clusters > p.num_tours) clusters = (int)p.num_tours;
    if (clusters < 1) return fail(TSP2OPT_E_UNSUPPORTED, "cluster does not fit the device");
    p.num_clusters = clusters;
    p.cstride = ((16 + 2 * CS * kSlotDw) + 63) & ~63;
    if ((size_t)clusters * (size_t)p.cstride * 4 > kCsyncBytes)
        return fail(TSP2OPT_E_UNSUPPORTED, "internal: cluster sync area too small");
    std::lock_guard<std::mutex> lk(g_chain[device < kMaxDev ? device : 0].mu);
    int rc = chain_begin(device, stream);       // after the previous cluster launch on this device
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(clusters * CS)), dim3((unsigned)pl.g.nt), pl.lds, stream, p);
    HIP_TRY(hipGetLastError());
    return chain_end(device, stream);
}

template <typename T>
int dispatch_fw(const Params& p, const Plan& pl, int cus, int device, hipStream_t stream)
{
    switch (pl.g.qclass) {
    case 128: return launch_fw<T, 128>(p, pl, cus, device, stream);
    case 256: return launch_fw<T, 256>(p, pl, cus, device, stream);
    case 512: return launch_fw<T, 512>(p, pl, cus, device, stream);
    }
    return fail(TSP2OPT_E_ARG, "internal: no full-window cluster kernel for this filter class");
}

template <typename T>
int dispatch_main(const Opts& o, const Params& p, const Plan& pl, int device, hipStream_t stream, int* bpc = nullptr)
{
    switch (pl.g.qclass) {
    case 0: return launch_main<T, 0>(o, p, pl, device, stream, bpc);
    case 16: return launch_main<T, 16>(o, p, pl, device, stream, bpc);
    case 32: return launch_main<T, 32>(o, p, pl, device, stream, bpc);
    case 64: return launch_main<T, 64>(o, p, pl, device, stream, bpc);
    case 128: return launch_main<T, 128>(o, p, pl, device, stream, bpc);
    case 256: return launch_main<T, 256>(o, p, pl, device, stream, bpc);
    case 512: return launch_main<T, 512>(o, p, pl, device, stream, bpc);
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

CPlan make_cluster_plan(const Opts& o, int n, int dtype, int cs_cap, bool smallest)
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
            for (int nt : {768, 512, 256}) {            // <= 768 threads: 3 waves/SIMD, no spills; ties go to the larger
                if (n < 700 && nt != 256) continue;
              // (dns: up to three row segments fewer than the threads allow -- listed cache groups are recomputed in
              //  parts dealt to ALL lanes, so lanes without a unit cost little, while a last group that is mostly
              //  padding (TSP-10000: 139 macro steps in groups of 32, 11 in the fifth) fills the waves with empty parts:
              //  16 segments of 157 steps instead of 18 of 139 are 13 % fewer wave instructions per listed sweep)
              for (int dns = 0; dns < 4; ++dns) {
                int NSw = nt / GPw - dns;
                if (NSw < 1) { if (dns) break; NSw = 1; }
                if (NSw > rmw) NSw = rmw;
                const int MS = (rmw + NSw - 1) / NSw;
                NSw = (rmw + MS - 1) / MS;              // no more row segments than MS steps each need: the walked
                                                        // rows overshoot the band by less than one segment
                if (MS < 4 && cs > 2) continue;         // too little work per cluster barrier
                const int units = GPw * NSw;
                const int passes = (units + nt - 1) / nt;
                const int wps = (nt / 64 + 3) / 4;      // waves per SIMD
                const int rowsw = 4 * NSw * MS;
                // cr == 1: every workgroup walks all rows -> rows and columns share one copy of the tour
                // (positions [0, n + 12), indices wrapping at n / 4 groups) when 4 divides n, else the window
                // is extended to the last walked column; cr > 1: column window of the tile + row band arrays
                const bool shared = cr == 1;
                // (the wrap is a single conditional subtraction: the last walked column block, row overshoot and
                //  both streams included, must stay below twice the wrap point, else the extended window is used)
                const bool wrap = shared && n % 4 == 0 && NSw * MS + 2 * GP + 8 <= 2 * (n / 4);
                int Qw;
                if (wrap) Qw = n / 4 + 4;
                else if (shared) Qw = NSw * MS + 2 * GP + 4;
                else Qw = NSw * MS + 2 * GPw + 4;
                size_t lds = base + 3 * align16((size_t)Qw * 16) +
                             (shared ? 0 : 3 * align16(sizeof(float) * (size_t)(rowsw + 8)));
                if (lds > kLdsMax) continue;
                // group-minimum cache: the finest granularity K (macro steps per entry) that fits
                int K = 0, Ksh = 0, NG = MS, lcap = 0;
                size_t cache = 0;
                if (units <= nt && o.cluster_inc) {
                    for (int k = 1, sh = 2; k <= 64; k *= 2, ++sh) {
                        const int ng = (MS + k - 1) / k;
                        const size_t groups = (size_t)units * ng;
                        const size_t cap = groups * (size_t)o.lcap16 / 16 + 1;
                        const size_t need = align16(4 * groups) + align16(2 * cap);
                        if (ng <= 64 && (size_t)nt * ng <= 65535 && lds + need <= kLdsMax) {
                            K = k; Ksh = sh; NG = ng; lcap = (int)cap; cache = need;
                            break;
                        }
                    }
                }
                // measured: incremental sweeps recompute a little under half of the pairs -- of the SLOWEST
                // workgroup: with row bands (cr > 1) one band usually lies inside the reversed segment and
                // recomputes nearly everything while the others wait for it
                // shared rows with an odd number of diagonal bands interleave the lane groups over the workgroups
                // (CView::gstr): equal work everywhere; contiguous diagonal bands wait for the band of the longest
                // diagonals (measured at TSP-10000, 16 bands: a fifth of the iteration)
                const bool il = shared && (cd & 1) && cd > 1;
                // macro steps per lane and sweep: a full sweep walks the units; listed groups (padding included) are
                // shared by all lanes
                const double steps = K ? (double)units * NG * K / nt : (double)MS * passes;
                const double t_iter = (K ? (cr > 1 ? 0.8 : (il || cd == 1 ? 0.55 : 0.68)) : 1.0) * 0.4 * steps * wps + 12.0 + 0.15 * cs;
                const double cost = smallest ? t_iter * cs : t_iter;    // total WG-time vs latency
                if (cost >= best_cost) continue;
                lds += cache;
                CGeo g{};
                g.n = n; g.nt = nt; g.GP = GP; g.CS = cs; g.CR = cr; g.CD = cd;
                g.GPw = GPw; g.NSw = NSw; g.MS = MS; g.Qw = Qw; g.rowsw = rowsw;
                g.sh = shared ? 1 : 0; g.wrapg = wrap ? n / 4 : 0x3fffffff;
                g.il = il ? 1 : 0;
                g.K = K ? K : 1; g.Ksh = K ? Ksh : 2; g.NG = NG; g.inc = K ? 1 : 0; g.lcap = lcap;
                {   // parts of at most 2^part_sh macro steps (the restart of a part costs about one macro step)
                    int psh = 0;
                    while ((2 << psh) <= g.K && psh < o.part_sh) ++psh;
                    g.Psh = psh;
                }
                g.eps2 = dtype == TSP2OPT_DTYPE_F64 ? 2.0f * 0x1p-16f : 2.0f * 0x1p-15f;
                g.xlds = 0;
                if (lds + align16(xbytes) <= kLdsMax) { g.xlds = 1; lds += align16(xbytes); }
                best.g = g; best.lds = lds; best.ok = true;
                best_cost = cost;
              }
            }
        }
    }
    return best;
}

template <typename T, int MAXNT>
int launch_cluster(CParams p, const CPlan& cp, int cus, int device, hipStream_t stream)
{
    auto kern = k_two_opt_cluster<T, MAXNT>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cp.lds));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, cp.g.nt, cp.lds));
    if (per_cu < 1) return fail(TSP2OPT_E_UNSUPPORTED, "internal: cluster workgroup does not fit a CU");
    int clusters = cus / cp.g.CS;                       // one workgroup per CU: all co-resident
    if (clusters > p.num_tours) clusters = (int)p.num_tours;
    if (clusters < 1) return fail(TSP2OPT_E_UNSUPPORTED, "cluster does not fit the device");
    p.num_clusters = clusters;
    p.cstride = ((16 + 2 * cp.g.CS * kSlotDw) + 63) & ~63;
    if ((size_t)clusters * (size_t)p.cstride * 4 > kCsyncBytes)
        return fail(TSP2OPT_E_UNSUPPORTED, "internal: cluster sync area too small");
    std::lock_guard<std::mutex> lk(g_chain[device < kMaxDev ? device : 0].mu);
    int rc = chain_begin(device, stream);       // after the previous cluster launch on this device
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(clusters * cp.g.CS)), dim3((unsigned)cp.g.nt), cp.lds,
                       stream, p);
    HIP_TRY(hipGetLastError());
    return chain_end(device, stream);
}



// Several workgroups per tour?  kind 0: no (one workgroup per tour), 1: band cluster
// (k_two_opt_cluster, any N), 2: full-window cluster (k_two_opt<.., true>, N within the fast path).
struct ClusterChoice {
    int kind;
    CPlan band;
    Plan fw;
};

ClusterChoice choose_cluster(const Opts& o, int n, int dtype, int64_t num_tours, int cus, const Plan& pl)
{
    ClusterChoice c{};
    c.kind = 0;
    int cap = (int)std::min<int64_t>(64, cus / std::max<int64_t>(1, num_tours));
    const bool forced = o.cluster_mode >= 2;
    if (forced) cap = std::min(o.cluster_mode, cus);
    // full-window cluster: needs the fast path (and one of the instantiated filter classes) with the
    // whole window in LDS; at least 4 lane groups per workgroup
    if (o.cluster_variant != 1 && cap >= 2 && pl.g.qclass != 0 && (forced || n >= 700)) {
        const int cs = std::min(std::min(cap, 64), std::max(1, pl.g.GP / 4));    // <= 64: cluster_exchange_w
        if (cs >= 2) {
            const Plan fw = make_plan(o, n, dtype, cs);
            if (fw.g.qclass >= 128) { c.kind = 2; c.fw = fw; return c; }
        }
    }
    if (o.cluster_variant == 2) return c;
    bool smallest = false;
    if (!forced) {
        if (pl.g.qclass == 0 && cap < 2) { cap = 64; smallest = true; }        // tiling for LDS only
        else if (pl.g.qclass != 0 && (n < 700 || cap < 3)) cap = 0;
        // (small tours are not worth a barrier; with two band workgroups per tour the one-workgroup
        //  kernel with its finer incremental sweeps is faster even on half the CUs)
    }
    c.band = make_cluster_plan(o, n, dtype, cap, smallest);
    if (c.band.ok) c.kind = 1;
    return c;
}

template <typename T>
int solve_t(const Opts& o, const void* points, int64_t* tours, int32_t* iters_out, int32_t* sweeps_out,
            double* lengths_out, int32_t* status_out, int64_t num_tours, int n, int tpi,
            int group_size, int max_iter, double time_limit_s, int dtype, int device, void* ws,
            hipStream_t stream)
{
    Plan pl = make_plan(o, n, dtype);
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    // A batch that leaves every workgroup a CU to itself: twice the threads per tour (twice the row segments, half
    // the tiles per unit; four times for N = 130..160).  The listed tiles of an iteration then take half as many
    // passes, and a lone workgroup's iteration is a latency chain, not an issue-rate problem; with two or more
    // workgroups per CU the narrower geometry wins (measured: N = 100..640 at 128..256 tours 7..28 % faster, at 384
    // tours even or slower, N = 50 slower).
    const int widen = pl.g.nt == 64 ? (n >= 130 ? 4 : n >= 96 ? 2 : 1) : (pl.g.nt == 128 || pl.g.nt == 256) ? 2 : 1;
    if (o.wide_few && !o.force_nt && pl.g.qclass && num_tours <= cus && widen > 1) {
        Opts o2 = o;
        o2.force_nt = widen * pl.g.nt;
        const Plan p2 = make_plan(o2, n, dtype);
        if (p2.g.qclass && p2.g.nt == widen * pl.g.nt) {
            pl = p2;
            if (pl.g.inc) {     // ... and waves without a unit: they apply the reversal while the others mark
                const int cap = pl.g.qclass == 256 ? 512 : 1024;    // kMaxThreads of the filter class
                pl.g.u0 = std::min(cap - pl.g.nt, 64 * o.extra_waves);
                pl.g.nt += pl.g.u0;
            }
        }
    }
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
    p.stamps = o.stamps ? reinterpret_cast<unsigned long long*>(w8 + 64) : nullptr;
    p.dbg_sweeps = o.dbg_sweeps;
    p.g = pl.g;
    // test knob: coupled mode replayed entirely by k_group_lockstep (phase 1 only validates the tours)
    const bool lockstep_only = group_size > 1 && o.force_lockstep != 0;
    if (lockstep_only) p.mode = kModeValidate;
    if (time_limit_s > 0.0) {
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
    int rc = prof_record(o, device, 0, stream);
    if (rc) return rc;
    // few tours (or N beyond one workgroup's LDS): several workgroups per tour
    ClusterChoice cc{};
    cc.kind = 0;
    if (group_size == 1 && o.cluster_mode != 0 && o.dbg_sweeps == 0) {
        cc = choose_cluster(o, n, dtype, num_tours, cus, pl);
        if (cc.kind != 0) HIP_TRY(hipMemsetAsync(w8 + wl.off_csync, 0, kCsyncBytes, stream));
        if (cc.kind == 1) {
            const CPlan& cp = cc.band;
            CParams c{};
            c.points = points; c.tours = tours; c.iters = p.iters; c.work_sweeps = p.work_sweeps;
            c.status = p.status; c.csync = reinterpret_cast<unsigned*>(w8 + wl.off_csync);
            c.num_tours = num_tours; c.tpi = tpi; c.max_iter = max_iter;
            c.time_limit_ticks = p.time_limit_ticks; c.g = cp.g;
            c.stamps = p.stamps;
            rc = cp.g.nt <= 512 ? launch_cluster<T, 512>(c, cp, cus, device, stream)
                                : launch_cluster<T, 768>(c, cp, cus, device, stream);
            if (rc) return rc;
        } else if (cc.kind == 2) {
            Params pc = p;
            pc.g = cc.fw.g;
            pc.csync = reinterpret_cast<unsigned*>(w8 + wl.off_csync);
            rc = dispatch_fw<T>(pc, cc.fw, cus, device, stream);
            if (rc) return rc;
        }
    }
    if (cc.kind == 0) {
        rc = dispatch_main<T>(o, p, pl, device, stream);
        if (rc) return rc;
    }
    rc = prof_record(o, device, 1, stream);
    if (rc) return rc;
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
        if (!lockstep_only) {
            rc = prof_record(o, device, 2, stream);
            if (rc) return rc;
            rc = dispatch_main<T>(o, pf, pl, device, stream);
            if (rc) return rc;
            rc = prof_record(o, device, 3, stream);
            if (rc) return rc;
        }
        const int lnt = n <= 256 ? 256 : 1024;
        const size_t llds = align16(sizeof(Ctl)) + align16(sizeof(uint16_t) * (size_t)(n + 1)) +
                            align16(sizeof(unsigned) * (size_t)group_size);
        if (llds > kLdsMax) return fail(TSP2OPT_E_UNSUPPORTED, "group_size too large");
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_group_lockstep<T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
        hipLaunchKernelGGL(k_group_lockstep<T>, dim3((unsigned)groups), dim3(lnt), llds, stream, pf,
                           target, lockstep_only ? 1 : 0);
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

// argument checks shared by tsp2opt_solve and tsp2opt_solve_noalloc (no HIP call)
int check_solve_args(const void* points, const int64_t* tours, const int32_t* iters_out,
                     const int32_t* sweeps_out, const double* lengths_out, const int32_t* status_out,
                     int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t group_size,
                     int32_t max_iterations, double time_limit_s, int32_t dtype)
{
    if (num_tours < 0 || n < 1) return fail(TSP2OPT_E_ARG, "num_tours < 0 or n < 1");
    if (num_tours > 0 && (!points || !tours)) return fail(TSP2OPT_E_ARG, "null points/tours");
    if (n > TSP2OPT_MAX_N) return fail(TSP2OPT_E_UNSUPPORTED, "n > 65535 is not supported");
    if (tours_per_instance < 1 || group_size < 1) return fail(TSP2OPT_E_ARG, "tours_per_instance/group_size < 1");
    if (tours_per_instance % group_size != 0 || num_tours % group_size != 0)
        return fail(TSP2OPT_E_ARG, "group_size must divide tours_per_instance and num_tours");
    if (dtype != TSP2OPT_DTYPE_F32 && dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    if (max_iterations < 0) return fail(TSP2OPT_E_ARG, "max_iterations < 0");
    // the reference has a wall-clock limit only in the plugin, whose batch is one tour
    // (nar/local_search/two_opt.py:33,63-65); the coupled legacy call has none
    if (time_limit_s > 0.0 && group_size > 1)
        return fail(TSP2OPT_E_ARG, "time_limit_s needs group_size == 1 (the coupled call has no time limit)");
    // the kernels use naturally aligned vector accesses and atomics: a misaligned pointer would fault
    const uintptr_t esz = dtype == TSP2OPT_DTYPE_F64 ? 8 : 4;
    if ((reinterpret_cast<uintptr_t>(points) % esz) || (reinterpret_cast<uintptr_t>(tours) % 8) ||
        (reinterpret_cast<uintptr_t>(iters_out) % 4) || (reinterpret_cast<uintptr_t>(sweeps_out) % 4) ||
        (reinterpret_cast<uintptr_t>(status_out) % 4) || (reinterpret_cast<uintptr_t>(lengths_out) % 8))
        return fail(TSP2OPT_E_ARG, "misaligned pointer (arrays must be aligned to their element size)");
    return 0;
}

int solve_any(const Opts& o, const void* points, int64_t* tours, int32_t* iters_out, int32_t* sweeps_out,
              double* lengths_out, int32_t* status_out, int64_t num_tours, int32_t n, int32_t tpi,
              int32_t group_size, int32_t max_iterations, double time_limit_s, int32_t dtype, int32_t device,
              void* ws, hipStream_t stream)
{
    // two_opt.py:40-48: the first sweep still runs and, if it finds an improvement, is applied
    // before the cap is tested; a cap of 0 therefore behaves like a cap of 1.
    if (max_iterations == 0) max_iterations = 1;
    if (dtype == TSP2OPT_DTYPE_F64)
        return solve_t<double>(o, points, tours, iters_out, sweeps_out, lengths_out, status_out, num_tours,
                               n, tpi, group_size, max_iterations, time_limit_s, dtype, device, ws, stream);
    return solve_t<float>(o, points, tours, iters_out, sweeps_out, lengths_out, status_out, num_tours,
                          n, tpi, group_size, max_iterations, time_limit_s, dtype, device, ws, stream);
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
    const Opts o = t_opts;
    int rc = check_solve_args(points, tours, iters_out, sweeps_out, lengths_out, status_out, num_tours, n,
                              tours_per_instance, group_size, max_iterations, time_limit_s, dtype);
    if (rc) return rc;
    if (reinterpret_cast<uintptr_t>(workspace) % 256)
        return fail(TSP2OPT_E_ARG, "workspace must be 256-byte aligned");
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
    rc = solve_any(o, points, tours, iters_out, sweeps_out, lengths_out, status_out, num_tours, n,
                   tours_per_instance, group_size, max_iterations, time_limit_s, dtype, device, ws, stream);
    if (own) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(ws);
    }
    return rc;
}

extern "C" int tsp2opt_init(int32_t device)
{
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    return scratch_setup(device, nullptr);
}

extern "C" int64_t tsp2opt_noalloc_max_bytes(void) { return (int64_t)kScratchBytes; }

extern "C" int tsp2opt_solve_noalloc(const void* points, int64_t* tours, int32_t* iters_out,
                                     int32_t* sweeps_out, double* lengths_out, int32_t* status_out,
                                     int64_t num_tours, int32_t n, int32_t tours_per_instance,
                                     int32_t group_size, int32_t max_iterations, double time_limit_s,
                                     int32_t dtype, int32_t device, void* hip_stream)
{
    Opts o = t_opts;
    o.profile = 0;                              // profiling creates HIP events: not on this path
    int rc = check_solve_args(points, tours, iters_out, sweeps_out, lengths_out, status_out, num_tours, n,
                              tours_per_instance, group_size, max_iterations, time_limit_s, dtype);
    if (rc) return rc;
    const size_t need = ws_layout(num_tours, group_size).total;
    if (need > kScratchBytes)
        return fail(TSP2OPT_E_WORKSPACE,
                    "batch too large for tsp2opt_solve_noalloc: tsp2opt_workspace_bytes(...) exceeds "
                    "tsp2opt_noalloc_max_bytes() (2 MiB, about 100 000 tours); use tsp2opt_solve");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
    void* ws = nullptr;
    rc = scratch_setup(device, &ws);            // first call on a device only; tsp2opt_init does it ahead
    if (rc) return rc;
    return solve_any(o, points, tours, iters_out, sweeps_out, lengths_out, status_out, num_tours, n,
                     tours_per_instance, group_size, max_iterations, time_limit_s, dtype, device, ws,
                     static_cast<hipStream_t>(hip_stream));
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
    t_opts.profile = on ? 1 : 0;
    return 0;
}

extern "C" int tsp2opt_profile_history_ms(int32_t device, int32_t calls_back, double* main_ms, double* forced_ms)
{
    if (device < 0 || device >= kMaxDev || !main_ms || calls_back < 0 || calls_back >= kProfRing)
        return fail(TSP2OPT_E_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(g_prof_mu[device]);
    ProfSlot& slot = g_prof[device];
    if (slot.calls - calls_back < 1) return fail(TSP2OPT_E_ARG, "no such profiled call on this device");
    ProfEntry& s = slot.ring[(slot.calls - 1 - calls_back) % kProfRing];
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

extern "C" int tsp2opt_profile_last_ms(int32_t device, double* main_ms, double* forced_ms)
{
    return tsp2opt_profile_history_ms(device, 0, main_ms, forced_ms);
}

// Diagnostic: with on != 0, wave 0 lane 0 of every workgroup accumulates shader-clock cycles per
// phase (filter sweep, argmin selection incl. exact re-evaluation, reversal, staging, whole tour)
// into workspace bytes [64,104); read them back from the workspace after a sync.
extern "C" int tsp2opt_debug_phase_cycles(int32_t on)
{
    t_opts.stamps = on ? 1 : 0;
    return 0;
}

// Diagnostic: sweeps > 0 makes tsp2opt_solve run only `sweeps` filter sweeps per tour (no argmin
// selection, no reversal; tours come back unchanged, iters_out = sweeps).  Measures the hot loop alone.
extern "C" int tsp2opt_debug_sweep_only(int32_t sweeps)
{
    t_opts.dbg_sweeps = sweeps > 0 ? sweeps : 0;
    return 0;
}

// Test/diagnostic knob: -1 automatic choice between one workgroup per tour and a cluster of
// workgroups per tour, 0 never cluster, k >= 2 use up to k workgroups per tour whenever possible.
extern "C" int tsp2opt_debug_cluster_mode(int32_t mode)
{
    t_opts.cluster_mode = mode;
    return 0;
}

extern "C" int tsp2opt_debug_cluster_inc(int32_t on)
{
    t_opts.cluster_inc = on ? 1 : 0;
    return 0;
}

extern "C" int tsp2opt_debug_tune(int32_t what, int32_t value)
{
    Opts& o = t_opts;
    if (what == 0) o.force_nt = value;
    else if (what == 1) o.lcap16 = value < 1 ? 1 : value > 16 ? 16 : value;
    else if (what == 2) o.inc_min_tiles = value;
    else if (what == 3) o.cluster_variant = value;
    else if (what == 4) o.lcap16s = value < 1 ? 1 : value > 16 ? 16 : value;
    else if (what == 5) o.lds_pad = value < 0 ? 0 : value;
    else if (what == 6) o.force_lockstep = value ? 1 : 0;
    else if (what == 7) o.part_sh = value < 0 ? 0 : value > 6 ? 6 : value;
    else if (what == 8) o.half_units = value ? 1 : 0;
    else if (what == 9) o.wide_few = value ? 1 : 0;
    else if (what == 13) o.cl_u0 = value < 0 ? 0 : value;
    else if (what == 10) o.extra_waves = value < 0 ? 0 : value > 12 ? 12 : value;
    else return fail(TSP2OPT_E_ARG, "unknown tuning knob");
    return 0;
}

extern "C" int tsp2opt_cluster_plan(int32_t n, int32_t dtype, int64_t num_tours, int32_t cus, int32_t* out12)
{
    const Opts o = t_opts;
    if (n < 1 || n > TSP2OPT_MAX_N || num_tours < 1 || cus < 1 || !out12) return fail(TSP2OPT_E_ARG, "bad argument");
    if (dtype != TSP2OPT_DTYPE_F32 && dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    const Plan pl = make_plan(o, n, dtype);
    for (int k = 0; k < 12; ++k) out12[k] = 0;
    if (o.cluster_mode == 0) return 0;
    const ClusterChoice cc = choose_cluster(o, n, dtype, num_tours, cus, pl);
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
    const Opts o = t_opts;
    if (n < 1 || n > TSP2OPT_MAX_N || !out7) return fail(TSP2OPT_E_ARG, "bad argument");
    if (dtype != TSP2OPT_DTYPE_F32 && dtype != TSP2OPT_DTYPE_F64) return fail(TSP2OPT_E_ARG, "bad dtype");
    const Plan pl = make_plan(o, n, dtype);
    out7[0] = pl.g.nt; out7[1] = pl.g.qclass; out7[2] = (int32_t)pl.lds;
    out7[3] = pl.g.NS; out7[4] = pl.g.G; out7[5] = pl.g.MS; out7[6] = 0;
    if (device >= 0) {
        DeviceGuard guard(device);
        if (!guard.ok) return fail(TSP2OPT_E_HIP, "hipSetDevice failed");
        Params p{};
        p.g = pl.g;
        int bpc = 0;
        const int rc = dtype == TSP2OPT_DTYPE_F64 ? dispatch_main<double>(o, p, pl, device, nullptr, &bpc)
                                                  : dispatch_main<float>(o, p, pl, device, nullptr, &bpc);
        if (rc) return rc;
        out7[6] = bpc;
    }
    return 0;
}
