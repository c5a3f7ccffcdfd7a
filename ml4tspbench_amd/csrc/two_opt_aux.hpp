// ml4tspbench_amd/csrc/two_opt_aux.hpp -- coupled-mode glue, tour lengths, validation/evaluation and numerics self-test kernels
// Part of the single translation unit tsp2opt.hip (design notes at its top and in DESIGN.md).
#pragma once

namespace t2o {

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
    if (st & (TSP2OPT_ST_BAD_INDEX | TSP2OPT_ST_NOT_CLOSED | TSP2OPT_ST_NONFINITE | TSP2OPT_ST_RANGE |
              TSP2OPT_ST_INTERNAL))
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

}  // namespace t2o
