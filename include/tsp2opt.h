/*
 * include/tsp2opt.h -- C ABI of libtsp2opt_hip.so
 *
 * MI355X (gfx950) implementation of ML4TSPBench's batched best-improvement 2-opt local search.
 * The calling convention follows the reference's own native-extension convention
 * (search/c_mcts/mcts.cpp:8-36 + search/c_mcts/__init__.py:11-29): extern "C", raw pointers and
 * integers, loaded with ctypes.CDLL, argtypes/restype declared on the Python side.
 * Unlike that code the library never returns library-owned memory: every buffer belongs to the
 * caller (PyTorch-ROCm tensors in practice) and all pointers except `hip_stream` are DEVICE
 * pointers of contiguous arrays on `device`.
 *
 * Every entry point returns 0 on success or a negative TSP2OPT_E_* code; the message for the
 * calling thread's last failure is tsp2opt_last_error().  No C++ exception crosses the ABI.
 */
#ifndef TSP2OPT_H_
#define TSP2OPT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSP2OPT_VERSION 100            /* 0.1.0 */

#define TSP2OPT_DTYPE_F32 0            /* arithmetic in float, like the plugin with fp32 points */
#define TSP2OPT_DTYPE_F64 1            /* arithmetic in double, like tsp_2opt (two_opt.py:67)   */

/* return codes */
#define TSP2OPT_OK              0
#define TSP2OPT_E_ARG         (-1)     /* bad argument (null pointer, n < 1, sizes inconsistent) */
#define TSP2OPT_E_UNSUPPORTED (-2)     /* n above the supported maximum, or an input this entry point does not take (see there) */
#define TSP2OPT_E_WORKSPACE   (-3)     /* workspace_bytes too small                              */
#define TSP2OPT_E_HIP         (-4)     /* a HIP runtime call failed (message has the HIP error)  */

/* per-tour status bits written to `status_out` */
#define TSP2OPT_ST_BAD_INDEX    1      /* a tour entry is outside [0, n): tour left untouched    */
#define TSP2OPT_ST_NOT_CLOSED   2      /* tour[n] != tour[0]: tour left untouched                */
#define TSP2OPT_ST_NONFINITE    4      /* instance has a NaN/Inf coordinate: 0 iterations, like
                                          the reference whose NaN minimum ends the loop          */
#define TSP2OPT_ST_RANGE        8      /* |coordinate| > 1e150: outside the supported range      */
#define TSP2OPT_ST_CAPPED      16      /* stopped by max_iterations                              */
#define TSP2OPT_ST_TIME        32      /* stopped by time_limit_s                                */
#define TSP2OPT_ST_INTERNAL    64      /* multi-workgroup barrier timed out (device oversubscribed
                                          by another kernel): tour left untouched, retry          */

/* Largest n the kernels accept (tour positions are held as 16-bit values on chip). */
#define TSP2OPT_MAX_N 65535

/*
 * tsp2opt_solve -- replaces the hot loop of
 *   batched_two_opt_torch            search/two_opt.py:21-48           (group_size = batch rows)
 *   ML4TSPNARTwoOpt._local_search    ml4tsp-library/ml4tsp/nar/local_search/two_opt.py:34-65
 *   ML4TSPNARLocalSearch.local_search's per-tour Python loop
 *                                    ml4tsp-library/ml4tsp/nar/local_search/base.py:58-64
 * for ALL tours of a batch in one call: the whole improvement loop runs on the device, no
 * per-iteration host round trip.
 *
 *   points   (num_instances, n, 2) float or double, node order, row-major
 *   tours    (num_tours, n+1) int64, tour[n] == tour[0]; UPDATED IN PLACE (pass a clone: the
 *            reference never mutates its input, two_opt.py:12)
 *   iters_out  (num_tours) int32 or NULL: the reference's `iterator` of the call the tour belongs
 *            to (its own count when group_size == 1, the group's shared count otherwise)
 *   sweeps_out (num_tours) int32 or NULL: full (i,j) sweeps the reference algorithm performs for
 *            that result: iterator+1 when it stopped on convergence, iterator when capped
 *   lengths_out (num_tours) double or NULL: closed-tour length of the result, summed in double
 *   status_out (num_tours) int32 or NULL: TSP2OPT_ST_* bits
 *   tours_per_instance  tour k belongs to instance k / tours_per_instance
 *            (base.py:61: idx // per_tours_num // per_heatmap_num)
 *   group_size  1: every tour stops on its own minimum (the plugin, two_opt.py:33 batch_size = 1)
 *               P: consecutive P tours stop together on the minimum over the whole group
 *                  (search/two_opt.py:35,40); needs tours_per_instance % P == 0, num_tours % P == 0
 *   max_iterations  two_opt.py:47 / max_iterations_2opt (default 5000 in both front ends)
 *   time_limit_s    <= 0: none ("auto"); > 0: per-tour wall-clock limit checked once per iteration
 *                   (plugin two_opt.py:63-65); only honoured when group_size == 1
 *   dtype     TSP2OPT_DTYPE_F32 / _F64: dtype of `points` AND of the arithmetic
 *   device    HIP device ordinal that owns the buffers
 *   workspace / workspace_bytes  device scratch of at least tsp2opt_workspace_bytes(...) bytes,
 *            256-byte aligned; NULL: the library allocates and frees it on `hip_stream`.
 *            Arrays must be aligned to their element size; misaligned pointers are rejected
 *            (TSP2OPT_E_ARG) rather than left to fault in a kernel.
 *   hip_stream  hipStream_t to enqueue on (torch.cuda.current_stream(device).cuda_stream); the
 *            call is asynchronous with respect to the host
 *
 * Results: bit-identical tours to the reference algorithm evaluated with IEEE-754 correctly
 * rounded +,-,*,sqrt in `dtype` and first-flat-index argmin tie-breaking.
 */
int tsp2opt_solve(const void* points, int64_t* tours, int32_t* iters_out, int32_t* sweeps_out,
                  double* lengths_out, int32_t* status_out, int64_t num_tours, int32_t n,
                  int32_t tours_per_instance, int32_t group_size, int32_t max_iterations,
                  double time_limit_s, int32_t dtype, int32_t device, void* workspace,
                  int64_t workspace_bytes, void* hip_stream);

/* Bytes of device scratch tsp2opt_solve needs for this problem size (> 0), or a negative code. */
int64_t tsp2opt_workspace_bytes(int64_t num_tours, int32_t n, int32_t group_size);

/*
 * tsp2opt_solve_noalloc -- tsp2opt_solve for callers that cannot allocate.  No counterpart in the
 * reference: it exists for callers that may not call hipMalloc (code under HIP graph capture, tight
 * serving loops, harnesses that forbid it).  Same arguments, semantics and results as tsp2opt_solve
 * minus `workspace`/`workspace_bytes`: the scratch (control words, per-group targets, per-tour
 * counters, the cluster barrier area) lives in a 2 MiB device array inside the library, which every GPU
 * the library runs on holds until the process ends.  The call allocates and frees nothing (no
 * hipMalloc/hipFree, no event or stream creation) and ignores tsp2opt_profile_enable.
 *   - Limit: tsp2opt_workspace_bytes(num_tours, n, group_size) <= tsp2opt_noalloc_max_bytes()
 *     (about 100 000 tours); larger batches fail with TSP2OPT_E_WORKSPACE.
 *   - One-time set-up per device (resolving the array's device address, creating the event that
 *     orders multi-workgroup launches) happens in tsp2opt_init(device), or on the first call on that
 *     device if tsp2opt_init was not called; call tsp2opt_init first when the first call would sit
 *     inside a graph capture.
 *   - Every call on a GPU shares that GPU's array: calls that overlap in time on one GPU (two streams,
 *     two threads) race.  Issue them on one stream, or order them with events.
 *   - Under stream capture the call records only memsets and kernel launches (the per-device event that
 *     orders multi-workgroup launches of different streams is neither waited on nor recorded while the
 *     stream captures): replays of the graph must not overlap other multi-workgroup calls on that GPU.
 */
int tsp2opt_solve_noalloc(const void* points, int64_t* tours, int32_t* iters_out, int32_t* sweeps_out,
                          double* lengths_out, int32_t* status_out, int64_t num_tours, int32_t n,
                          int32_t tours_per_instance, int32_t group_size, int32_t max_iterations,
                          double time_limit_s, int32_t dtype, int32_t device, void* hip_stream);
int tsp2opt_init(int32_t device);
int64_t tsp2opt_noalloc_max_bytes(void);

/*
 * tsp2opt_tour_lengths -- closed-tour length of every tour, summed in double from `dtype`
 * coordinates.  Replaces the per-edge Python loop of TSPEvaluator.evaluate
 * (utils/evaluator.py:5-13) for a whole batch; used for the final "tours + lengths" gather.
 */
int tsp2opt_tour_lengths(const void* points, const int64_t* tours, double* lengths_out,
                         int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t dtype,
                         int32_t device, void* hip_stream);

/*
 * tsp2opt_evaluate -- the step right after the local search, for a whole batch on the device:
 *   valid_out[k]  bit 0: tour k's first n entries are a permutation of 0..n-1
 *                        (ML4TSPNARLocalSearch.is_valid_tour, nar/local_search/base.py:31-32,67-70;
 *                         `assert sorted(tour[:-1]) == range(N)`, models/modelzoo/gnn/tsp_gnn.py:214-215)
 *                 bit 1: tour[n] == 0 (base.py:71);   bit 2: tour[n] == tour[0]
 *   lengths_out[k] closed-tour length in double (TSPEvaluator.evaluate, utils/evaluator.py:5-13)
 *   best_index_out[i] / best_length_out[i] (may be NULL): first minimal-length tour among instance
 *                 i's tours_per_instance tours, as a global tour index (np.min over the parallel
 *                 samples, tsp_gnn.py:220-221)
 * Entries outside [0, n) make the tour invalid and its length NaN; nothing is dereferenced out of
 * range.
 */
int tsp2opt_evaluate(const void* points, const int64_t* tours, double* lengths_out,
                     int32_t* valid_out, int64_t* best_index_out, double* best_length_out,
                     int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t dtype,
                     int32_t device, void* hip_stream);

/*
 * tsp2opt_greedy_decode -- greedy edge-merge decoder, the step right before the local search.
 * Replaces tsp_greedy (search/greedy_search.py:19-74) + merge_cython
 * (search/cython_merge/sources/cython_merge.pyx:19-120) for a whole batch of heat maps:
 *   heat      (num_maps, n, n) float or double, dense (the reference densifies sparse heat maps
 *             first, greedy_search.py:41-47); NOT symmetrised (A + A^T is formed here, :36-38)
 *   points    (num_instances, n, 2) double (the reference casts to double, greedy_search.py:15)
 *   tours_out (num_maps, n+1) int64: tour from node 0 back to node 0
 *   maps_per_instance  heat map k belongs to instance k / maps_per_instance (parallel_sampling)
 *   workspace at least tsp2opt_greedy_workspace_bytes(...) bytes, 256-byte aligned (4 n^2 + 28 n bytes per
 *            heat map: fp32 filter keys, candidate lists, neighbour table; unused for n <= 128)
 * Edges are taken in increasing order of -A_ij/|v_i-v_j| (double); exactly equal keys of different
 * edges are ordered by flat index i*n+j (the reference's order for ties is numpy's unstable argsort:
 * undefined).  Keys are compared as fp32 approximations with a margin and exactly (double) inside it: the
 * tours are those of the exact order.  n <= 10200 (16-bit node ids; a heat map's merge state, 12 B per node,
 * lives in the LDS of its workgroup).  Asynchronous on hip_stream.
 */
int tsp2opt_greedy_decode(const void* heat, const double* points, int64_t* tours_out,
                          int64_t num_maps, int32_t n, int32_t maps_per_instance, int32_t heat_dtype,
                          int32_t device, void* workspace, int64_t workspace_bytes, void* hip_stream);
int64_t tsp2opt_greedy_workspace_bytes(int64_t num_maps, int32_t n, int32_t heat_dtype);

/*
 * tsp2opt_greedy_decode_sparse -- the same decoder for SPARSE heat maps, without densifying them.
 * Replaces the sparse branch of tsp_greedy (search/greedy_search.py:39-47: two scipy coo matrices turned into
 * dense N x N arrays and added) + merge_cython for a whole batch:
 *   values     (num_maps, num_edges) float or double: heat of every listed edge, per heat map
 *   edge_index (num_instances, 2, num_edges) int64, row-major, num_instances = ceil(num_maps /
 *              maps_per_instance): edge e of instance q is (ei[q][0][e], ei[q][1][e]); one list per instance
 *              (the kNN graph of utils/utils.py:168-189, whose length n*k is the same for every instance),
 *              shared by the maps_per_instance heat maps of that instance; duplicates and both directions of
 *              a pair are summed like the reference's coo matrices do.  The reference decodes one instance
 *              per call (num_instances = 1); a batch runs all its instances side by side
 *   points, tours_out, maps_per_instance: as tsp2opt_greedy_decode
 * Every node keeps the complete, exactly ordered list of its listed partners (workspace: 12 B per listed
 * edge and instance + 12 B per listed edge and heat map, instead of 8 n^2 per heat map); pairs without a positive
 * sum (not listed, or listed with zeros) all have the key -0.0 and are taken, after every listed edge, in
 * flat-index order -- the tie rule of the dense entry point and of the oracle (the reference's order among equal
 * keys is numpy's unstable argsort) -- and last of all, in the same order, those among them whose two points
 * coincide (their key -0.0 / 0 is NaN, which numpy sorts last).
 * A listed pair whose heat sum is negative (its key would sort after the unlisted pairs) or a node with more
 * than 7000 listed edges gives TSP2OPT_E_UNSUPPORTED: densify and call tsp2opt_greedy_decode.  n <= 10900.
 * The call synchronises `hip_stream` (it reads back an error word).  Diagnostics of heat map 0 are left in
 * the workspace at byte 64 as eight uint64: 10 ns ticks of set-up / listed-edge loop / unlisted pairs /
 * closing walk, then rounds, rounds that fell back to one exactly compared edge, 0, merges made from listed
 * edges.
 */
int tsp2opt_greedy_decode_sparse(const void* values, const int64_t* edge_index, const double* points,
                                 int64_t* tours_out, int64_t num_maps, int32_t n, int64_t num_edges,
                                 int32_t maps_per_instance, int32_t heat_dtype, int32_t device, void* workspace,
                                 int64_t workspace_bytes, void* hip_stream);
int64_t tsp2opt_greedy_sparse_workspace_bytes(int64_t num_maps, int32_t n, int64_t num_edges,
                                              int32_t maps_per_instance, int32_t heat_dtype);

/*
 * tsp2opt_ls_step -- one call of the reference's other move operators for every tour of a batch:
 *   op 0: get_two_opt_tour  (search/local_search.py:27-53; cost and reversal as written there: the cost ADDS
 *         the weight of edge (t[i-1], t[i]), local_search.py:23, and swap() reverses positions i .. j-1)
 *   op 1: get_relocate_tour (search/local_search.py:75-105 = search/relocate.py:23-38, which tsp_relocate
 *         applies once per tour, relocate.py:40-61)
 * The best move -- first strict improvement in the reference's scan order, candidates with
 * numpy.isclose(0, delta) skipped -- is applied in place; delta_out[k] is its delta (0.0: no move).
 *   points   (num_instances, n, 2) double (scipy.spatial.distance_matrix of float64 points)
 *   tours    (num_tours, n+1) int64, updated in place
 *   fixed_i  (num_tours) int32 or NULL: the operators' fixed_i argument (guided local search)
 *   guide_k / penalty  (num_tours) double and (num_tours, n, n) double, or both NULL: weights
 *            dist + guide_k * penalty (edge_weight_guided, search/local_search.py:200)
 *   status_out (num_tours) int32 or NULL: TSP2OPT_ST_BAD_INDEX (tour untouched) or 0
 * tsp2opt_local_search -- local_search() (search/local_search.py:125-134): both operators in turn
 * until a whole round improves nothing; rounds_out[k] counts the rounds, TSP2OPT_ST_CAPPED if max_rounds
 * stopped it.  All arithmetic is exact double in the reference's operation order: tours are bit-identical.
 */
int tsp2opt_ls_step(const double* points, int64_t* tours, double* delta_out, int32_t* status_out,
                    int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t op,
                    const int32_t* fixed_i, const double* guide_k, const double* penalty, int32_t device,
                    void* hip_stream);
int tsp2opt_local_search(const double* points, int64_t* tours, int32_t* rounds_out, int32_t* status_out,
                         int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t max_rounds,
                         int32_t device, void* hip_stream);

/*
 * tsp2opt_gls_rounds -- (regret) guided local search, whole outer rounds on the device, for every tour of a batch.
 * Replaces the loop body of tsp_gls (search/local_search.py:176-231; the same loop is ML4TSPNARGuidedLS._local_search,
 * ml4tsp-library/ml4tsp/nar/local_search/guided_ls.py:23-88): per round the perturbation loop -- the tour edge of
 * maximum utility regret / (1 + penalty) (first among equals, :192-202) is penalised (:206-208) and get_two_opt_tour /
 * get_relocate_tour run with its end nodes as fixed rows under the weights dist + k * penalty (:209-223) until
 * `perturbation_moves` moves were made -- then local_search under plain distances (:226) and the comparison that
 * keeps the best tour (:227-229).  The reference's wall-clock test (:187) is the caller's: `rounds` rounds per call.
 *   tours      (num_tours, n+1) int64: the CURRENT tour of every search, in / out
 *   best_tours (num_tours, n+1) int64, best_cost (num_tours) double: best tour so far and its cost, in / out
 *   guide_k    (num_tours) double: k = 0.1 * initial cost / n (:179), written by the init call, read afterwards
 *   penalty    (num_tours, n, n) double, in / out (zeros before the init call)
 *   regret     (num_tours / tours_per_map, n, n) double: max(-heat, 0) (:167,177), shared by tours_per_map tours
 *   init != 0  first do what precedes the loop (:178-182): k, a first local_search, best = its result
 *   status_out TSP2OPT_ST_BAD_INDEX (tour untouched), TSP2OPT_ST_CAPPED (a perturbation loop found no guided move
 *              100 000 times in a row, where the reference would spin for ever), or 0
 * Exact double arithmetic in the reference's operation order: tours are bit-identical to the reference's for the
 * same number of rounds.  One workgroup per tour; tour and coordinates stay in LDS for the whole call.
 */
int tsp2opt_gls_rounds(const double* points, int64_t* tours, int64_t* best_tours, double* best_cost,
                       double* guide_k, double* penalty, const double* regret, int32_t* status_out,
                       int64_t num_tours, int32_t n, int32_t tours_per_instance, int32_t tours_per_map,
                       int32_t perturbation_moves, int32_t rounds, int32_t init, int32_t device,
                       void* hip_stream);

/*
 * tsp2opt_sqrt_check -- numerics self-test used by the GPU test-suite: out[k] = sqrt(in[k])
 * computed with the exact device routine the delta evaluation uses (dtype as above).  Lets the
 * tests prove 0-ulp agreement with the host's correctly rounded sqrt (SURVEY.md hard part b).
 */
int tsp2opt_sqrt_check(const void* in, void* out, int64_t count, int32_t dtype, int32_t device,
                       void* hip_stream);

/* Kernel variant / launch geometry the library would pick for n (for logs and tests): the geometry of batches with
 * more tours than compute units (smaller batches get two to four times the threads per tour).
 * out[0]=threads per workgroup, out[1]=fast-path class (0: exact-only kernel), out[2]=LDS bytes,
 * out[3]=row segments, out[4]=diagonal groups, out[5]=macro steps, out[6]=resident workgroups/CU */
int tsp2opt_plan(int32_t n, int32_t dtype, int32_t device, int32_t* out7);

/* Diagnostic, no counterpart in the reference: phase timers of the greedy decoder's merge loop.
 * enable = 1 zeroes and switches them on for `device`; enable = 0 switches them off and copies
 * out8 = {list-advance cycles, row-rescan cycles, choose+apply cycles, merges, rescans, loop cycles, rounds,
 * rounds that inserted one exactly compared edge} summed over the heat maps decoded in between.
 * Synchronises the device.  enable = 3 / 2: decode heat maps of <= 128 nodes with the general kernels instead of
 * the on-chip one / back to the default (for A/B runs).  The switches belong to the calling thread. */
int tsp2opt_debug_greedy_cycles(int32_t enable, int32_t device, uint64_t* out8);

/*
 * Optional timing of the persistent kernel: after tsp2opt_profile_enable(1) every tsp2opt_solve
 * brackets its main kernel (and, in coupled mode, the continuation kernel) with HIP events on the
 * caller's stream; tsp2opt_profile_last_ms waits for them and returns the durations of the latest
 * call on `device` (forced_ms may be NULL).  Measurement aid for bench.py's roofline line; no
 * counterpart in the reference (it only accumulates time.time() deltas, tsp_gnn.py:168,178).
 */
int tsp2opt_profile_enable(int32_t on);
int tsp2opt_profile_last_ms(int32_t device, double* main_ms, double* forced_ms);
/* The same for the call `calls_back` calls before the latest one on `device` (0 = latest, up to 63):
 * lets a timed loop run without synchronising between its steps and read all durations afterwards. */
int tsp2opt_profile_history_ms(int32_t device, int32_t calls_back, double* main_ms, double* forced_ms);

/*
 * Diagnostic build aid (no counterpart in the reference): with on != 0 the persistent kernel also
 * accumulates shader-clock cycles per phase into workspace bytes [64,104) as five uint64:
 * filter sweep, argmin selection (reductions + exact re-evaluation), reversal, staging, whole tour
 * (thread 0 of each workgroup, summed over tours).  Never quote run times of such a launch.
 * Only the diagnostic build of the library (python -m ml4tspbench_amd.build --fine) carries the timers;
 * in the shipped library the call is accepted and has no effect.
 */
int tsp2opt_debug_phase_cycles(int32_t on);
/* Diagnostic: sweeps > 0 makes tsp2opt_solve run only that many filter sweeps per tour (hot loop in
 * isolation: no argmin selection, no reversal; tours unchanged).  0 restores normal operation. */
int tsp2opt_debug_sweep_only(int32_t sweeps);
/* Test knob for the choice between one workgroup per tour and a cluster of workgroups per tour
 * (few tours, or n beyond one workgroup's LDS): -1 automatic (default), 0 never cluster, k >= 2 use
 * up to k workgroups per tour whenever the geometry allows. */
int tsp2opt_debug_cluster_mode(int32_t mode);
/* Diagnostic: on = 0 builds cluster geometries without the group-minimum cache (every sweep of the
 * cluster kernel is a full sweep); default 1. */
int tsp2opt_debug_cluster_inc(int32_t on);
/* Diagnostic tuning knobs (benchmark sweeps only): what = 0: threads per workgroup of the
 * one-workgroup-per-tour kernel (0 = automatic); what = 1: capacity of the cluster kernel's
 * changed-group list in sixteenths of all groups (default 13); what = 2: tiles a lane must own for the
 * tile-minimum cache to be used (default 3); what = 3: cluster kernel variant, 0 automatic, 1 band
 * kernel only, 2 full-window kernel only; what = 4: changed-tile list capacity of the one-workgroup /
 * full-window kernels in sixteenths of the tiles (default 13); what = 5: unused LDS bytes added to the
 * one-workgroup kernel's request (lowers the workgroups resident per CU: occupancy sweeps; default 0);
 * what = 6: 1 replays coupled calls (group_size > 1) entirely in the lock-step kernel (tests; default 0);
 * what = 7: band cluster kernel, log2 of the macro steps per part when a listed cache group is recomputed
 * (default 4); what = 8: 0 plans small tours without the tile cache in whole tiles per lane instead of macro-step
 * units (default 1); what = 9: 0 plans a batch of at most one tour per compute unit like a large batch instead of
 * with two to four times the threads per tour (default 1; results never depend on it: tests run both); what = 10:
 * waves without a unit in that wide geometry, 0..12 (default 4); what = 13: lanes without a unit in a workgroup of the
 * full-window cluster (default 64).  The knobs belong to the calling thread. */
int tsp2opt_debug_tune(int32_t what, int32_t value);
/* Geometry the library would pick for `num_tours` tours of n nodes on a device with `cus` compute
 * units when several workgroups share a tour (host-only, for logs and tests):
 * out[0]=workgroups per tour (0: one workgroup per tour, tsp2opt_plan applies), out[1]=row bands
 * (0: full-window cluster, lane groups interleaved over the workgroups), out[2]=diagonal bands, out[3]=threads, out[4]=lane groups, out[5]=row segments, out[6]=macro steps,
 * out[7]=LDS bytes, out[8]=incremental sweeps, out[9]=macro steps per cache entry, out[10]=cache
 * entries per lane, out[11]=exact coordinates in LDS */
int tsp2opt_cluster_plan(int32_t n, int32_t dtype, int64_t num_tours, int32_t cus, int32_t* out12);

const char* tsp2opt_last_error(void);
int tsp2opt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TSP2OPT_H_ */
