#!/bin/bash
# tools/profile_round.sh TAG [stage ...] -- evidence for profiles/: bench lines, rocprofv3 kernel
# stats, PMC passes, phase breakdowns.  Stages: bench kt pmc configs ktconfigs variants phases dist strong
# pipeline lsops pmcconfigs pmcdecoder (default all).
# Everything lands in gpurun_out/TAG_*.  Run on the GPU box from the repo root.
set -u
TAG=${1:-rXX}; shift || true
STAGES=${*:-bench kt pmc configs ktconfigs pmcconfigs pmcdecoder variants phases dist strong pipeline lsops}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
for st in $STAGES; do
  echo "== stage $st"
  case $st in
  bench) timeout -k 10 200 python bench.py > $O/${TAG}_bench_tsp500.json 2> $O/${TAG}_bench_tsp500.err || exit 1 ;;
  kt) timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python bench.py --steps 4 --warmup 1 --no-cpu --no-configs > $O/${TAG}_kt.log 2>&1 || exit 1
      cp $(find $O/${TAG}_kt -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats.csv ;;
  pmc) bash tools/pmc_passes.sh || exit 1
       python tools/summarize_pmc.py > $O/${TAG}_tsp500_pmc_summary.json || exit 1 ;;
  configs) for c in tsp50 tsp100 tsp1000 tsp10000; do
      echo "   bench $c"
      timeout -k 10 250 python bench.py --config $c --steps 5 --warmup 2 > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err || exit 1
    done ;;
  ktconfigs) for c in tsp100 tsp1000 tsp10000; do
      echo "   kernel trace $c"
      timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_kt_$c -- python bench.py --config $c --steps 2 --warmup 1 --no-cpu > $O/${TAG}_kt_$c.log 2>&1 || exit 1
      cp $(find /tmp/${TAG}_kt_$c -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_$c.csv
    done ;;
  pmcconfigs) # one SQ pass per other configuration (needs the configs stage's bench lines for the iteration counts)
    for ct in tsp50:1280 tsp100:10000 tsp1000:128 tsp10000:16; do
      c=${ct%%:*}; t=${ct##*:}
      echo "   pmc $c"
      timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/${TAG}_pv_$c -- python bench.py --config $c --steps 1 --warmup 0 --no-cpu > $O/${TAG}_pv_$c.log 2>&1 || exit 1
      python tools/summarize_pmc_valu.py $c $(find /tmp/${TAG}_pv_$c -name '*counter_collection.csv' | head -1) $O/${TAG}_bench_$c.json $t > $O/${TAG}_${c}_pmc_valu.json || exit 1
    done ;;
  pmcdecoder) # counters of the dense decoder's kernels (TSP-500 x 2048 heat maps)
    n=0
    for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
      n=$((n+1))
      timeout -s KILL 150 rocprofv3 --pmc $pass --output-format csv -d /tmp/${TAG}_pd$n -- python tools/pipeline_bench.py --config tsp500 --check 0 --steps 1 > $O/${TAG}_pd$n.log 2>&1 || exit 1
    done
    python tools/summarize_pmc_decoder.py $(find /tmp/${TAG}_pd1 -name '*counter_collection.csv' | head -1) $(find /tmp/${TAG}_pd2 -name '*counter_collection.csv' | head -1) $(find /tmp/${TAG}_pd3 -name '*counter_collection.csv' | head -1) > $O/${TAG}_decoder_tsp500_pmc.json || exit 1 ;;
  pipeline) for c in tsp50 tsp100 tsp500; do
      echo "   pipeline $c (dense heat maps)"
      timeout -k 10 250 python tools/pipeline_bench.py --config $c --steps 5 > $O/${TAG}_pipeline_$c.json 2> $O/${TAG}_pipeline_$c.err || exit 1
    done
    for c in tsp1000 tsp10000; do
      echo "   pipeline $c (k = 50 lists)"
      timeout -k 10 250 python tools/pipeline_bench.py --config $c --knn 50 --sparse --steps 5 --check 1 > $O/${TAG}_pipeline_sparse_$c.json 2> $O/${TAG}_pipeline_sparse_$c.err || exit 1
    done ;;
  lsops) timeout -k 10 250 python tools/ls_bench.py > $O/${TAG}_ls_ops_bench.txt 2>&1 || exit 1 ;;
  variants) timeout -k 10 200 python bench.py --mode grouped --no-cpu > $O/${TAG}_bench_tsp500_grouped.json 2> $O/${TAG}_bench_tsp500_grouped.err || exit 1
      timeout -k 10 200 python bench.py --dtype f32 --no-cpu > $O/${TAG}_bench_tsp500_f32.json 2> $O/${TAG}_bench_tsp500_f32.err || exit 1 ;;
  phases) # the phase timers only exist in the diagnostic build (python -m ml4tspbench_amd.build --fine)
    for c in tsp50 tsp100 tsp500 tsp1000; do
      echo "   phases $c"
      TSP2OPT_LIB=$PWD/ml4tspbench_amd/libtsp2opt_hip_fine.so timeout -k 10 120 python tools/phase_breakdown.py $c > $O/${TAG}_phase_$c.txt 2>&1 || exit 1
    done
    TSP2OPT_LIB=$PWD/ml4tspbench_amd/libtsp2opt_hip_fine.so timeout -k 10 200 python tools/phase_breakdown.py tsp10000 f64 1000 > $O/${TAG}_phase_tsp10000.txt 2>&1 || exit 1 ;;
  strong) timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu --no-configs --scaling strong > $O/${TAG}_bench_strong1.json 2> $O/${TAG}_bench_strong1.err || exit 1 ;;
  dist) timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu > $O/${TAG}_bench_dist1.json 2> $O/${TAG}_bench_dist1.err || exit 1 ;;
  esac
done
echo profile_round done
