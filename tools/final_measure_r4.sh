#!/bin/bash
# Round-4 measurement pass on the GPU box (run from the repo root): what profiles/r04_* cite for the final build, into gpurun_out/final4/.  Every file is
# keyed with the hash of the library it was taken on (tools/collect_profiles_r4.py writes lib_sha16 into each).
#   part 1 (bash tools/final_measure_r4.sh 1): bench.py; per workload (north star quadrotor N=20 x 8192, BASELINE config 3 quadrotor N=50 x 8192, config 4
#           cart-pole N=100 x 16384) rocprofv3 kernel stats of the timed region, FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS per solve in separate passes
#   part 2 (… 2): timing build break-downs, wave-cycle counter groups, 2-rank rehearsal, SQP loop, fuzz sweeps of the on-chip families
set -o pipefail
out=gpurun_out/final4; mkdir -p $out; export TMPDIR=/tmp
part=${1:-1}
WL=("q20 --workload quadrotor" "q50 --workload quadrotor --horizon 50 --batch 8192" "cp100 --workload cartpole")
if [ "$part" = "1" ]; then
  python bench.py > $out/bench.json 2> $out/bench.err || exit 1
  tail -c 300 $out/bench.json; echo
  for cfg in "${WL[@]}"; do
    set -- $cfg; tag=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats_$tag -- python3 bench.py "$@" --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $out/kstats_$tag.log 2>&1 || exit 1
    rm -f $out/pmc_$tag.txt
    for ctr in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
      rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_${tag}_$ctr -- python3 bench.py "$@" --steps 3 --warmup 2 --no-extras --no-cpu-baseline > $out/pmc_${tag}_$ctr.log 2>&1 || exit 1
      python tools/pmc_summary.py $out/pmc_${tag}_$ctr $ctr mpcqp_oc_ 5 >> $out/pmc_$tag.txt
    done
    echo "== $tag"; cut -c1-400 $out/pmc_$tag.txt; grep -h "mpcqp_oc_" $out/kstats_$tag/*/*kernel_stats.csv | cut -c1-200 | head -4
  done
else
  export MPCQP_LIB=optimal_control_problem_amd/libmpcqp_timing.so
  python tools/timing_breakdown.py quadrotor 8192 > $out/timing_breakdown_q20.txt 2>&1
  python tools/timing_breakdown.py quadrotor 8192 - 50 > $out/timing_breakdown_q50.txt 2>&1
  python tools/timing_breakdown.py cartpole 16384 - 100 > $out/timing_breakdown_cp100.txt 2>&1
  unset MPCQP_LIB
  for cfg in "${WL[@]}"; do set -- $cfg; tag=$1; shift; bash tools/pmc_groups.sh $tag "$@" > /dev/null 2>&1; cp gpurun_out/pmcg_$tag/summary.txt $out/pmcg_$tag.txt; done
  python bench.py --gpus 2 --share-gpu --no-extras --no-cpu-baseline --batch 4096 --steps 5 > $out/bench_gpus2.json 2> $out/bench_gpus2.err
  python tools/sqp_bench.py quadrotor 20 8192 10 0.5 1 16 > $out/sqp_device_loop.json 2> $out/sqp_device_loop.err
  python tools/fuzz_oc.py 60 11000 oc4 > $out/fuzz_oc4.txt 2>&1
  python tools/fuzz_oc.py 40 12000 oc8 > $out/fuzz_oc8.txt 2>&1
  python tools/fuzz_gpu.py 150 13000 > $out/fuzz_gpu.txt 2>&1
fi
echo done
