#!/bin/bash
# Round-3 measurement pass on the GPU box (run from the repo root): what profiles/r03_* cite for the final build, into gpurun_out/final3/
#   per workload (north star quadrotor N=20 x 8192, BASELINE config 3 quadrotor N=50 x 8192, config 4 cart-pole N=100 x 16384):
#   rocprofv3 kernel stats of the timed region, FETCH_SIZE and WRITE_SIZE per launch in separate passes (TCC_HIT / TCC_MISS too for the north star)
# usage: bash tools/final_measure_r3.sh
set -o pipefail
out=gpurun_out/final3; mkdir -p $out; export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err || exit 1
tail -c 400 $out/bench.json; echo
for cfg in "q20 --workload quadrotor" "q50 --workload quadrotor --horizon 50 --batch 8192" "cp100 --workload cartpole"; do
  set -- $cfg; tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats_$tag -- python3 bench.py "$@" --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $out/kstats_$tag.log 2>&1 || exit 1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_${tag}_$ctr -- python3 bench.py "$@" --steps 3 --warmup 2 --no-extras --no-cpu-baseline > $out/pmc_${tag}_$ctr.log 2>&1 || exit 1
    python tools/pmc_summary.py $out/pmc_${tag}_$ctr $ctr >> $out/pmc_$tag.txt
  done
  echo "== $tag"; cat $out/pmc_$tag.txt; grep -h mpcqp_res_kernel $out/kstats_$tag/*/*kernel_stats.csv | head -2
done
for cfg in "q20 --workload quadrotor" "q50 --workload quadrotor --horizon 50 --batch 8192" "cp100 --workload cartpole"; do
  set -- $cfg; tag=$1; shift
  for ctr in TCC_HIT_sum TCC_MISS_sum; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_${tag}_$ctr -- python3 bench.py "$@" --steps 3 --warmup 2 --no-extras --no-cpu-baseline > $out/pmc_${tag}_$ctr.log 2>&1 || exit 1
    python tools/pmc_summary.py $out/pmc_${tag}_$ctr $ctr >> $out/pmc_$tag.txt
  done
  cat $out/pmc_$tag.txt
done
MPCQP_LIB=optimal_control_problem_amd/libmpcqp_timing.so python tools/timing_breakdown.py quadrotor 8192 > $out/timing_breakdown_q20.txt 2>&1
MPCQP_LIB=optimal_control_problem_amd/libmpcqp_timing.so python tools/timing_breakdown.py quadrotor 8192 - 50 > $out/timing_breakdown_q50.txt 2>&1
MPCQP_LIB=optimal_control_problem_amd/libmpcqp_timing.so python tools/timing_breakdown.py cartpole 16384 - 100 > $out/timing_breakdown_cp100.txt 2>&1
python bench.py --gpus 2 --share-gpu --no-extras --no-cpu-baseline --batch 4096 --steps 5 > $out/bench_gpus2.json 2> $out/bench_gpus2.err
python tools/sqp_bench.py quadrotor 20 8192 10 0.5 1 16 > $out/sqp_device_loop.json 2> $out/sqp_device_loop.err
python tools/config_sweep.py > $out/config_sweep.json 2> $out/config_sweep.err
python tools/stageqp_bench.py > $out/stageqp_bench.txt 2> $out/stageqp_bench.err
python tools/host_pipeline.py 4 6 8 12 > $out/host_pipeline.txt 2> $out/host_pipeline.err
tools/probes/bin/chain_stage_probe > $out/chain_stage_probe.txt 2>&1
echo done
