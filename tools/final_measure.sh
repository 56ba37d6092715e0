#!/bin/bash
# Round-end measurement pass on the GPU box (run from the repo root): everything profiles/ cites for the final build, into gpurun_out/final/
# usage: bash tools/final_measure.sh [quick]
set -o pipefail
out=gpurun_out/final; mkdir -p $out; export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err || exit 1
tail -c 600 $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $out/kstats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline > $out/pmc_write.log 2>&1 || exit 1
python tools/pmc_summary.py $out/pmc_fetch FETCH_SIZE > $out/pmc.txt; python tools/pmc_summary.py $out/pmc_write WRITE_SIZE >> $out/pmc.txt; cat $out/pmc.txt
MPCQP_LIB=optimal_control_problem_amd/libmpcqp_timing.so python tools/timing_breakdown.py quadrotor 8192 > $out/timing_breakdown.txt 2>&1
python bench.py --gpus 2 --share-gpu --no-extras --no-cpu-baseline --batch 4096 --steps 5 > $out/bench_gpus2.json 2> $out/bench_gpus2.err
[ "$1" = quick ] && exit 0
python tools/soak.py 12 > $out/soak.txt 2>&1; tail -1 $out/soak.txt
python tools/fuzz_oc.py 60 0 > $out/fuzz_oc.txt 2>&1; tail -1 $out/fuzz_oc.txt
(for cfg in "quadrotor 10" "quadrotor 12" "quadrotor 15" "quadrotor 20" "quadrotor 25" "cartpole 40" "cartpole 50" "cartpole 60" "double_integrator 60" "double_integrator 80"; do set -- $cfg; bash tools/variant_sweep.sh $1 $2 8192 default oc4 gres4 gres2; done) > $out/grid.txt 2>&1
echo done
