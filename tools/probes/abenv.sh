#!/bin/bash
# A/B of environment switches that are read at create or at every solve: each setting in a process of its own (tools/ab_bits.py keeps a variable set once it
# has set it).  usage, from the repo root on the GPU box: bash tools/probes/abenv.sh workload:horizon:batch - NAME=VALUE NAME=VALUE ... -     ("-" = no switch)
c=$1; shift
for e in "$@"; do
  if [ "$e" = "-" ]; then AB_REPS=7 AB_CASES=$c timeout -k 10 200 python tools/ab_bits.py optimal_control_problem_amd/libmpcqp.so 2>&1 | grep "libmpcqp.so" | sed "s/$/ base/";
  else env $e AB_REPS=7 AB_CASES=$c timeout -k 10 200 python tools/ab_bits.py optimal_control_problem_amd/libmpcqp.so 2>&1 | grep "libmpcqp.so" | sed "s/$/ $e/"; fi
done
