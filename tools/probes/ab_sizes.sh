for cfg in "quadrotor 50 8192" "cartpole 100 8192" "double_integrator 20 4096" "quadrotor 5 8192" "quadrotor 25 8192"; do set -- $cfg
  for lib in tools/probes/bin/libmpcqp_final6.so optimal_control_problem_amd/libmpcqp.so; do
    MPCQP_LIB=$lib timeout -k 10 200 python bench.py --workload $1 --horizon $2 --batch $3 --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 N=$2', '$lib'[-12:], round(d['value']), round(d['roofline']['kernel_ms'],3), d['solve_stats']['kernel_variant'][:28])"
  done
done
