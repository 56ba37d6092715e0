# kernel time of one workload under kernel families (MPCQP_VARIANT), same box: bash tools/variant_sweep.sh double_integrator 20 4096 [variants...]
w=${1:-double_integrator}; n=${2:-20}; b=${3:-4096}; shift 3
vs=${@:-default res1 res2 res4 gres4 gres2 stream}
for v in $vs; do
  if [ "$v" = default ]; then unset MPCQP_VARIANT; else export MPCQP_VARIANT=$v; fi
  timeout -k 10 200 python bench.py --workload $w --horizon $n --batch $b --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w N=$n x $b', '$v', round(d['value']), 'QP/s', round(d['roofline'].get('kernel_ms'),3), 'ms', d['solve_stats']['kernel_variant'], d['solve_stats']['lds_bytes_per_qp'])" || echo "$v failed"
done
